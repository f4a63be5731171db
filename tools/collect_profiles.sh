#!/bin/bash
# Round evidence on one MI355X (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# writes gpurun_out/<tag>_bench_n1.json, <tag>_rocprofv3_kernel_stats.csv, <tag>_pmc_hbm_traffic.json, <tag>_pmc_mfma_busy.json;
# copy them into profiles/ afterwards.  The PMC passes run alone (--kernel-trace only), one counter each.
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o r -- $BENCH > $OUT/${TAG}_stats.log 2>&1
# counter passes with the radial MLPs on the main stream: TCC counters are device-wide, concurrent kernels
# of a second stream would be charged to whichever dispatch is being sampled
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -o r -- $BENCH --no-overlap > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -o r -- $BENCH --no-overlap > $OUT/${TAG}_write.log 2>&1
# matrix-pipe busy cycles (north_star: "rocprof HBM GB/s and MFMA-busy counters"): its own pass, kernel trace only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_mfma -o r -- $BENCH --no-overlap > $OUT/${TAG}_mfma.log 2>&1
cd $ROOT
python tools/pmc_mfma_reduce.py $(find $OUT/${TAG}_mfma -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_mfma_busy.json
cp $OUT/${TAG}_pmc_mfma_busy.json profiles/${TAG}_pmc_mfma_busy.json
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
python tools/pmc_reduce.py $(find $OUT/${TAG}_fetch -name "*counter_collection.csv" | head -1) \
                           $(find $OUT/${TAG}_write -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_hbm_traffic.json
# stamp the traffic file with the kernel sources it was measured on (bench.py marks `roofline.traffic` stale when they differ)
python - <<PY
import json, sys
sys.path.insert(0, '$ROOT')
from bench import fused_source_hashes
p = '$OUT/${TAG}_pmc_hbm_traffic.json'
d = json.load(open(p))
d['__kernel_sources__'] = fused_source_hashes()
json.dump(d, open(p, 'w'), indent=1)
PY
# the bench line itself, with the fresh traffic file visible to bench.py
cp $OUT/${TAG}_pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
rm -rf $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_stats $OUT/${TAG}_mfma   # raw traces are large; summaries stay
tail -c 600 $OUT/${TAG}_bench_n1.json
