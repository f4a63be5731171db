#!/bin/bash
# instruction-fetch counters of the fused kernels (the reverse kernel is ~60 KB of straight-line code per tile pass)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/ic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u | grep -i "ICACHE\|IFETCH\|INST_PREFETCH\|WAIT_IFETCH\|DCACHE" | tr '\n' ' ' > $OUT/avail.txt; cat $OUT/avail.txt; echo
CMD="python $ROOT/tools/microbench.py --terms 4 --iters 2 --only fused"
i=0
for grp in "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH_LEVEL SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o r -- $CMD > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY' | tee -a $ROOT/gpurun_out/icache_counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'conv_bwdf' in k or 'conv_fwdf' in k:
        acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
  else
    echo "group $i ($grp): no counter file" | tee -a $ROOT/gpurun_out/icache_counters.txt; tail -2 $OUT/p$i.log
  fi
done
rm -rf $OUT/p*
