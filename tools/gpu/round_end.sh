#!/bin/bash
# Round-end evidence on one MI355X:  gpurun --timeout 4500 -- bash tools/gpu/round_end.sh r06
# full -m gpu suite + smoke (tools/gpu/tests_only.sh: exit codes printed, nothing piped through tail),
# tools/collect_profiles.sh <tag> (rocprofv3 stats, PMC traffic, matrix-pipe busy, the default bench line with its CPU baseline),
# bench lines of the lmax-3 shapes and of the 8-way brick proxy, SQ counters, MD host cost, the world-1 RCCL soak.
set -o pipefail
TAG=${1:-r06}
if [ -z "$SKIP_TESTS" ]; then   # (SKIP_TESTS=1: the suite already ran on this tree in an earlier call)
bash tools/gpu/tests_only.sh
echo "tests_only rc=$?" | tee -a gpurun_out/tests.log
cp gpurun_out/tests.log gpurun_out/${TAG}_gpu_tests.txt
fi
timeout 1800 bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; echo "collect_profiles rc=$?"
tail -2 gpurun_out/${TAG}_collect.log | cut -c1-300
for m in sevennet_l3i5 sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null > gpurun_out/${TAG}_bench_${m}_n1.json; echo "bench $m rc=$?"
python -c "
import json,sys
d=json.loads(open('gpurun_out/${TAG}_bench_${m}_n1.json').read()); r=d['roofline']
print('$m', d['ms_per_step'], d['value'], r['avg_ms'], r['frac'], d['config'].get('sclk_mhz'), d['config'].get('socket_power_w'))"
done
timeout 900 python bench.py --no-cpu-baseline --model sevennet_mf_ompa --reps 29 --steps 5 --warmup 2 2>/dev/null > gpurun_out/${TAG}_bench_mf_ompa_195k_n1.json; echo "bench mf_ompa 195k rc=$?"
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --brick-proxy 8 2>/dev/null > gpurun_out/${TAG}_bench_brick8.json; echo "bench brick proxy rc=$?"
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --host native 2>/dev/null > gpurun_out/${TAG}_bench_native_host_n1.json; echo "bench native host rc=$?"
timeout 900 bash tools/gpu/rccl_world1_soak.sh > gpurun_out/${TAG}_soak.log 2>&1; echo "soak rc=$?"; tail -5 gpurun_out/${TAG}_soak.log
# SQ-level counters of the fused kernels (vector instructions per launch, wait / issue shares, matrix-pipe share)
GROUPS_ONLY="1 3 4" timeout 900 bash tools/gpu/sq_counters.sh > gpurun_out/${TAG}_sq.log 2>&1; echo "sq_counters rc=$?"
cp gpurun_out/sq/summary.txt gpurun_out/${TAG}_pmc_sq_fused_kernels.txt 2>/dev/null
timeout 300 python tools/md_host_cost.py > gpurun_out/${TAG}_md_host_cost.txt 2>&1; tail -3 gpurun_out/${TAG}_md_host_cost.txt
