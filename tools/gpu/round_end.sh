#!/bin/bash
# Round-end evidence on one MI355X:  gpurun --timeout 3400 -- bash tools/gpu/round_end.sh r03
# full -m gpu suite (exit code captured) + smoke, tools/collect_profiles.sh <tag>, bench lines of the lmax-3 shapes, the
# world-1 RCCL soak.
TAG=${1:-r03}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/round_end_tests_full.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/round_end_tests.log
grep -E "passed|failed|error" gpurun_out/round_end_tests_full.log | tail -3 | tee -a gpurun_out/round_end_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/round_end_tests.log
timeout 1500 bash tools/collect_profiles.sh $TAG 2>&1 | tail -2 | cut -c1-300
for m in sevennet_l3i5 sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | tee gpurun_out/${TAG}_bench_${m}_n1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$m', d['ms_per_step'], d['value'], r['avg_ms'], r['frac'])"
done
timeout 900 bash tools/gpu/rccl_world1_soak.sh 2>&1 | tail -5
