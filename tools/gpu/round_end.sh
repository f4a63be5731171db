#!/bin/bash
# Round-end evidence on one MI355X:  gpurun --timeout 3000 -- bash tools/gpu/round_end.sh
# full -m gpu suite (exit code captured), tools/collect_profiles.sh r02, bench lines of the lmax-3 shapes.
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/round_end_tests_full.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/round_end_tests.log
grep -E "passed|failed|error" gpurun_out/round_end_tests_full.log | tail -3 | tee -a gpurun_out/round_end_tests.log
timeout 1500 bash tools/collect_profiles.sh r02 2>&1 | tail -2 | cut -c1-300
for m in sevennet_l3i5 sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | tee gpurun_out/round_end_bench_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], d['value'], r['avg_ms'], r['frac'])"
done
