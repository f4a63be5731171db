#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/g32_tests_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/g32_tests_full.log | tail -3
for m in sevennet_0 sevennet_l3i5; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | tee gpurun_out/g32_bench_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], d['value'], r['avg_ms'], r['frac']); print({k:v for k,v in r['kernel_ms_per_step'].items() if 'fwd' in k})"
done
