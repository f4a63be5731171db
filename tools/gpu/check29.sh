#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/g29_tests_full.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/g29_tests.log
grep -E "passed|failed|error" gpurun_out/g29_tests_full.log | tail -3 | tee -a gpurun_out/g29_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/g29_smoke.log
timeout 300 python bench.py --no-cpu-baseline --model sevennet_mf_ompa 2>/dev/null | tee gpurun_out/g29_bench_mf.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], d['value'], r['avg_ms'], r['frac']); print(r['kernel_ms_per_step'])"
