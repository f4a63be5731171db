#!/bin/bash
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/g27_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], r['avg_ms'], r['frac']); print(r['kernel_ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --model sevennet_l3i5 2>/dev/null | tee gpurun_out/g27_bench_l3i5.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], r['avg_ms'], r['frac']); print(r['kernel_ms_per_step'])"
