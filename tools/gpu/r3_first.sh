#!/bin/bash
# round-3 opening measurement: precision modes at MD-scale forces, step time per mode with the round-2 kernels
timeout 900 python tools/gpu/terms_accuracy.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_terms_accuracy.log
for t in 2 3; do
timeout 300 python bench.py --no-cpu-baseline --terms $t 2>/dev/null | tee gpurun_out/r3_bench_t$t.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('terms $t', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), r['kernel_ms_per_step'])"
done 2>&1 | tee gpurun_out/r3_bench_terms.log
