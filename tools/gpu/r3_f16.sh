#!/bin/bash
# f16x3 in-kernel products: op-level test, MD-scale parity tests, accuracy table, step time per mode
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_matches" 2>&1 | tail -15 | tee gpurun_out/r3_f16_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "md_scale or small_cell or rank_without" 2>&1 | tail -15 | tee gpurun_out/r3_f16_engine.log
timeout 900 python tools/gpu/terms_accuracy.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_terms_accuracy2.log
for t in 4 2; do
timeout 300 python bench.py --no-cpu-baseline --terms $t 2>/dev/null | tee gpurun_out/r3_bench_f16_t$t.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('terms $t', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), r['kernel_ms_per_step'])"
done 2>&1 | tee gpurun_out/r3_bench_f16.log
