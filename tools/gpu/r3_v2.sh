#!/bin/bash
# v2 reverse body + f16x3 with a-priori per-edge bound: tests, accuracy, step time (A/B against bf16x3)
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_matches" 2>&1 | tail -15 | tee gpurun_out/r3_v2_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "md_scale or small_cell or rank_without or fused_engine" 2>&1 | tail -15 | tee gpurun_out/r3_v2_engine.log
timeout 900 python tools/gpu/terms_accuracy.py --modes separate,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_terms_accuracy3.log
for t in 4 2 4; do
timeout 300 python bench.py --no-cpu-baseline --terms $t 2>/dev/null | tee gpurun_out/r3_bench_v2_t$t.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('terms $t', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), r['kernel_ms_per_step'])"
done 2>&1 | tee gpurun_out/r3_bench_v2.log
for m in sevennet_l3i5 sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | tee gpurun_out/r3_bench_v2_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$m', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), r['kernel_ms_per_step'])"
done 2>&1 | tee -a gpurun_out/r3_bench_v2.log
