#!/bin/bash
# timing decomposition of the fused kernels (kernel-tuning build exp/libx_exp.so: SNET_CODEGEN_OPTS=fexp=22d6a77ad5ac):
# diag bits: 1 no tensor product, 2 no g_h2 products, 4 no w products, 8 no g_out loads, 16 no g_xe stores,
# 32 no slab staging, 64 no x prefetch, 128 no barrier
export SNET_HIP_LIB=$PWD/exp/libx_exp.so
timeout 600 python tools/microbench.py --terms 4 --iters 5 --only "conv_bwd_fused[22" \
  --fv "4,0,3;4,0,3,1;4,0,3,2;4,0,3,4;4,0,3,6;4,0,3,7;4,0,3,8;4,0,3,16;4,0,3,32;4,0,3,64;4,0,3,128;4,0,3,24;4,0,3,255;4,1,3;8,1,4;4,0,2;8,0,2;8,1,2;12,0,3;12,1,3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_decomp_bwd.log
timeout 600 python tools/microbench.py --terms 4 --iters 5 --only "conv_fwd_fused[22" \
  --fv "12,1,3;12,1,3,1;8,1,2;12,0,3;4,0,3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_decomp_fwd.log
unset SNET_HIP_LIB
# the shipped library: step time with the Cauchy-Schwarz row bounds
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3_bench_cs.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), r['frac'], r['kernel_ms_per_step'])"
bash tools/gpu/rccl_world1_soak.sh
