#!/bin/bash
# first interaction layer's reverse kernel: one (a) / two (b) channel tiles per block, launch configurations
for v in l0a l0b; do
export SNET_HIP_LIB=$PWD/exp/libx_$v.so
echo "== $v"
timeout 600 python tools/microbench.py --terms 4 --iters 7 --layer 0 --only "conv_bwd_fused[ecc" --fv "8,0,2;4,0,3;4,0,2;12,0,3;4,1,3" 2>&1 | grep -v "amdgpu.ids\|^lib="
timeout 600 python tools/microbench.py --terms 4 --iters 7 --layer 0 --only "conv_fwd_fused[ecc" --fv "8,1,2;12,1,3;12,0,3;8,0,2" 2>&1 | grep -v "amdgpu.ids\|^lib="
done | tee gpurun_out/r3_l0.log
