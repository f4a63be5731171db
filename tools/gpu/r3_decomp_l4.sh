#!/bin/bash
# timing decomposition of the LAST layer's reverse kernel (three one-path x blocks: its sub-steps are almost pure overhead)
export SNET_HIP_LIB=$PWD/exp/libx_exp4.so
timeout 600 python tools/microbench.py --terms 4 --iters 5 --layer 4 --only "conv_bwd_fused[005" \
  --fv "4,0,3;4,0,3,1;4,0,3,6;4,0,3,7;4,0,3,8;4,0,3,16;4,0,3,32;4,0,3,64;4,0,3,128;4,0,3,160;4,0,3,166;4,0,3,255;4,0,2;8,0,2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_decomp_l4.log
