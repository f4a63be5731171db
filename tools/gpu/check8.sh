#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 --terms 2 --fv "4,0,3;4,0,3,8;4,0,3,256;4,0,3,16;4,0,3,64;4,0,3,1;4,0,3,6;4,0,3,32;4,0,3,128;4,0,3,255" > gpurun_out/g8_micro.log 2>&1
cat gpurun_out/g8_micro.log | grep -v amdgpu.ids
