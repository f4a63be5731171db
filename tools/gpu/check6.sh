#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 --fv "8,0,2,31;8,0,2,63;8,0,2,95;8,0,2,127;8,0,2,159;8,0,2,255;8,0,2,32;8,0,2,128;8,0,2,160;4,0,2,31;4,0,2,63;4,0,2,255" > gpurun_out/g6_micro_diag.log 2>&1
cat gpurun_out/g6_micro_diag.log | grep -v amdgpu.ids
