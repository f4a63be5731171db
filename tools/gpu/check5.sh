#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -3
timeout 300 python tools/microbench.py --only "_fused[" --iters 3 --fv "8,0,2;4,0,2;8,1,2;4,1,2;12,1,3" > gpurun_out/g5_micro.log 2>&1
cat gpurun_out/g5_micro.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py --only "_fused[" --iters 3 --terms 2 --fv "8,0,2;4,0,2" > gpurun_out/g5_micro_t2.log 2>&1
cat gpurun_out/g5_micro_t2.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 --fv "8,0,2,1;8,0,2,2;8,0,2,4;8,0,2,8;8,0,2,16;8,0,2,6;8,0,2,15;8,0,2,31" > gpurun_out/g5_micro_diag.log 2>&1
cat gpurun_out/g5_micro_diag.log | grep -v amdgpu.ids
