#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -5
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 --terms 2 --fv "8,0,2;4,0,2;4,0,3;4,1,3;12,0,3;12,1,3;8,1,4;4,0,3,8;4,0,3,16;4,0,3,255" > gpurun_out/g9_micro_t2.log 2>&1
cat gpurun_out/g9_micro_t2.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 --terms 3 --fv "8,0,2;4,0,2;12,1,3" > gpurun_out/g9_micro_t3.log 2>&1
cat gpurun_out/g9_micro_t3.log | grep -v amdgpu.ids
