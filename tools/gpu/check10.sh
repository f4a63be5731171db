#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -8
timeout 300 python tools/microbench.py --only "conv_fwd_fused[" --iters 3 --terms 2 --fv "8,0,2;4,0,2;4,0,3;12,0,3;12,1,3;8,1,2" > gpurun_out/g10_micro_t2.log 2>&1
cat gpurun_out/g10_micro_t2.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py --only "conv_fwd_fused[" --iters 3 --terms 3 --fv "8,0,2;4,0,2;8,1,2" > gpurun_out/g10_micro_t3.log 2>&1
cat gpurun_out/g10_micro_t3.log | grep -v amdgpu.ids
