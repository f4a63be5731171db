#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -3
timeout 300 python tools/gpu/terms_accuracy.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g4_terms.log
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 \
   --fv "8,0,2;4,0,2;4,1,3;6,1,3;6,0,3;6,1,2;12,1,3;8,1,4" > gpurun_out/g4_micro_bwd.log 2>&1
cat gpurun_out/g4_micro_bwd.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py --only "_fused[" --iters 3 --terms 2 > gpurun_out/g4_micro_t2.log 2>&1
cat gpurun_out/g4_micro_t2.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py --only "_fused[" --iters 3 --terms 1 > gpurun_out/g4_micro_t1.log 2>&1
cat gpurun_out/g4_micro_t1.log | grep -v amdgpu.ids
