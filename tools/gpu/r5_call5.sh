#!/bin/bash
# forward kernel: short tiles spread over the lane groups (frow) against the previous row order, microbench + step + parity
cd $GRAFT_REPO_ROOT
for L in 1 0 4; do LAYER=$L bash tools/gpu/ab_libs.sh "conv_fwd_fused" nofrow | sed "s/^/L$L /"; done > gpurun_out/r5_call5.log 2>&1
bash tools/gpu/ab_step.sh nofrow >> gpurun_out/r5_call5.log 2>&1
echo "== parity" >> gpurun_out/r5_call5.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -q -x 2>&1 | tail -5 >> gpurun_out/r5_call5.log
cat gpurun_out/r5_call5.log
