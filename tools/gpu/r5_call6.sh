#!/bin/bash
cd $GRAFT_REPO_ROOT
LAYER=1 bash tools/gpu/ab_libs.sh "conv_fwd_fused" nofrow > gpurun_out/r5_call6.log 2>&1
echo "== parity" >> gpurun_out/r5_call6.log
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error" >> gpurun_out/r5_call6.log
cat gpurun_out/r5_call6.log
