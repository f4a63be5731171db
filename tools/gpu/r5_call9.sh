#!/bin/bash
cd $GRAFT_REPO_ROOT
V="$@"
for L in 1 4 0; do LAYER=$L bash tools/gpu/ab_libs.sh "conv_fwd_fused" $V | sed "s/^/L$L /"; done > gpurun_out/r5_call9.log 2>&1
for v in $V; do SNET_HIP_LIB=$PWD/exp/libx_$v.so timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "fused or conv" -rf 2>&1 | grep -E "passed|failed|FAILED" | sed "s/^/$v /" >> gpurun_out/r5_call9.log; done
cat gpurun_out/r5_call9.log
