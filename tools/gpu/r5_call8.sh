#!/bin/bash
cd $GRAFT_REPO_ROOT
LAYER=1 bash tools/gpu/ab_libs.sh "conv_fwd_fused" f6 f4 > gpurun_out/r5_call8.log 2>&1
cat gpurun_out/r5_call8.log
