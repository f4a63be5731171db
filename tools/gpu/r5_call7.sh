#!/bin/bash
# forward kernel phase stamps (full and light) beside the shipped kernel
cd $GRAFT_REPO_ROOT
for v in tree stampfl stampf; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  echo "== $v"
  timeout 300 python tools/microbench.py --layer 1 --model sevennet_0 --terms 4 --iters 7 --only "conv_fwd_fused" --stamps 2>&1 | grep -E "^conv_|stamps|phase"
done > gpurun_out/r5_call7.log 2>&1
cat gpurun_out/r5_call7.log
