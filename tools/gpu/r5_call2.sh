#!/bin/bash
# round 5, call 2: packed-fp32 reverse bodies (exp/libx_pk.so) against the working tree, all three SevenNet-0 layer classes + l3i5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for ml in sevennet_0:1 sevennet_0:0 sevennet_0:4 sevennet_l3i5:2; do
m=${ml%%:*}; l=${ml##*:}
for rep in 1 2; do
for v in tree pk; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python tools/microbench.py --model $m --layer $l --terms 4 --iters 7 --only "conv_bwd_fused" 2>&1 | grep -E "^conv_" | sed "s/^/$v $ml  /"
done; done; done
echo "== parity (fused == separate kernels)"
SNET_HIP_LIB=$PWD/exp/libx_pk.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_fused_matches_separate_kernels" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r5_call2.log
