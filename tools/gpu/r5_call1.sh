#!/bin/bash
# round 5, call 1: phase stamps of the shipped reverse kernel + screening of the in-wave pipeline variants (one box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== stamps (exp/libx_stamp.so)"
SNET_HIP_LIB=$PWD/exp/libx_stamp.so timeout 300 python tools/microbench.py --layer 1 --terms 4 --iters 5 --only "conv_bwd_fused" --stamps 2>&1 | grep -v "^lib="
echo "== variants"
for rep in 1 2; do
for v in tree pipe1 pipe2 pipe2s; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python tools/microbench.py --layer 1 --terms 4 --iters 7 --only "conv_bwd_fused" 2>&1 | grep -E "^conv_" | sed "s/^/$v  /"
done; done
echo "== parity of the variants (fused == separate kernels, middle layer shapes)"
for v in pipe1 pipe2 pipe2s; do
  SNET_HIP_LIB=$PWD/exp/libx_$v.so timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_fused_matches_separate_kernels" 2>&1 | tail -2 | sed "s/^/$v  /"
done
} 2>&1 | tee gpurun_out/r5_call1.log
