#!/bin/bash
# round 5, call 3: no-gx instantiation (tree) against the previous kernels (exp/libx_nopk.so = round-4 bodies, gx always formed) and packed
# bodies on every shape (exp/libx_pkall.so): first / last layer reverse kernels, then the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for ml in sevennet_0:0 sevennet_0:4 sevennet_0:1; do
m=${ml%%:*}; l=${ml##*:}
for rep in 1 2; do
for v in nopk tree pkall; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python tools/microbench.py --model $m --layer $l --terms 4 --iters 7 --only "conv_bwd_fused" 2>&1 | grep -E "^conv_" | sed "s/^/$v $ml  /"
done; done; done
unset SNET_HIP_LIB
bash tools/gpu/ab_step.sh nopk pkall
echo "== parity"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_native_model_gpu.py -q -x -k "conv_fused_matches_separate_kernels or fused_convolution_module or native_model_vs_engine" 2>&1 | tail -2
SNET_HIP_LIB=$PWD/exp/libx_pkall.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_fused_matches_separate_kernels" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r5_call3.log
