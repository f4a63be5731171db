#!/bin/bash
# microbench screening of several experimental libraries (exp/libx_<name>.so) against the working tree's, in ONE gpurun call:
#   gpurun --timeout 900 -- bash tools/gpu/ab_libs.sh "conv_bwd_fused[22" name1 name2 ...
ONLY=$1; shift
for rep in 1 2; do
for v in tree "$@"; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python tools/microbench.py --layer ${LAYER:-1} --model ${MODEL:-sevennet_0} --terms 4 --iters 7 --only "$ONLY" ${FV:+--fv "$FV"} 2>&1 | grep -E "^conv_" | sed "s/^/$v  /"
done; done 2>&1 | tee gpurun_out/ab_libs.log
unset SNET_HIP_LIB
