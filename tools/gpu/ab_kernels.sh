#!/bin/bash
# A/B of the fused tensor-product kernels of two builds inside ONE gpurun call (boxes differ by a few percent):
#   gpurun --timeout 1500 -- bash tools/gpu/ab_kernels.sh exp/libx_old.so [model:layer ...]
# the working tree's library ("new") against $1 ("old"), interleaved per shape, microbench timings of the reverse and forward kernels
OLD=$PWD/${1:?old library}; shift
SHAPES=${@:-"sevennet_0:1 sevennet_0:0 sevennet_0:4"}
for ml in $SHAPES; do
  m=${ml%%:*}; l=${ml##*:}
  for v in old new old new; do
    if [ $v = old ]; then export SNET_HIP_LIB=$OLD; else unset SNET_HIP_LIB; fi
    timeout 600 python tools/microbench.py --model $m --layer $l --terms 4 --iters 5 --only "conv_" 2>&1 | grep -E "conv_(bwd|fwd)_fused|conv_bwd_node" | sed "s/^/$v $m:$l  /"
  done
done 2>&1 | tee gpurun_out/ab_kernels.log
unset SNET_HIP_LIB
