#!/bin/bash
# step-level A/B of experiment libraries (exp/libx_<name>.so) against the working tree's, interleaved in ONE gpurun call:
#   gpurun --timeout 900 -- bash tools/gpu/ab_step.sh name1 name2 ...
for rep in 1 2; do
for v in tree "$@"; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
keys=('conv_bwd_fused[22d6a77ad5ac]','conv_fwd_fused[22d6a77ad5ac]','conv_bwd_node[segment_sum]','node_linear_fwd','node_linear_bwd')
print('$v', round(d['ms_per_step'],2), ' '.join(f'{k.get(x,0):.3f}' for x in keys))"
done; done 2>&1 | tee gpurun_out/ab_step.log
unset SNET_HIP_LIB
