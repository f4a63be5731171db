#!/bin/bash
# reverse middle: packed gy on/off; forward: packed bodies on/off; per kernel class in the step
for rep in 1 2; do for v in tree nopkgy nofpk d1; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python tools/microbench.py --layer 1 --model sevennet_0 --terms 4 --iters 7 --only "conv_" 2>&1 | grep -E "conv_(bwd|fwd)_fused\[" | sed "s/^/$v  /"
done; done
unset SNET_HIP_LIB
for v in tree nofpk nopkgy d1 tree nofpk nopkgy d1; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],2), ' '.join(f'{x.split(\"[\")[-1][:6]}:{v:.3f}' for x,v in k.items() if 'conv_' in x and 'fused' in x))"
done
