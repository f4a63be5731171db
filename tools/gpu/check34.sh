#!/bin/bash
for v in f1 f2; do
echo "== $v"
SNET_HIP_LIB=$PWD/exp/libx_$v.so timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], d['value'], d['config']['energy']); print({k:v for k,v in r['kernel_ms_per_step'].items() if 'fwd_fused' in k})"
done
