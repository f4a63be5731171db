#!/bin/bash
# A/B of two builds in ONE gpurun call (boxes differ by a few percent): exp/libx_head.so (a copy of the previous build's
# libsnet_hip.so, or built from another tree) against the working tree's library, interleaved
for rep in 1 2; do
for v in head new; do
if [ $v = head ]; then export SNET_HIP_LIB=$PWD/exp/libx_head.so; else unset SNET_HIP_LIB; fi
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
k=r['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],2), 'bwd', round(r['avg_ms'],3), {a: b for a, b in k.items() if 'ecc5' in a or '005c' in a})"
done; done 2>&1 | tee gpurun_out/ab_bench.log
