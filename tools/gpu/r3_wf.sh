#!/bin/bash
# in-kernel software pipelining knobs of the reverse kernel (wfirst: both tiles' weight products before the first tensor-
# product body; gpf: g-part fragments requested before the bodies), interleaved with the shipped build
for rep in 1 2; do
for v in new wfirst gpf wfirstgpf; do
if [ $v = new ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
k=r['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],2), 'bwd', round(r['avg_ms'],3), 'bwd_l4', k.get('conv_bwd_fused[005c575f8ec2]'), 'bwd_l0', k.get('conv_bwd_fused[ecc5d202727d]'))"
done; done 2>&1 | tee gpurun_out/r3_wf_ab.log
