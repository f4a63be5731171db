#!/bin/bash
# D3 dispersion tests; forward kernel staging A/B (exp/libx_f12.so: 12-wave forward kernel with register staging of the weight
# slabs instead of direct global->LDS; exp/libx_head.so = the shipped configuration), interleaved
timeout 900 python -m pytest tests/test_d3_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r3_d3_tests.log
for rep in 1 2; do
for v in head f12; do
export SNET_HIP_LIB=$PWD/exp/libx_$v.so
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
k=r['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],2), 'bwd', round(r['avg_ms'],3), 'fwd_mid', k.get('conv_fwd_fused[22d6a77ad5ac]'), 'fwd_l4', k.get('conv_fwd_fused[005c575f8ec2]'), 'fwd_l0', k.get('conv_fwd_fused[ecc5d202727d]'))"
done; done 2>&1 | tee gpurun_out/r3_f12_ab.log
unset SNET_HIP_LIB
for m in sevennet_l3i5 sevennet_mf_ompa; do
for v in head f12; do
export SNET_HIP_LIB=$PWD/exp/libx_$v.so
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$m $v', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3))"
done; done 2>&1 | tee -a gpurun_out/r3_f12_ab.log
