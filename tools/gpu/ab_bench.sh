#!/bin/bash
# A/B of two builds in ONE gpurun call (boxes differ by a few percent): exp/libx_head.so (built from another tree:
#   git worktree add /tmp/wt HEAD && (cd /tmp/wt && python -m sevennet_amd.build) && cp /tmp/wt/sevennet_amd/libsnet_hip.so exp/libx_head.so)
# against the working tree's library, interleaved
for rep in 1 2; do
for v in head new; do
if [ $v = head ]; then export SNET_HIP_LIB=$PWD/exp/libx_head.so; else unset SNET_HIP_LIB; fi
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
k=r['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],2), 'bwd', round(r['avg_ms'],3), 'fwd_mid', k.get('conv_fwd_fused[22d6a77ad5ac]'), 'bwd_l4', k.get('conv_bwd_fused[005c575f8ec2]'), 'bwd_l0', k.get('conv_bwd_fused[ecc5d202727d]'))"
done; done 2>&1 | tee gpurun_out/ab_bench.log
