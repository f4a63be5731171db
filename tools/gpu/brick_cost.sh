#!/bin/bash
# per-rank compute cost of a W-way decomposition of the benchmark cell on one GPU (no exchange)
for w in ${WORLDS:-2 4 8}; do
timeout 600 python tools/brick_cost.py --world $w 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/brick_cost.log
