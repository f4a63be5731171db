#!/bin/bash
# per-rank compute of the actual bricks of a W-way split on one GPU (no exchange) + host enqueue time, both hosts at brick size
for w in 2 4 8; do timeout 300 python tools/brick_cost.py --world $w 2>/dev/null | head -1; done | tee gpurun_out/r3_bricks.log
for h in python native; do
timeout 300 python bench.py --no-cpu-baseline --reps 12 --host $h 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$h 13824 atoms', round(d['ms_per_step'],3), 'kernel sum', round(sum(d['roofline']['kernel_ms_per_step'].values()),3))"
done | tee -a gpurun_out/r3_bricks.log
