#!/bin/bash
for reps in 12 8; do
for host in python native; do
echo "== reps $reps host $host"
timeout 300 python bench.py --no-cpu-baseline --reps $reps --host $host --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], sum(d['roofline']['kernel_ms_per_step'].values()))"
done; done 2>&1 | tee gpurun_out/g23_small.log
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/g23_bench.json | cut -c1-250
