#!/bin/bash
# the CPU baseline on BASELINE config 2's size (10 648 atoms, ~4 min of host time): bench line with that sample
timeout 1200 python bench.py --cpu-reps 11 --steps 5 --warmup 2 2> gpurun_out/bench_cpu10k.err | tee gpurun_out/bench_cpu10k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cpu_baseline'])"
