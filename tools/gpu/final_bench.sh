#!/bin/bash
# the driver's bench line (default flags, CPU baseline included) and config 5's whole cell, final build
python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err; tail -c 300 gpurun_out/r03_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_n1.json')); r=d['roofline']; k=r['kernel_ms_per_step']
print('sevennet_0', d['ms_per_step'], d['value'], r['avg_ms'], r['frac'], 'hidden', k.get('radial_mlp_hidden_fwd'), d['cpu_baseline']['by_threads'])"
timeout 900 python bench.py --no-cpu-baseline --model sevennet_mf_ompa --reps 29 --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/r03_bench_mf_ompa_195k_n1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('mf_ompa 195k', round(d['ms_per_step'],2), 'ms', round(d['value']), 'atom-steps/s', d['config']['atoms'], d['config']['edges'], d['config'].get('peak_device_memory_gb'))"
