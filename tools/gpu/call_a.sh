#!/bin/bash
# round 6: telemetry pass outside the bracket, the bench dry-run test with its full failure text
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06_bench_baseline.json 2> gpurun_out/r06_bench_baseline.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_baseline.json')); c=d['config']
print(d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'], d['value'], {k:c.get(k) for k in ('sclk_mhz','socket_power_w','temp_junction_c','telemetry_samples','sclk_mhz_min_max','socket_power_w_min_max','telemetry_pass_ms_per_step')})
PY
timeout 300 python bench.py --reps 6 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -c 1500
timeout 900 python -m pytest tests -x -q -m gpu -k "fused_convolution_module or bench_multi_rank" 2>&1 | tail -15
