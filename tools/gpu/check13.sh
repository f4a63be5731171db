#!/bin/bash
mkdir -p gpurun_out
SNET_PY_ARENA_GB=80 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/g13_bench_arena.json 2> gpurun_out/g13_bench_arena.err
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/g13_bench_exp.json 2> gpurun_out/g13_bench_exp.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/g13_bench_base.json 2> gpurun_out/g13_bench_base.err
python - <<'PY'
import json
for f in ('arena', 'exp', 'base'):
    try:
        d = json.loads(open(f'gpurun_out/g13_bench_{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['ms_per_step'], 2), 'dominant', d['roofline']['kernel'], round(d['roofline']['avg_ms'], 3))
    except Exception as e:
        print(f, 'FAILED', e, open(f'gpurun_out/g13_bench_{f}.err').read()[-800:])
PY
