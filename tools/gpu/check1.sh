#!/bin/bash
# round-2 GPU check #1: parity of the fused kernels, then timing (run through gpurun from the repo root)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -25 > gpurun_out/g1_fused_ops.log
tail -3 gpurun_out/g1_fused_ops.log
timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "fused_engine" 2>&1 | tail -25 > gpurun_out/g1_fused_engine.log
tail -3 gpurun_out/g1_fused_engine.log
timeout 200 python tools/microbench.py --only fused --iters 5 > gpurun_out/g1_micro.log 2>&1
timeout 200 python tools/microbench.py --only hidden --iters 5 >> gpurun_out/g1_micro.log 2>&1
cat gpurun_out/g1_micro.log | tail -12
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g1_bench_fused.json 2> gpurun_out/g1_bench_fused.err
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --fused off > gpurun_out/g1_bench_off.json 2> gpurun_out/g1_bench_off.err
python - <<'PY'
import json
for f in ('gpurun_out/g1_bench_fused.json', 'gpurun_out/g1_bench_off.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/g1_all.log
tail -5 gpurun_out/g1_all.log
