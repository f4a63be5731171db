#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/g12_all.log
tail -6 gpurun_out/g12_all.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 200 python tools/microbench.py --only "_fused[" --iters 3 --terms 2 2>&1 | grep -v amdgpu.ids
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/g12_bench_fused.json 2> gpurun_out/g12_bench_fused.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fused off > gpurun_out/g12_bench_off.json 2> gpurun_out/g12_bench_off.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --host native > gpurun_out/g12_bench_native.json 2> gpurun_out/g12_bench_native.err
SNET_NO_FUSED=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --host native --fused off > gpurun_out/g12_bench_native_off.json 2> gpurun_out/g12_bench_native_off.err
python - <<'PY'
import json
for f in ('fused', 'off', 'native', 'native_off'):
    try:
        d = json.loads(open(f'gpurun_out/g12_bench_{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['ms_per_step'], 2), 'dominant', d['roofline']['kernel'], round(d['roofline']['avg_ms'], 3), d['roofline']['kernel_ms_per_step'])
    except Exception as e:
        print(f, 'FAILED', e, open(f'gpurun_out/g12_bench_{f}.err').read()[-600:])
PY
