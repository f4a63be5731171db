#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "lmax3_shapes or rccl_halo" > gpurun_out/g18_tests.log 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/g18_tests.log | tail -8
timeout 300 python bench.py --model sevennet_l3i5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g18_bench_l3i5.json 2> gpurun_out/g18_bench_l3i5.err
timeout 300 python bench.py --model sevennet_l3i5 --steps 5 --warmup 2 --no-cpu-baseline --fused off > gpurun_out/g18_bench_l3i5_off.json 2> gpurun_out/g18_bench_l3i5_off.err
timeout 300 python bench.py --model sevennet_mf_ompa --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g18_bench_mf.json 2> gpurun_out/g18_bench_mf.err
timeout 300 python bench.py --model sevennet_mf_ompa --steps 5 --warmup 2 --no-cpu-baseline --fused off > gpurun_out/g18_bench_mf_off.json 2> gpurun_out/g18_bench_mf_off.err
python - <<'PY'
import json
for f in ('l3i5', 'l3i5_off', 'mf', 'mf_off'):
    try:
        d = json.loads(open(f'gpurun_out/g18_bench_{f}.json').read().strip().splitlines()[-1])
        print(f, d['config']['atoms'], round(d['ms_per_step'], 2), round(d['value']), d['roofline']['kernel'], round(d['roofline']['avg_ms'], 3), round(d['roofline']['frac'], 3))
        print('   ', dict(list(d['roofline']['kernel_ms_per_step'].items())[:8]))
    except Exception as e:
        print(f, 'FAILED', e, open(f'gpurun_out/g18_bench_{f}.err').read()[-500:])
PY
