#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "rccl_halo or fused_engine or sevennet_0_shape" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -3
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/g16_bench.json 2> gpurun_out/g16_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/g16_bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'], 2), d['roofline']['kernel'], round(d['roofline']['avg_ms'], 3), d['roofline']['frac'], d['roofline'].get('traffic'))
PY
