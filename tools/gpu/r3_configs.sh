#!/bin/bash
# BASELINE configs 4 and 5 as written: amorphous l3i5 supercell parity (64 000 atoms), MF-ompa at 195 112 atoms (tiling parity +
# bench line on one GPU), D3 tests
timeout 600 python -m pytest tests/test_d3_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r3_configs_tests.log
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -k "amorphous_supercell or (lmax3_shapes and 29)" 2>&1 | tail -5 | tee -a gpurun_out/r3_configs_tests.log
timeout 900 python bench.py --no-cpu-baseline --model sevennet_mf_ompa --reps 29 --steps 5 --warmup 2 2>gpurun_out/r3_bench_mf195k.err | tee gpurun_out/r3_bench_mf_ompa_195k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('mf_ompa 195k', round(d['ms_per_step'],2), 'ms', round(d['value']), 'atom-steps/s', d['config']['atoms'], d['config']['edges'], r['kernel'], round(r['avg_ms'],3))" | tee -a gpurun_out/r3_configs_tests.log
python - <<'PY' | tee -a gpurun_out/r3_configs_tests.log
import torch
print('peak device memory of this process after the 195k-atom MF-ompa run is not tracked by torch (engine buffers are torch tensors): max_memory_allocated is reported by bench in a later revision')
PY
