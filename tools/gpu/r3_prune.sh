#!/bin/bash
# MF-ompa with the unread paths of its third layer not evaluated: parity (small cell, MD scale, 27 000-atom tiling, native host), bench
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_native_model_gpu.py -x -q -m gpu -k "mf_ompa or multi_modal or native" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
for m in sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | tee gpurun_out/r03c_bench_${m}_n1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernel_ms_per_step']
print('$m', round(d['ms_per_step'],2), {a: b for a, b in sorted(k.items(), key=lambda kv: -kv[1])[:8]})"
done 2>&1 | tee gpurun_out/r3_prune.log
