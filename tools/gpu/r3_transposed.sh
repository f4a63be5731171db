#!/bin/bash
# last layer's source gradient: transposed scalar convolution (gather form) vs g_xe rows + segment sum; parity first
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "md_scale or small_cell or ghost or sevennet_0 or parity" 2>&1 | tail -5
for rep in 1 2; do
for v in off on; do
if [ $v = off ]; then F=--no-transposed; else F=; fi
timeout 300 python bench.py --no-cpu-baseline $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
k=r['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],2), {a:b for a,b in k.items() if '005c' in a or 'transposed' in a or 'segment' in a or '0b99' in a})"
done; done 2>&1 | tee gpurun_out/r3_transposed.log
