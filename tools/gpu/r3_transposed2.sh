#!/bin/bash
# transposed scalar convolution in both hosts: parity tests, then python / native bench lines
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_native_model_gpu.py tests/test_md_host_gpu.py -x -q -m gpu -k "not full_size and not amorphous" 2>&1 | tail -6
for h in; do
timeout 300 python bench.py --no-cpu-baseline --host $h 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$h', round(d['ms_per_step'],2))"
done 2>&1 | tee gpurun_out/r3_transposed2.log
SNET_NO_TRANSPOSED=1 timeout 300 python bench.py --no-cpu-baseline --host native 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('native, per-edge rows', round(d['ms_per_step'],2))" | tee -a gpurun_out/r3_transposed2.log
