#!/bin/bash
# node-level kernels: op tests, then the linears of the middle layer with A rows direct (SNET_GEMM_ALDS=0) / through LDS, and the step
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gate or gemm or linear" 2>&1 | tail -3
for v in 0 1; do
echo "== SNET_GEMM_ALDS=$v"
export SNET_GEMM_ALDS=$v
timeout 300 python tools/microbench.py --only si --iters 7 2>/dev/null | grep -v "^lib="
timeout 300 python tools/microbench.py --only sc_ --iters 7 2>/dev/null | grep -v "^lib="
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],2), d['config']['energy'])"
done 2>&1 | tee gpurun_out/r3_node_ops.log
