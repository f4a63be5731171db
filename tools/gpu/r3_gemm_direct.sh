#!/bin/bash
# barrier-free GEMM k loop (weight fragments straight from L2, SNET_GEMM_DIRECT=1) against the LDS form: op tests, linears, step
SNET_GEMM_DIRECT=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_split or linear" 2>&1 | tail -2
for v in 0 1; do
echo "== SNET_GEMM_DIRECT=$v"
export SNET_GEMM_DIRECT=$v
timeout 300 python tools/microbench.py --only si --iters 7 2>/dev/null | grep -v "^lib=\|aliased\|f16"
timeout 300 python tools/microbench.py --only sc_ --iters 7 2>/dev/null | grep -v "^lib=\|f16"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('step', round(d['ms_per_step'],2), 'linears', k.get('node_linear_fwd'), k.get('node_linear_bwd'), d['config']['energy'])"
done 2>&1 | tee gpurun_out/r3_gemm_direct.log
