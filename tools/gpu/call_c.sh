#!/bin/bash
# node-side kernels: uniform wave index (GEMM, hidden MLP, gates, row bounds) + fast store path of the grouped GEMM, against the tree before them
for rep in 1 2; do for v in tree pre; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$v', round(d['ms_per_step'],3), ' '.join(f'{x}:{k[x]:.3f}' for x in ('node_linear_fwd','node_linear_bwd','radial_mlp_hidden_fwd','row_bounds','conv_bwd_node[segment_sum]') if x in k), 'kernels', d['config']['kernel_ms_per_step_rank0'])"
done; done
unset SNET_HIP_LIB
timeout 1200 python -m pytest tests -x -q -m gpu -k "gemm or gate or engine or smoke or native or mlp or linear" > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tests_b.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --brick-proxy 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); bp=d['brick_proxy']; print('brick', bp['ms_per_step'], bp['ms_per_step_native_host'], bp['ideal_ms'], bp['kernel_ms_top'])"
