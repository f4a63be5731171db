#!/bin/bash
# hidden radial layers in one launch: parity subset, bench line, brick proxy
timeout 1500 python -m pytest tests -x -q -m gpu  > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tests_b.log
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('tree', round(d['ms_per_step'],3), ' '.join(f'{x}:{k[x]:.3f}' for x in ('radial_mlp_hidden_fwd','node_linear_fwd','node_linear_bwd','gate_fwd','gate_bwd','row_bounds','conv_fwd_fused[22d6a77ad5ac]','conv_bwd_fused[22d6a77ad5ac]') if x in k), 'kernels', d['config']['kernel_ms_per_step_rank0'], 'disp', d['config']['dispatches_per_step_rank0'])"
done
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --brick-proxy 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); bp=d['brick_proxy']; print('brick', bp['ms_per_step'], bp['ms_per_step_native_host'], bp['ideal_ms'], bp['dispatches_per_step'], bp['kernel_ms_per_step'], bp['kernel_ms_top'])"
