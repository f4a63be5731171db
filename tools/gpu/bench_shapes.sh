#!/bin/bash
# bench lines of the three shapes in one call (clock state of the box shows in the VALU-bound hidden-MLP kernel)
TAG=${1:-r03}
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for m in sevennet_0 sevennet_l3i5 sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | tee gpurun_out/${TAG}_bench_${m}_n1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernel_ms_per_step']
print('$m', round(d['ms_per_step'],2), 'dominant', round(r['avg_ms'],3), r['frac'], 'hidden', k.get('radial_mlp_hidden_fwd'))"
done 2>&1 | tee gpurun_out/bench_shapes.log
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
