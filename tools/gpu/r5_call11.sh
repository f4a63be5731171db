#!/bin/bash
# lmax-3 forward kernels with the output-offset table in LDS: parity subset + bench lines of the two lmax-3 models
cd $GRAFT_REPO_ROOT
{ timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py tests/test_native_model_gpu.py -q -k "l3i5 or ompa or lmax3 or amorphous or config4 or l3" -rf 2>&1 | grep -E "passed|failed|FAILED"
for m in sevennet_l3i5 sevennet_mf_ompa; do
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null > gpurun_out/r05b_bench_${m}_n1.json; echo "bench $m rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r05b_bench_${m}_n1.json').read()); r=d['roofline']; k=r['kernel_ms_per_step']
print('$m', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), round(r['frac'],3), {a: round(b,2) for a,b in sorted(k.items(), key=lambda kv:-kv[1])[:6]})"
done; } > gpurun_out/r5_call11.log 2>&1
cat gpurun_out/r5_call11.log
