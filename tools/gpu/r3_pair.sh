#!/bin/bash
# two-channel-tile blocks in the reverse kernel + vectorized segment sums: correctness, then A/B (exp/libx_head.so = same tree
# built with SNET_CODEGEN_OPTS=nopairct=1)
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_native_model_gpu.py -x -q -k "fused or md_scale or small_cell or native_model_vs_engine or segment_sum or amorphous or (lmax3_shapes and 29) or halo_kernels" 2>&1 | tail -4 | tee gpurun_out/r3_pair_tests.log
bash tools/gpu/ab_bench.sh
for m in sevennet_l3i5 sevennet_mf_ompa; do
for v in head new; do
if [ $v = head ]; then export SNET_HIP_LIB=$PWD/exp/libx_head.so; else unset SNET_HIP_LIB; fi
timeout 300 python bench.py --no-cpu-baseline --model $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$m $v', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3))"
done; done 2>&1 | tee -a gpurun_out/ab_bench.log
