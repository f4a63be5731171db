#!/bin/bash
# launch-configuration / prefetch variants of the fused kernels (kernel-tuning builds under exp/)
for lib in exp gpf wf; do
  [ -f exp/libx_$lib.so ] || continue
  export SNET_HIP_LIB=$PWD/exp/libx_$lib.so
  echo "== $lib"
  timeout 600 python tools/microbench.py --terms 4 --iters 7 --only "conv_bwd_fused[22" --fv "4,0,3;4,0,2;8,0,2;4,0,3;4,0,2" 2>&1 | grep "fv="
  timeout 600 python tools/microbench.py --terms 2 --iters 7 --only "conv_bwd_fused[22" --fv "4,0,3;4,0,2" 2>&1 | grep "fv=" | sed 's/^/bf16x3 /'
done 2>&1 | tee gpurun_out/r3_variants.log
export SNET_HIP_LIB=$PWD/exp/libx_exp.so
timeout 600 python tools/microbench.py --terms 4 --iters 7 --only "conv_fwd_fused[22" --fv "12,1,3;12,0,3;8,0,2;8,1,2;12,1,3;12,0,3" 2>&1 | grep "fv=" | tee -a gpurun_out/r3_variants.log
unset SNET_HIP_LIB
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "edge_vectors" 2>&1 | tail -2 | tee -a gpurun_out/r3_variants.log
for h in positions edges; do
timeout 300 python bench.py --no-cpu-baseline --h2d $h 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('h2d $h step', round(d['ms_per_step'],2), r['kernel'], round(r['avg_ms'],3), d['config']['h2d_in_step'][:60])"
done | tee -a gpurun_out/r3_variants.log
