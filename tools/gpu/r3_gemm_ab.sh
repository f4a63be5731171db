#!/bin/bash
# split-precision grouped GEMM variants (exp/libx_*.so built with SNET_BUILD_DEFS) on the middle layer's linears
for v in base occ3 pipe3 pipe2; do
if [ $v = base ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
echo "== $v"
timeout 300 python tools/microbench.py --only si --iters 7 2>/dev/null | grep -v "^lib="
timeout 300 python tools/microbench.py --only sc_ --iters 7 2>/dev/null | grep -v "^lib="
done 2>&1 | tee gpurun_out/r3_gemm_ab.log
