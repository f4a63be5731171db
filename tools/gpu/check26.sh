#!/bin/bash
for lib in base gpf gw; do
echo "== $lib"
SNET_HIP_LIB=$PWD/exp/libx_$lib.so timeout 300 python tools/microbench.py --terms 2 --iters 3 --only conv_bwd_fused[ --fv "4,0,3;4,0,2;8,0,2" 2>&1 | grep -i "fused\|error" | grep -v no_gxe
done | tee gpurun_out/g26_pf.log
