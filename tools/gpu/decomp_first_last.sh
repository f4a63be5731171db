#!/bin/bash
# timing of workgroup shapes / decomposition of the first layer's reverse kernel with the tuning build (SNET_CODEGEN_OPTS=fexp=<tag>):
#   exp/libx_fe_ecc5.so;  variants "waves,glds,occ[,diag]" (diag bits: codegen_fused.py)
export SNET_HIP_LIB=$PWD/exp/libx_fe_ecc5.so
timeout 600 python tools/microbench.py --layer 0 --terms 4 --iters 7 --only "conv_bwd_fused_no_gxe" \
    --fv "${FV:-8,0,2;8,0,2;8,0,4;8,0,3;8,0,2;8,0,4;8,0,3;8,1,4}" 2>&1 | grep -E "^conv_" | sed "s/^/L0  /" | tee gpurun_out/decomp_first_last.log
unset SNET_HIP_LIB
