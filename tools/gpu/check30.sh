#!/bin/bash
V="4,0,3;4,0,2;8,0,2;8,1,2;4,1,3;12,0,3;12,1,3;8,1,4"
echo "== layer 4 (005c575f8ec2)"
SNET_HIP_LIB=$PWD/exp/libx_l4.so timeout 400 python tools/microbench.py --terms 2 --iters 3 --layer 4 --only fused[ --fv "$V" 2>&1 | grep -i "fused\|error"
echo "== layer 0 (ecc5d202727d)"
SNET_HIP_LIB=$PWD/exp/libx_l0.so timeout 400 python tools/microbench.py --terms 2 --iters 3 --layer 0 --only fused --fv "$V" 2>&1 | grep -i "fused\|error"
