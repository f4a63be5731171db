#!/bin/bash
# round 5, call 4: persistent workgroups (persist=n) against the working tree: kernel timing, light stamps, parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for rep in 1 2; do
for v in tree pers4 pers8 pers16; do
  if [ $v = tree ]; then unset SNET_HIP_LIB; else export SNET_HIP_LIB=$PWD/exp/libx_$v.so; fi
  timeout 300 python tools/microbench.py --layer 1 --terms 4 --iters 7 --only "conv_bwd_fused" 2>&1 | grep -E "^conv_" | sed "s/^/$v  /"
done; done
echo "== light stamps, persist=8"
SNET_HIP_LIB=$PWD/exp/libx_pers8l.so timeout 300 python tools/microbench.py --layer 1 --terms 4 --iters 5 --only "conv_bwd_fused_tail" --stamps 2>&1 | grep -v "^lib=\|amdgpu.ids\|not built" | grep -vE " +0 cycles per wave"
echo "== parity"
for v in pers8 pers4; do SNET_HIP_LIB=$PWD/exp/libx_$v.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_fused_matches_separate_kernels or fused_convolution_module" 2>&1 | tail -2 | sed "s/^/$v  /"; done
} 2>&1 | tee gpurun_out/r5_call4.log
