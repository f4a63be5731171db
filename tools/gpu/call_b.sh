#!/bin/bash
# round 6: parity of the dieted fused kernels, then A/B against the pre-diet library (exp/libx_base.so) inside one call
set -o pipefail
timeout 1500 python -m pytest tests -x -q -m gpu -k "fused or conv or engine or smoke or md_scale or native or tiled or amorphous" > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/tests_b.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/gpu/ab_kernels.sh exp/libx_base.so "sevennet_0:1 sevennet_0:0 sevennet_0:4 sevennet_l3i5:1" 2>&1 | tail -40
bash tools/gpu/ab_step.sh base 2>&1 | tail -6
