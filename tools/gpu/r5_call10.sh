#!/bin/bash
# lmax-3 forward kernels (not power-throttled: profiles/r05_power_probe.txt): do the schedule variants pay there?
cd $GRAFT_REPO_ROOT
{ MODEL=sevennet_l3i5 LAYER=1 bash tools/gpu/ab_libs.sh "conv_fwd_fused" "$@" | sed "s/^/l3i5 L1 /"
  MODEL=sevennet_mf_ompa LAYER=1 bash tools/gpu/ab_libs.sh "conv_fwd_fused" "$@" | sed "s/^/ompa L1 /"; } > gpurun_out/r5_call10.log 2>&1
cat gpurun_out/r5_call10.log
