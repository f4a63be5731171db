#!/bin/bash
timeout 1200 python -m pytest tests -x -q -m gpu -k "fused or conv or engine or smoke or native" > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tests_b.log
bash tools/gpu/ab_kernels.sh exp/libx_base.so "sevennet_0:1 sevennet_0:0 sevennet_0:4 sevennet_l3i5:1" 2>&1 | grep "fwd_fused\["
bash tools/gpu/ab_step.sh base 2>&1 | tail -4
