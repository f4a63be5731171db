#!/bin/bash
# wave-uniform index (tree) against the round-5 kernels and a parity subset
bash tools/gpu/ab_kernels.sh exp/libx_base.so "sevennet_0:1" 2>&1 | grep "fused\["
bash tools/gpu/ab_step.sh base 2>&1 | tail -4
timeout 1200 python -m pytest tests -x -q -m gpu -k "fused or conv or engine or smoke or md_ or native" > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tests_b.log
