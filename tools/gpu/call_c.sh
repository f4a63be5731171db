#!/bin/bash
timeout 600 python tools/gpu/dbg_gx.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
timeout 1500 python -m pytest tests -x -q -m gpu -k "fused or conv or engine or smoke or md_scale or native or tiled or amorphous" > gpurun_out/tests_b.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/tests_b.log
bash tools/gpu/ab_kernels.sh exp/libx_base.so "sevennet_0:1" 2>&1 | grep "fused\[" | tail -12
bash tools/gpu/ab_step.sh base 2>&1 | tail -4
