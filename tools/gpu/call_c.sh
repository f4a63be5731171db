#!/bin/bash
timeout 900 python tools/gpu/dbg_md3.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 900 python -m pytest tests/test_md_host_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/md_loop.py --oracle-steps 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/md_loop_f16x3.txt
SNET_FUSED_TERMS=2 timeout 600 python tools/md_loop.py --oracle-steps 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/md_loop_bf16x3.txt
timeout 600 python tools/md_loop.py --oracle-steps 0 --dt 1.0 --steps 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/md_loop_dt1.txt
timeout 300 python tools/md_host_cost.py > gpurun_out/r06_md_host_cost.txt 2>&1; tail -3 gpurun_out/r06_md_host_cost.txt
