#!/bin/bash
timeout 900 python tools/gpu/diag_mf3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g22_diag.log | tail -60
