#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/g22_tests.log
timeout 300 python bench.py --no-cpu-baseline 2> gpurun_out/g22_bench.err | tee gpurun_out/g22_bench.json | cut -c1-300
