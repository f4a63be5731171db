#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "rccl_halo or fused_engine or sevennet_0_shape" > gpurun_out/g17_tests.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/g17_tests.log | tail -5
