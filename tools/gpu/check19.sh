#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "lmax3_shapes" > gpurun_out/g19_tests.log 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/g19_tests.log | tail -8
