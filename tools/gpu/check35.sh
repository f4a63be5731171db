#!/bin/bash
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "loopback or native_halo or rccl" > gpurun_out/g35_tests_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rror" gpurun_out/g35_tests_full.log | tail -5
tail -30 gpurun_out/g35_tests_full.log | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | tail -25
