#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "fused_matches" 2>&1 | grep -E "^E |passed|failed|FAILED|Error" | head -40 | tee gpurun_out/r3_pair2_tests.log
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_native_model_gpu.py -q -k "md_scale or small_cell or native_model_vs_engine or fused_engine" 2>&1 | grep -E "^E |passed|failed|FAILED|Error" | head -40 | tee -a gpurun_out/r3_pair2_tests.log
