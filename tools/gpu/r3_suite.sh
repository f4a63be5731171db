#!/bin/bash
# full -m gpu suite + smoke, then SQ counters of the fused kernels (f16x3, v2 reverse body)
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_full.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r3_tests.log
grep -E "passed|failed|error" gpurun_out/r3_tests_full.log | tail -3 | tee -a gpurun_out/r3_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r3_tests.log
timeout 900 bash tools/gpu/sq_counters.sh > gpurun_out/r3_sq.log 2>&1; tail -30 gpurun_out/sq/summary.txt
