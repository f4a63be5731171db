#!/bin/bash
# the -m gpu suite with its exit code and the smoke test:  gpurun --timeout 1800 -- bash tools/gpu/tests_only.sh
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_full.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/tests.log
grep -E "passed|failed|error" gpurun_out/tests_full.log | tail -3 | tee -a gpurun_out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/tests.log
