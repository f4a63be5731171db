#!/bin/bash
# the -m gpu suite and the smoke test, exit codes un-maskable:  gpurun --timeout 1800 -- bash tools/gpu/tests_only.sh
# (no `| tail` on either command: a pipe would report tail's exit status and drop the traceback)
set -o pipefail
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_full.log 2>&1
prc=$?
echo "pytest rc=$prc" | tee gpurun_out/tests.log
grep -E "passed|failed|error" gpurun_out/tests_full.log | tail -3 | tee -a gpurun_out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_full.log 2>&1
src=$?
cat gpurun_out/smoke_full.log | tee -a gpurun_out/tests.log
echo "smoke rc=$src" | tee -a gpurun_out/tests.log
[ $prc -eq 0 ] && [ $src -eq 0 ]
