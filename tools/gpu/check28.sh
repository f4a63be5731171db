#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids > gpurun_out/g28_tests_full.log; grep -E "passed|failed|error" gpurun_out/g28_tests_full.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/g28_smoke.log
timeout 1500 bash tools/collect_profiles.sh r02 2>&1 | tail -3
