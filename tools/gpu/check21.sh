#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/g21_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/g21_smoke.log
