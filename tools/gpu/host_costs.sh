#!/bin/bash
# host-to-host costs at the benchmark size: native sequencer, ASE-style calculator, LAMMPS-style MD host
timeout 300 python bench.py --no-cpu-baseline --host native 2>/dev/null | tee gpurun_out/bench_native_host.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('native host', d['ms_per_step'], d['value'])"
timeout 600 python tools/calculator_cost.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/calculator_cost.log
timeout 600 python tools/md_host_cost.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/md_host_cost.log
