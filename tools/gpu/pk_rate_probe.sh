#!/bin/bash
[ -x tools/gpu/pkrate/pkrate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/gpu/pkrate/pkrate.hip -o tools/gpu/pkrate/pkrate
./tools/gpu/pkrate/pkrate 2>&1 | tee gpurun_out/pk_rate_probe.log
