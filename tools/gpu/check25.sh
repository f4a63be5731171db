#!/bin/bash
./tools/gpu/pkrate/pkrate 2>&1 | tee gpurun_out/g25_pkrate.log
