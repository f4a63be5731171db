#!/bin/bash
timeout 900 python tools/gpu/dbg_md3.py 2>&1 | grep -v amdgpu.ids
