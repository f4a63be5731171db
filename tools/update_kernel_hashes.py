#!/usr/bin/env python3
"""Rewrite tests/golden/fused_kernel_sha1.json from the generator as it stands (run after a DELIBERATE change of the fused kernels;
tests/test_codegen_fused_cpu.py::test_generated_sources_are_the_committed_ones compares against it)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sevennet_amd import codegen_fused  # noqa: E402
from sevennet_amd.shapes import aot_conv_specs  # noqa: E402

out = {t: hashlib.sha1(codegen_fused.gen_conv_fused(sp).encode()).hexdigest()
       for t, sp in sorted(aot_conv_specs().items()) if codegen_fused.fusable(sp)}
path = os.path.join(ROOT, 'tests', 'golden', 'fused_kernel_sha1.json')
with open(path, 'w') as f:
    json.dump(out, f, indent=1)
print(f'{len(out)} shapes -> {path}')
