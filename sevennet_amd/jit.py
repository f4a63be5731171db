"""On-demand compilation of tensor-product shapes.

The reference's accelerator hook constructs `convolution_cls(**kwargs)` for ANY irreps
(sevenn/nn/convolution.py:237-247); its flash / cuEq / OpenEquivariance back ends JIT their kernels.
libsnet_hip.so carries the shapes of sevennet_amd/shapes.py ahead of time; any other shape is generated
(codegen.py / codegen_fused.py), compiled with hipcc for gfx950 into a small shared library under the
cache directory, and registered with the running library:

    ensure_conv_shape(spec)  ->  tag      (no-op when the shape is already registered)

    $SNET_JIT_CACHE (default ~/.cache/sevennet_amd/jit)/<tag>_<source hash>.so

The shape library holds only the generated kernels and their static registrars; it links against
libsnet_hip.so, and `snet_conv_register_library(path)` (C-ABI, a dlopen) runs the registrars, after which
`snet_conv_plan_create(tag)` / `snet_fused_plan_create` find the shape.  A C++ host calls the same entry
point with a library built offline by `python -m sevennet_amd.jit '<irreps_x>' '<irreps_sh>' '<irreps_out>'`
or by `sevennet_amd.build.build(extra_configs=[...])`.  There is no fallback: without hipcc the error says so.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import tempfile
import threading

from . import _lib, codegen, codegen_fused
from .build import ARCH, CSRC, INCLUDE, LIB, _flags, _hipcc
from .model_spec import ConvSpec

_lock = threading.Lock()


def cache_dir() -> str:
    d = os.environ.get('SNET_JIT_CACHE') or os.path.join(os.path.expanduser('~'), '.cache', 'sevennet_amd', 'jit')
    os.makedirs(d, exist_ok=True)
    return d


def shape_sources(spec: ConvSpec):
    """[(file name, HIP source)] of one shape: the separate kernels, plus the fused ones where they exist"""
    srcs = [(f'conv_{spec.tag}.hip', codegen.gen_conv(spec))]
    if codegen_fused.fusable(spec):
        srcs.append((f'convf_{spec.tag}.hip', codegen_fused.gen_conv_fused(spec)))
    return srcs


def compile_shape(spec: ConvSpec, out_dir: str = None, verbose: bool = False) -> str:
    """Generate + compile one shape into <out_dir>/<tag>_<hash>.so (returns the path; reuses a cached file)."""
    out_dir = out_dir or cache_dir()
    srcs = shape_sources(spec)
    h = hashlib.sha1()
    for _, text in srcs:
        h.update(text.encode())
    for hdr in ('snet_common.h', 'snet_split.h'):
        with open(os.path.join(CSRC, hdr), 'rb') as f:
            h.update(f.read())
    h.update(' '.join(_flags()).encode())
    path = os.path.join(out_dir, f'{spec.tag}_{h.hexdigest()[:12]}.so')
    if os.path.exists(path):
        return path
    os.makedirs(os.path.join(CSRC, 'generated'), exist_ok=True)
    codegen.write_if_changed(os.path.join(CSRC, 'generated', 'sh_generated.h'), codegen.gen_sh_header(3))
    with tempfile.TemporaryDirectory(dir=out_dir) as tmp:
        files = []
        for name, text in srcs:
            p = os.path.join(tmp, name)
            with open(p, 'w') as f:
                f.write(text)
            files.append(p)
        lib_dir = os.path.dirname(os.path.abspath(LIB))
        cmd = [_hipcc()] + _flags() + ['-shared', '-x', 'hip'] + files + \
              ['-x', 'none', f'-L{lib_dir}', f'-l:{os.path.basename(LIB)}', f'-Wl,-rpath,{lib_dir}', '-o', os.path.join(tmp, 'shape.so')]
        if verbose:
            print('[sevennet_amd.jit]', ' '.join(cmd), file=sys.stderr, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on tensor-product shape {spec.tag} ({spec.key}):\n{r.stderr[-4000:]}')
        os.replace(os.path.join(tmp, 'shape.so'), path)   # atomic: concurrent processes race harmlessly
    return path


def ensure_conv_shape(spec: ConvSpec, verbose: bool = False) -> str:
    """Make sure `spec`'s kernels are registered with libsnet_hip.so; compile + load them if they are not."""
    lib = _lib.load()
    tag = spec.tag
    with _lock:
        if tag in _lib.compiled_conv_tags():
            return tag
        path = compile_shape(spec, verbose=verbose)
        _lib.check(lib.snet_conv_register_library(path.encode()), 'snet_conv_register_library')
        if tag not in _lib.compiled_conv_tags():
            raise RuntimeError(f'{path} was loaded but did not register shape {tag}')
    return tag


def main(argv=None):
    import argparse
    from .conv_plugin import conv_spec_from_instructions
    from .irreps import Irreps
    ap = argparse.ArgumentParser(description='compile one uvu tensor-product shape into a shape library')
    ap.add_argument('irreps_x')
    ap.add_argument('irreps_sh')
    ap.add_argument('irreps_out', help='sorted irreps_mid, one block per instruction')
    ap.add_argument('--instructions', default=None,
                    help='"i,j,k;i,j,k;..." (default: every allowed (x, sh) -> out path in e3nn order, as '
                         'sevenn/nn/convolution.py:61-82 builds them)')
    ap.add_argument('--out-dir', default=None)
    a = ap.parse_args(argv)
    if a.instructions:
        ins = [tuple(int(v) for v in t.split(',')) + ('uvu', True) for t in a.instructions.split(';')]
    else:
        x, sh, mid = Irreps(a.irreps_x), Irreps(a.irreps_sh), Irreps(a.irreps_out)
        ins, used = [], set()
        for i, (mul, l1, p1) in enumerate(x):
            for j, (_, l2, p2) in enumerate(sh):
                for k, (mo, l3, p3) in enumerate(mid):
                    if k not in used and mo == mul and p3 == p1 * p2 and abs(l1 - l2) <= l3 <= l1 + l2:
                        ins.append((i, j, k, 'uvu', True))
                        used.add(k)
                        break
        ins.sort(key=lambda t: t[2])
    spec = conv_spec_from_instructions(a.irreps_x, a.irreps_sh, a.irreps_out, ins)
    print(compile_shape(spec, out_dir=a.out_dir, verbose=True))


if __name__ == '__main__':
    main()
