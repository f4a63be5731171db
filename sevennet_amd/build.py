"""Ahead-of-time build of libsnet_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m sevennet_amd.build [-j N] [--force]

Generates the spherical-harmonics header and one .hip per tensor-product shape
(codegen.py), compiles every translation unit with
`hipcc --offload-arch=gfx950 -O3` and links `sevennet_amd/libsnet_hip.so`
in-tree (so it travels with the repo snapshot to the GPU box).
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

from . import codegen, codegen_fused
from .shapes import aot_conv_specs

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.environ.get('SNET_BUILD_LIB') or os.path.join(HERE, 'libsnet_hip.so')
# experiment builds (SNET_BUILD_LIB=... SNET_CODEGEN_OPTS=...) keep their sources/objects apart
_sfx = ('_' + os.path.basename(LIB).replace('.so', '')) if os.environ.get('SNET_BUILD_LIB') else ''
GEN = os.path.join(CSRC, 'generated' + _sfx)
OBJ = os.path.join(CSRC, 'build' + _sfx)
ARCH = 'gfx950'
STATIC_SOURCES = ['snet_api.cpp', 'snet_model.cpp', 'snet_halo.cpp', 'snet_gemm.hip', 'snet_mlp.hip', 'snet_edge.hip', 'snet_node.hip', 'snet_force.hip', 'snet_neighbor.hip', 'snet_md.hip', 'snet_d3.hip', 'snet_d3_ref.cpp']


def _rocm_root() -> str:
    """ROCM_PATH, else the prefix of the hipcc on PATH, else /opt/rocm"""
    env = os.environ.get('ROCM_PATH')
    if env and os.path.isdir(env):
        return env
    exe = shutil.which('hipcc')
    if exe:
        return os.path.dirname(os.path.dirname(os.path.realpath(exe)))
    return '/opt/rocm'


def _hipcc() -> str:
    exe = shutil.which('hipcc') or os.path.join(_rocm_root(), 'bin', 'hipcc')
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; the HIP force engine cannot be built')
    return exe


def _flags() -> List[str]:
    # -fno-slp-vectorize: SLP packing into v_pk_*_f32 duplicates operands into register pairs and
    # doubled the VGPR count of the tensor-product kernels (217 -> 115), halving their occupancy
    # SNET_BUILD_DEFS="-DSNET_GEMM_OCC=3 ...": extra definitions of an experiment build (tools/gpu/ab_bench.sh)
    return [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fno-slp-vectorize',
            f'-I{INCLUDE}', f'-I{CSRC}'] + os.environ.get('SNET_BUILD_DEFS', '').split()


def _stamp(src: str) -> str:
    h = hashlib.sha1()
    for p in [src, os.path.join(CSRC, 'snet_common.h'), os.path.join(CSRC, 'snet_split.h'), os.path.join(INCLUDE, 'snet_hip.h'),
              os.path.join(CSRC, 'generated', 'sh_generated.h')]:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(_flags()).encode())
    return h.hexdigest()


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src).rsplit('.', 1)[0] + '.o')
    stamp_file = obj + '.stamp'
    stamp = _stamp(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    cmd = [_hipcc()] + _flags() + ['-x', 'hip', '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stderr[-4000:]}')
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    return obj


def write_d3_blob() -> str:
    """data/d3_params.bin next to the library: the D3 tables and functional parameters of data/d3_params.npz in the flat
    layout csrc/snet_d3_ref.cpp reads (the reference's `pair_*` ABI has no table arguments: its library has them compiled in)"""
    import struct

    import numpy as np
    src = os.path.join(HERE, 'data', 'd3_params.npz')
    dst = os.path.join(os.path.dirname(LIB), 'data', 'd3_params.bin')
    if os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
        return dst
    z = np.load(src)
    out = [b'SNETD3P1', struct.pack('<q', z['c6ab'].shape[0])]
    for k in ('r0ab', 'c6ab', 'r2r4', 'rcov'):
        out.append(np.ascontiguousarray(z[k], dtype='<f8').tobytes())
    sets = [(0, 'damp_zero'), (1, 'damp_bj')]
    out.append(struct.pack('<i', len(sets)))
    for code, name in sets:
        names, pars = z[name + '_names'].tolist(), np.asarray(z[name + '_params'], np.float64)
        out.append(struct.pack('<ii', code, len(names)))
        for n, p in zip(names, pars):
            b = n.encode()
            assert len(b) < 32
            out.append(b.ljust(32, b'\0') + np.ascontiguousarray(p, dtype='<f8').tobytes())
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, 'wb') as f:
        f.write(b''.join(out))
    return dst


def build_lammps_harness(verbose: bool = True) -> str:
    """tests/lammps_mock/run_pair: the LAMMPS pair styles of lammps/*.cpp compiled by g++ against the runnable single-rank mock of
    the LAMMPS API (tests/lammps_mock/) and linked to libsnet_hip.so -- test scaffolding (tests/test_lammps_glue_gpu.py runs
    `pair_coeff -> init_style -> compute` through it on the GPU box; the binary travels in-tree like the library)."""
    root = os.path.dirname(HERE)
    mock = os.path.join(root, 'tests', 'lammps_mock')
    out = os.path.join(mock, 'run_pair')
    srcs = [os.path.join(mock, 'run_pair.cpp'), os.path.join(mock, 'lmp_mock_runtime.cpp'),
            os.path.join(root, 'lammps', 'pair_e3gnn_hip.cpp'), os.path.join(root, 'lammps', 'pair_d3_hip.cpp')]
    deps = srcs + [LIB] + [os.path.join(mock, f) for f in os.listdir(mock) if f.endswith('.h')] + \
        [os.path.join(root, 'lammps', f) for f in os.listdir(os.path.join(root, 'lammps')) if f.endswith('.h')] + \
        [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    gxx = shutil.which('g++')
    if gxx is None:
        raise RuntimeError('g++ not found: the LAMMPS mock harness cannot be built')
    rocm = _rocm_root()
    cmd = [gxx, '-std=c++17', '-O1', '-D__HIP_PLATFORM_AMD__', f'-I{mock}', f'-I{INCLUDE}', f'-I{rocm}/include'] + srcs + \
        ['-o', out, f'-L{os.path.dirname(LIB)}', '-l:' + os.path.basename(LIB), f'-L{rocm}/lib', '-lamdhip64',
         '-Wl,-rpath,$ORIGIN/../../sevennet_amd', f'-Wl,-rpath,{rocm}/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'g++ failed on the LAMMPS mock harness:\n{r.stderr[-4000:]}')
    if verbose:
        print(f'[sevennet_amd.build] built {out}', flush=True)
    return out


def build(jobs: int = 0, force: bool = False, extra_configs=(), verbose: bool = True) -> str:
    write_d3_blob()
    os.makedirs(GEN, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.join(CSRC, 'generated'), exist_ok=True)
    codegen.write_if_changed(os.path.join(CSRC, 'generated', 'sh_generated.h'), codegen.gen_sh_header(3))
    specs = aot_conv_specs(extra_configs)
    sources = [os.path.join(CSRC, s) for s in STATIC_SOURCES]
    keep = set()
    for tag, spec in specs.items():
        path = os.path.join(GEN, f'conv_{tag}.hip')
        codegen.write_if_changed(path, codegen.gen_conv(spec))
        sources.append(path)
        keep.add(os.path.basename(path))
        if codegen_fused.fusable(spec):  # fused radial-weight + tensor-product kernels: their own TU
            fpath = os.path.join(GEN, f'convf_{tag}.hip')
            codegen.write_if_changed(fpath, codegen_fused.gen_conv_fused(spec))
            sources.append(fpath)
            keep.add(os.path.basename(fpath))
    for f in os.listdir(GEN):  # drop stale generated shapes
        if (f.startswith('conv_') or f.startswith('convf_') or f.startswith('_conv')) and f not in keep:
            os.remove(os.path.join(GEN, f))
    jobs = jobs or min(16, os.cpu_count() or 4)
    if verbose:
        print(f'[sevennet_amd.build] {len(sources)} translation units ({len(specs)} conv shapes), -j{jobs}',
              flush=True)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), sources))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
    if verbose:
        print(f'[sevennet_amd.build] built {LIB}', flush=True)
    if not os.environ.get('SNET_BUILD_LIB'):   # (experiment libraries do not rebuild the harness)
        try:
            build_lammps_harness(verbose)
        except Exception as exc:  # noqa: BLE001  -- test scaffolding must not fail the product build (its tests then say so themselves)
            print(f'[sevennet_amd.build] LAMMPS mock harness NOT built: {exc}', file=sys.stderr, flush=True)
    return LIB


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-j', type=int, default=0)
    ap.add_argument('--force', action='store_true')
    a = ap.parse_args(argv)
    build(a.j, a.force)


if __name__ == '__main__':
    main()
