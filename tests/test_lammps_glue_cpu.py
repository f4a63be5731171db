"""CPU: the LAMMPS glue (lammps/pair_e3gnn_hip.{h,cpp}) parses and type-checks against a MOCK of the LAMMPS
API subset it uses (tests/lammps_mock/ -- not LAMMPS; no LAMMPS tree or MPI exists in the development image),
and the patch script does to a LAMMPS-shaped tree what the reference's patch_lammps.sh:74-134 does to a real
one (backup, copy sources, append to cmake/CMakeLists.txt, refuse a second run)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime.h'),
                    reason='needs g++ and the HIP headers')
def test_pair_style_sources_type_check_against_mock_lammps():
    r = subprocess.run(['g++', '-std=c++17', '-fsyntax-only', '-Wall', '-D__HIP_PLATFORM_AMD__',
                        '-I', os.path.join(ROOT, 'tests', 'lammps_mock'), '-I', os.path.join(ROOT, 'include'),
                        '-I', '/opt/rocm/include', os.path.join(ROOT, 'lammps', 'pair_e3gnn_hip.cpp')],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    own = [ln for ln in r.stderr.splitlines() if 'pair_e3gnn_hip' in ln and ('warning' in ln or 'error' in ln)]
    assert own == [], own
    # the style registration block the LAMMPS build scans for
    h = open(os.path.join(ROOT, 'lammps', 'pair_e3gnn_hip.h')).read()
    assert 'PairStyle(e3gnn, PairE3GNNHip)' in h and 'PairStyle(e3gnn/parallel, PairE3GNNHipParallel)' in h
    c = open(os.path.join(ROOT, 'lammps', 'pair_e3gnn_hip.cpp')).read()
    # every C-ABI entry point the glue calls is declared in include/snet_hip.h
    import re
    api = open(os.path.join(ROOT, 'include', 'snet_hip.h')).read()
    for fn in sorted(set(re.findall(r'\b(snet_[a-z0-9_]+)\(', c))):
        assert re.search(r'\b' + fn + r'\(', api), fn


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_pair_style_d3_type_checks_against_mock_lammps():
    """lammps/pair_d3_hip.{h,cpp}: the reference's `pair_style d3` grammar (pair_d3.cu:261-285,644-656) over the library's
    `pair_*` D3 binding -- parses and type-checks against the mock; every library function it calls is declared"""
    import re
    r = subprocess.run(['g++', '-std=c++17', '-fsyntax-only', '-Wall', '-I', os.path.join(ROOT, 'tests', 'lammps_mock'),
                        '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'lammps', 'pair_d3_hip.cpp')],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert [ln for ln in r.stderr.splitlines() if 'pair_d3_hip' in ln and ('warning' in ln or 'error' in ln)] == []
    assert 'PairStyle(d3, PairD3Hip)' in open(os.path.join(ROOT, 'lammps', 'pair_d3_hip.h')).read()
    c = open(os.path.join(ROOT, 'lammps', 'pair_d3_hip.cpp')).read()
    api = open(os.path.join(ROOT, 'include', 'snet_d3_ref.h')).read() + open(os.path.join(ROOT, 'include', 'snet_hip.h')).read()
    for fn in sorted(set(re.findall(r'\b(pair_(?:init|set_atom|set_domain|run_settings|run_coeff|run_compute|get_energy|get_force|get_stress|fin)|snet_[a-z0-9_]+)\(', c))):
        assert re.search(r'\b' + fn + r'\(', api), fn


def test_patch_script_on_a_lammps_shaped_tree(tmp_path):
    lmp = tmp_path / 'lammps'
    (lmp / 'cmake').mkdir(parents=True)
    (lmp / 'src').mkdir()
    (lmp / 'cmake' / 'CMakeLists.txt').write_text('project(lammps)\nset(CMAKE_CXX_STANDARD 11)\nadd_library(lammps)\n')
    libdir = tmp_path / 'lib'
    libdir.mkdir()
    (libdir / 'libsnet_hip.so').write_bytes(b'')
    script = os.path.join(ROOT, 'lammps', 'patch_lammps_hip.sh')
    r = subprocess.run(['bash', script, str(lmp), str(libdir)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for f in ('pair_e3gnn_hip.h', 'pair_e3gnn_hip.cpp', 'pair_d3_hip.h', 'pair_d3_hip.cpp', 'snet_hip.h', 'snet_d3_ref.h'):
        assert (lmp / 'src' / f).exists(), f
    cm = (lmp / 'cmake' / 'CMakeLists.txt').read_text()
    assert 'set(CMAKE_CXX_STANDARD 17)' in cm and 'libsnet_hip.so' in cm and 'find_package(hip REQUIRED)' in cm
    assert 'Torch' not in cm and 'comm_brick' not in cm          # neither LibTorch nor a comm patch
    assert (lmp / '_backups' / 'CMakeLists.txt').read_text().count('CMAKE_CXX_STANDARD 11') == 1
    again = subprocess.run(['bash', script, str(lmp), str(libdir)], capture_output=True, text=True)
    assert again.returncode != 0 and 'already patched' in again.stdout
    bad = subprocess.run(['bash', script, str(tmp_path / 'nope')], capture_output=True, text=True)
    assert bad.returncode != 0


@pytest.mark.skipif(shutil.which('g++') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime.h'),
                    reason='needs g++ and the HIP headers')
def test_runnable_lammps_mock_builds_links_and_fails_cleanly_without_a_gpu(tmp_path):
    """the harness of tests/test_lammps_glue_gpu.py (round 5): g++ compiles the real pair-style sources against the runnable mock and
    links them to libsnet_hip.so here, without a GPU; the driver's error paths that need no device work: usage, unknown style,
    unreadable structure, and -- on a box without a ROCm device -- the pair style's own refusal instead of a crash"""
    from sevennet_amd import _lib
    from sevennet_amd.build import build, build_lammps_harness
    if not os.path.exists(_lib.LIB_PATH):
        build(verbose=False)
    exe = build_lammps_harness(verbose=False)
    assert os.path.exists(exe) and os.access(exe, os.X_OK)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and 'usage: run_pair' in r.stderr
    struct = tmp_path / 's.txt'
    struct.write_text('2 1\n4 0 0\n0 4 0\n0 0 4\n0.5\n1 0 0 0\n1 1.5 0 0\n')
    out = str(tmp_path / 'o.json')
    r = subprocess.run([exe, 'lj/cut', str(struct), out, '--', '*', '*'], capture_output=True, text=True)
    assert r.returncode == 3 and 'unknown pair style' in r.stderr
    r = subprocess.run([exe, 'd3', str(tmp_path / 'missing.txt'), out, '9000', '1600', 'damp_bj', 'pbe', '--', '*', '*', 'H'],
                       capture_output=True, text=True)
    assert r.returncode == 3 and 'cannot open' in r.stderr
    # pair_style d3 validates its arguments before it touches the device (pair_d3.cu:261-285)
    r = subprocess.run([exe, 'd3', str(struct), out, '9000', '1600', '--', '*', '*', 'H'], capture_output=True, text=True)
    assert r.returncode == 3 and 'needs Four arguments' in r.stderr
    r = subprocess.run([exe, 'd3', str(struct), out, '9000', '1600', 'damp_foo', 'pbe', '--', '*', '*', 'H'], capture_output=True, text=True)
    assert r.returncode == 3 and 'Unknown damping' in r.stderr
    r = subprocess.run([exe, 'd3', str(struct), out, '9000', '1600', 'damp_bj', 'pbe', '--', '*', '*', 'Xx'], capture_output=True, text=True)
    assert r.returncode == 3 and 'unknown element' in r.stderr
    import torch
    if not torch.cuda.is_available():   # no device: the e3gnn constructor refuses (the engine has no CPU path), nothing segfaults
        r = subprocess.run([exe, 'e3gnn', str(struct), out, '--', '*', '*', 'nope.snet', 'H'], capture_output=True, text=True)
        assert r.returncode == 3 and 'no ROCm device' in r.stderr
        r = subprocess.run([exe, 'd3', str(struct), out, '9000', '1600', 'damp_bj', 'pbe', '--', '*', '*', 'H'], capture_output=True, text=True)
        assert r.returncode == 3 and 'pair_style d3' in r.stderr
