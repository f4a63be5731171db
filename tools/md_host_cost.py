#!/usr/bin/env python3
"""Wall time of one snet_md_compute call (what a LAMMPS pair style pays per step) at the benchmark size:
host flattening of the neighbor rows + H2D + GPU filter / graph build + model + D2H, all inclusive.

    python tools/md_host_cost.py [--reps 23]
"""
import argparse
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=23)
    ap.add_argument('--skin', type=float, default=1.0)
    a = ap.parse_args()
    from test_md_host_gpu import MdHost
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, 0)
    pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.05, 2)
    n, rc = len(pos), cfg['cutoff'] + a.skin
    # one process holding the whole periodic cell: ghosts = images within rc of the box
    L = np.diag(cell)
    xs, tags = [pos], [np.arange(1, n + 1)]
    for sx in (-1, 0, 1):
        for sy in (-1, 0, 1):
            for sz in (-1, 0, 1):
                if (sx, sy, sz) == (0, 0, 0):
                    continue
                img = pos + np.array([sx, sy, sz]) * L
                sel = np.all((img > -rc) & (img < L + rc), axis=1)
                xs.append(img[sel])
                tags.append(np.nonzero(sel)[0] + 1)
    x, tag = np.concatenate(xs), np.concatenate(tags)
    t0 = time.time()
    tree = cKDTree(x)
    nb = tree.query_ball_point(x[:n], rc, workers=-1)
    rows = [np.asarray([j for j in r if j != i], np.int32) for i, r in enumerate(nb)]
    print(f'{n} local + {len(x) - n} ghost atoms, {sum(len(r) for r in rows)} neighbor slots (skin {a.skin} A); '
          f'list built on the host in {time.time() - t0:.1f} s')
    import ctypes as C
    import torch
    from sevennet_amd import _lib
    host = MdHost(cfg, sd)
    lib = host.lib
    nall = len(x)
    ilist = np.arange(n, dtype=np.int32)
    numneigh = np.zeros(nall, np.int32)
    first = (C.c_void_p * nall)()
    for i, r in enumerate(rows):   # what LAMMPS hands over: per-atom row pointers, marshalled ONCE here
        numneigh[i] = len(r)
        first[i] = r.ctypes.data
    xx = np.ascontiguousarray(x, np.float64)
    tg = np.ascontiguousarray(tag, np.int32)
    ty = np.ones(nall, np.int32)
    tmap = np.array([-1, 0], np.int32)
    f = np.zeros((nall, 3))
    eng, vir = C.c_double(0.0), np.zeros(6)
    nn, ne = C.c_int64(), C.c_int64()
    P = lambda arr: C.c_void_p(arr.ctypes.data)  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(unchanged=False):
        f[:] = 0.0
        if unchanged:   # what the pair style says between two LAMMPS list rebuilds (neighbor->ago > 0)
            _lib.check(lib.snet_md_list_unchanged(host.h), 'snet_md_list_unchanged')
        _lib.check(lib.snet_md_compute(host.h, n, P(ilist), P(numneigh), C.cast(first, C.c_void_p), nall, P(xx), P(ty), P(tg), 4,
                                       P(tmap), 1, 0, 0, 1, 0, P(f), C.cast(C.byref(eng), C.c_void_p), P(vir), None, None,
                                       None, C.byref(nn), C.byref(ne), st), 'snet_md_compute')

    call()   # warm-up (arena, plans)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    tu = []
    for _ in range(5):
        t0 = time.perf_counter()
        call(True)
        tu.append(time.perf_counter() - t0)
    print(f'edges inside the cutoff: {ne.value}; max |f| {np.abs(f).max():.4f}')
    print(f'snet_md_compute per call BETWEEN list rebuilds (snet_md_list_unchanged: positions only travel): median '
          f'{np.median(tu) * 1e3:.1f} ms, min {min(tu) * 1e3:.1f} ms')
    print(f'snet_md_compute per call (host flatten + H2D + GPU graph build + model + D2H): median {np.median(ts) * 1e3:.1f} ms, '
          f'min {min(ts) * 1e3:.1f} ms')


if __name__ == '__main__':
    main()
