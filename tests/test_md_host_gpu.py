"""GPU: snet_md_compute -- the LAMMPS-facing host (neighbor list in, forces accumulated out) --
against the fp64 CPU oracle on the same periodic system (energies, forces, LAMMPS-order virials of
pair_e3gnn.cpp:206-270: xx,yy,zz,xy,xz,yz = model components [0,1,2,3,5,4], :254-255).  The fake LAMMPS domain below has
what a pair style sees: owned atoms, ghost images with the owners' tags, a FULL neighbor list
built with a skin (so it holds pairs beyond the cutoff), special-bond bits, a permuted ilist."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

from helpers import synthetic_system

pytestmark = pytest.mark.gpu

SKIN = 1.0


def lammps_domain(pos, cell, owned, rc):
    """(x[nall,3], tag[nall], nlocal, neighbor rows) of the process that owns `owned` atoms."""
    pos, cell = np.asarray(pos, float), np.asarray(cell, float)
    n = len(pos)
    reach = [int(np.ceil(rc / (abs(np.linalg.det(cell)) / np.linalg.norm(np.cross(cell[(k + 1) % 3], cell[(k + 2) % 3]))))) for k in range(3)]
    loc = np.nonzero(owned)[0]
    xs, tags = [pos[loc]], [loc + 1]
    tree_loc = cKDTree(pos[loc])
    for sx in range(-reach[0], reach[0] + 1):
        for sy in range(-reach[1], reach[1] + 1):
            for sz in range(-reach[2], reach[2] + 1):
                img = pos + np.array([sx, sy, sz]) @ cell
                cand = np.ones(n, bool)
                if (sx, sy, sz) == (0, 0, 0):
                    cand[loc] = False
                d, _ = tree_loc.query(img, distance_upper_bound=rc)
                sel = cand & np.isfinite(d)
                xs.append(img[sel])
                tags.append(np.nonzero(sel)[0] + 1)
    x, tag = np.concatenate(xs), np.concatenate(tags)
    tree = cKDTree(x)
    rows = []
    for i in range(len(loc)):
        r = np.array([j for j in tree.query_ball_point(x[i], rc) if j != i], np.int32)
        rows.append(r)
    return x, tag, len(loc), rows


class MdHost:
    def __init__(self, cfg, sd):
        from sevennet_amd import _lib
        from sevennet_amd.native_model import NativeModel
        self.lib = _lib.load()
        self.model = NativeModel(cfg, sd)
        self.h = C.c_void_p()
        _lib.check(self.lib.snet_md_create(self.model.handle, C.byref(self.h)), 'snet_md_create')

    def __del__(self):
        self.lib.snet_md_destroy(self.h)

    def compute(self, x, tag, nlocal, rows, types, ghost_mode=0, ilist=None, tag_bytes=4, eflag_atom=1, vflag_atom=1,
                special_bits=False, unchanged=False):
        from sevennet_amd import _lib
        if unchanged:   # what the pair style does when neighbor->ago > 0
            _lib.check(self.lib.snet_md_list_unchanged(self.h), 'snet_md_list_unchanged')
        nall = len(x)
        ilist = np.arange(nlocal, dtype=np.int32) if ilist is None else np.asarray(ilist, np.int32)
        numneigh = np.zeros(nall, np.int32)
        rows = [np.ascontiguousarray(r, np.int32).copy() for r in rows]
        if special_bits:  # LAMMPS encodes special-bond flags in the top bits of a neighbor entry
            for r in rows[::3]:
                r[::2] |= (1 << 30)
        first = (C.c_void_p * nall)()
        for i, r in enumerate(rows):
            numneigh[i] = len(r)
            first[i] = r.ctypes.data
        x = np.ascontiguousarray(x, np.float64)
        tg = np.ascontiguousarray(tag, np.int32 if tag_bytes == 4 else np.int64)
        ty = np.ascontiguousarray(types, np.int32) + 1  # LAMMPS types are 1-based
        ntypes = int(ty.max())
        tmap = np.arange(-1, ntypes, dtype=np.int32)    # type t -> species t-1
        f = np.zeros((nall, 3)); eng = C.c_double(0.0); vir = np.zeros(6)
        eatom = np.zeros(nall); vatom = np.zeros((nall, 6))
        n2a = np.full(nall, -1, np.int32)
        nn, ne = C.c_int64(), C.c_int64()
        P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        rc = self.lib.snet_md_compute(self.h, len(ilist), P(ilist), P(numneigh), C.cast(first, C.c_void_p), nall, P(x), P(ty),
                                      P(tg), tag_bytes, P(tmap), ntypes, ghost_mode, eflag_atom, 1, vflag_atom, P(f),
                                      C.cast(C.byref(eng), C.c_void_p), P(vir), P(eatom), P(vatom) if vflag_atom else None,
                                      P(n2a), C.byref(nn), C.byref(ne), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, 'snet_md_compute')
        return dict(f=f, energy=eng.value, virial=vir, eatom=eatom, vatom=vatom, node_to_atom=n2a[:nn.value],
                    n_nodes=nn.value, n_edges=ne.value)


F_TOL = 1e-4  # eV/A, absolute (BASELINE.json north_star)


def _reference(cfg, sd, types, ei, ev):
    """fp64 oracle: OracleModel.forward + force_virial_from_edge (force_output.py:171-230 restated)"""
    from helpers import oracle_model
    out = oracle_model(cfg, sd).forward(np.asarray(types), np.asarray(ei), np.asarray(ev, np.float64))
    return {k: (v.detach().numpy().reshape(1) if k == 'energy' else v.detach().numpy()) for k, v in out.items()
            if k in ('energy', 'forces', 'atomic_energy', 'virial', 'atomic_virial')}


def _setup(case):
    from sevennet_amd.shapes import mini_sevennet_0_config, unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    if case == 'unit':
        cfg, cutoff, nsp, reps = unit_test_config(), 4.0, 4, (2, 2, 2)
    elif case == 'unit_tiny_cell':  # cell edge 5.43 < 2 x (cutoff + skin): several images of one atom per row
        cfg, cutoff, nsp, reps = unit_test_config(), 4.0, 4, (1, 1, 1)
    else:
        cfg, cutoff, nsp, reps = mini_sevennet_0_config(), 5.0, 2, (3, 3, 3)
    sd = random_state_dict(cfg, seed=5)
    types, pos, cell, ei, ev = synthetic_system(reps, sigma=0.07, seed=21, cutoff=cutoff, n_species=nsp)
    return cfg, sd, cutoff, types, pos, cell, ei, ev


@pytest.mark.parametrize('case', ['unit', 'unit_tiny_cell', 'mini'])
def test_md_host_serial_matches_engine(case):
    cfg, sd, cutoff, types, pos, cell, ei, ev = _setup(case)
    ref = _reference(cfg, sd, types, ei, ev)
    n = len(types)
    x, tag, nlocal, rows = lammps_domain(pos, cell, np.ones(n, bool), cutoff + SKIN)
    assert nlocal == n and len(x) > n
    ty_all = np.asarray(types)[tag - 1]
    host = MdHost(cfg, sd)
    perm = np.random.default_rng(1).permutation(n).astype(np.int32)
    for kw in (dict(), dict(ilist=perm, tag_bytes=8, special_bits=True)):
        out = host.compute(x, tag, nlocal, rows, ty_all, **kw)
        assert out['n_edges'] == ei.shape[1] and out['n_nodes'] == n
        fs = np.abs(ref['forces']).max()
        assert abs(out['energy'] - float(ref['energy'][0])) <= 2e-6 * abs(float(ref['energy'][0])) + 1e-6
        assert np.abs(out['f'][:n] - ref['forces']).max() <= min(F_TOL, max(1e-6, 3e-5 * fs))
        assert np.abs(out['f'][n:]).max() == 0.0        # ghosts are aliased, never written
        v = ref['virial'][[0, 1, 2, 3, 5, 4]]           # model xx yy zz xy yz zx -> LAMMPS xx yy zz xy xz yz
        assert np.abs(out['virial'] - v).max() <= max(1e-6, 3e-5 * np.abs(v).max())
        assert np.abs(out['eatom'][:n] - ref['atomic_energy']).max() <= max(5e-6, 3e-5 * np.abs(ref['atomic_energy']).max())
        va = ref['atomic_virial'][:, [0, 1, 2, 3, 5, 4]]
        assert np.abs(out['vatom'][:n] - va).max() <= max(1e-6, 3e-5 * np.abs(va).max())
    # results are ADDED to the host arrays, and a second call out of the same workspace is identical
    again = host.compute(x, tag, nlocal, rows, ty_all)
    first = host.compute(x, tag, nlocal, rows, ty_all)
    assert np.array_equal(again['f'], first['f']) and again['energy'] == first['energy']


def test_md_host_reuses_the_list_between_rebuilds():
    """MD steps between two LAMMPS neighbor-list rebuilds: positions move (inside the skin), the list does not -- with
    snet_md_list_unchanged the host reuses the flattened list / node maps / species it uploaded at the rebuild and must give
    exactly what a from-scratch call gives for the new positions (the edge set inside the cutoff is re-derived either way)"""
    cfg, sd, cutoff, types, pos, cell, ei, ev = _setup('mini')
    n = len(types)
    x, tag, nlocal, rows = lammps_domain(pos, cell, np.ones(n, bool), cutoff + SKIN)
    ty_all = np.asarray(types)[tag - 1]
    host, fresh = MdHost(cfg, sd), MdHost(cfg, sd)
    host.compute(x, tag, nlocal, rows, ty_all)                       # step of the rebuild
    rng = np.random.default_rng(3)
    d_owned = rng.normal(0.0, 0.03, (n, 3))                          # every image of an atom moves with its owner
    for step in range(2):
        x2 = x + (step + 1) * d_owned[tag - 1]
        a = host.compute(x2, tag, nlocal, rows, ty_all, unchanged=True)
        b = fresh.compute(x2, tag, nlocal, rows, ty_all)
        assert a['n_edges'] == b['n_edges'] and a['energy'] == b['energy']
        assert np.array_equal(a['f'], b['f']) and np.array_equal(a['virial'], b['virial'])
    # a changed list without the hint is picked up (the hint is one-shot)
    c = host.compute(x, tag, nlocal, rows, ty_all)
    d = fresh.compute(x, tag, nlocal, rows, ty_all)
    assert np.array_equal(c['f'], d['f'])


def test_md_host_new_list_of_the_same_size_is_not_served_from_a_stale_topology_cache():
    """Round 6 (found by the MD loop, tools/md_loop.py): NativeModel switches the model's topology cache on (it keys on buffer
    addresses and sizes), snet_md_compute rewrites the same buffers every step -- a NEW list with the same node and edge counts
    (here: every neighbor row reversed, positions moved) used to be evaluated with the previous list's source grouping and tile
    lists: exact energy, forces 10-30 % off.  The host now invalidates the cache on every call."""
    cfg, sd, cutoff, types, pos, cell, ei, ev = _setup('mini')
    n = len(types)
    x, tag, nlocal, rows = lammps_domain(pos, cell, np.ones(n, bool), cutoff + SKIN)
    ty_all = np.asarray(types)[tag - 1]
    host, fresh = MdHost(cfg, sd), MdHost(cfg, sd)
    a = host.compute(x, tag, nlocal, rows, ty_all)
    rows2 = [r[::-1].copy() for r in rows]
    x2 = x + np.random.default_rng(5).normal(0.0, 0.02, (n, 3))[tag - 1]
    b = host.compute(x2, tag, nlocal, rows2, ty_all)          # second call on the same host: same sizes, other edge order
    c = fresh.compute(x2, tag, nlocal, rows2, ty_all)
    assert b['n_edges'] == c['n_edges'] == a['n_edges']
    assert b['energy'] == c['energy'] and np.array_equal(b['f'], c['f']) and np.array_equal(b['virial'], c['virial'])
    assert np.abs(b['f'] - a['f']).max() > 1e-4                # (the second configuration really is another one)


def test_md_host_rejects_bad_input():
    cfg, sd, cutoff, types, pos, cell, ei, ev = _setup('unit')
    x, tag, nlocal, rows = lammps_domain(pos, cell, np.ones(len(types), bool), cutoff + SKIN)
    host = MdHost(cfg, sd)
    with pytest.raises(RuntimeError, match='species'):
        host.compute(x, tag, nlocal, rows, np.full(len(x), 7))       # type -> species 7 of a 4-species model
    with pytest.raises(RuntimeError, match='ghost_mode'):
        host.compute(x, tag, nlocal, rows, np.asarray(types)[tag - 1], ghost_mode=3)
    with pytest.raises(RuntimeError, match='atomic stress'):
        host.compute(x, tag, nlocal, rows, np.asarray(types)[tag - 1], ghost_mode=1)


class _TwoRankHalo:
    """test-side stand-in for LAMMPS' comm: ghost node rows <- owner rows, and the reverse sum"""

    def __init__(self, world):
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.plan = {}  # (rank, peer) -> (ghost node ids on rank, node ids on peer)

    def member(self, r, world):
        grp = self

        class M:
            def forward(self, x, n_local):
                grp.slots[r] = x
                torch.cuda.synchronize(); grp.barrier.wait()
                for p in range(world):
                    if (r, p) in grp.plan:
                        gi, pi = grp.plan[(r, p)]
                        x[gi] = grp.slots[p][pi]
                torch.cuda.synchronize(); grp.barrier.wait()

            def reverse(self, gx, n_local):
                grp.slots[r] = gx
                torch.cuda.synchronize(); grp.barrier.wait()
                for p in range(world):
                    if (p, r) in grp.plan:
                        gi, pi = grp.plan[(p, r)]   # peer p holds my rows pi as its ghost nodes gi
                        gx.index_add_(0, pi, grp.slots[p][gi])
                torch.cuda.synchronize(); grp.barrier.wait()
        return M()


def test_md_host_ghost_nodes_two_ranks_equal_single_process():
    """pair e3gnn/parallel semantics: ghosts are graph nodes, features exchanged through the halo
    hooks, ghost forces left in f[ghost] for the host's reverse_comm (done by hand below)."""
    cfg, sd, cutoff, types, pos, cell, ei, ev = _setup('mini')
    ref = _reference(cfg, sd, types, ei, ev)
    n, world = len(types), 2
    frac = pos @ np.linalg.inv(cell)
    owner = (frac[:, 0] - np.floor(frac[:, 0]) >= 0.5).astype(int)
    doms = [lammps_domain(pos, cell, owner == r, cutoff + SKIN) for r in range(world)]
    grp = _TwoRankHalo(world)
    local_node = np.full(n, -1)
    for r in range(world):
        local_node[np.nonzero(owner == r)[0]] = np.arange((owner == r).sum())
    expect_nodes = []
    for r, (x, tag, nlocal, rows) in enumerate(doms):
        gid = tag - 1
        ghost_ids = [g for g in dict.fromkeys(gid[nlocal:].tolist()) if owner[g] != r]  # first-seen order, unique
        expect_nodes.append(ghost_ids)
        for p in range(world):
            sel = [(nlocal + k, local_node[g]) for k, g in enumerate(ghost_ids) if owner[g] == p]
            if sel:
                a = np.array(sel)
                grp.plan[(r, p)] = (torch.as_tensor(a[:, 0], device='cuda:0'), torch.as_tensor(a[:, 1], device='cuda:0'))
    hosts = [MdHost(cfg, sd) for _ in range(world)]
    results, errors = [None] * world, []

    def run(r):
        try:
            x, tag, nlocal, rows = doms[r]
            hosts[r].model.set_halo(grp.member(r, world), fold_forces=False)
            results[r] = hosts[r].compute(x, tag, nlocal, rows, np.asarray(types)[tag - 1], ghost_mode=1, vflag_atom=0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            grp.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errors, errors
    F = np.zeros((n, 3)); Ea = np.zeros(n); e_tot = 0.0; vir = np.zeros(6)
    for r, ((x, tag, nlocal, rows), out) in enumerate(zip(doms, results)):
        assert out['n_nodes'] == nlocal + len(expect_nodes[r]) and len(expect_nodes[r]) > 0
        assert [int(tag[a]) - 1 for a in out['node_to_atom'][nlocal:]] == expect_nodes[r]
        # snet_md_nodes (what a pair style calls to lay out its RCCL halo) numbers the nodes identically
        n2a = np.full(len(x), -1, np.int32)
        nn = C.c_int64()
        il = np.arange(nlocal, dtype=np.int32)
        tg = np.ascontiguousarray(tag, np.int32)
        from sevennet_amd import _lib
        _lib.check(hosts[r].lib.snet_md_nodes(nlocal, C.c_void_p(il.ctypes.data), len(x), C.c_void_p(tg.ctypes.data), 4, 1,
                                              C.c_void_p(n2a.ctypes.data), C.byref(nn)), 'snet_md_nodes')
        assert nn.value == out['n_nodes'] and np.array_equal(n2a[:nn.value], out['node_to_atom'])
        np.add.at(F, tag - 1, out['f'])              # LAMMPS reverse_comm: ghost forces go to their owners
        np.add.at(Ea, tag[:nlocal] - 1, out['eatom'][:nlocal])
        e_tot += out['energy']; vir += out['virial']
    assert abs(e_tot - float(ref['energy'][0])) <= 2e-6 * abs(float(ref['energy'][0]))
    assert np.abs(F - ref['forces']).max() <= min(F_TOL, max(1e-6, 3e-5 * np.abs(ref['forces']).max()))
    assert np.abs(Ea - ref['atomic_energy']).max() <= max(5e-6, 3e-5 * np.abs(ref['atomic_energy']).max())
    v = ref['virial'][[0, 1, 2, 3, 5, 4]]
    assert np.abs(vir - v).max() <= max(1e-6, 3e-5 * np.abs(v).max())


def test_md_host_isolated_atoms_no_edges():
    """two atoms farther apart than the cutoff (non-periodic box, neighbor rows hold only the out-of-range
    partner): no edge survives the filter, energies are the bare atomic terms, forces are zero"""
    from helpers import oracle_model
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = unit_test_config()
    sd = random_state_dict(cfg, seed=5)
    x = np.array([[0.0, 0.0, 0.0], [4.5, 0.0, 0.0]])           # cutoff 4.0 < 4.5 < cutoff + skin
    rows = [np.array([1], np.int32), np.array([0], np.int32)]
    host = MdHost(cfg, sd)
    out = host.compute(x, np.array([1, 2]), 2, rows, np.array([2, 0]))
    assert out['n_edges'] == 0 and out['n_nodes'] == 2
    assert np.abs(out['f']).max() == 0.0 and np.abs(out['virial']).max() == 0.0
    ref = oracle_model(cfg, sd).forward(np.array([2, 0]), np.zeros((2, 0), np.int64), np.zeros((0, 3)))
    assert np.abs(out['eatom'] - ref['atomic_energy'].numpy()).max() < 1e-5
    assert abs(out['energy'] - float(ref['energy'])) < 1e-5


def test_md_host_at_the_benchmark_size_matches_the_engine():
    """The LAMMPS-facing host at BASELINE config 3's size: one process holding the whole 97 336-atom cell the way LAMMPS presents it
    (owned atoms + 54 k periodic ghost images carrying their owners' tags, a FULL neighbor list with a 1-A skin: 5.2 M slots for 2.7 M
    edges inside the cutoff) through snet_md_compute == the engine on the GPU-built periodic graph of the same positions -- energy,
    per-atom energies, forces, LAMMPS-order virial --, and again after the atoms moved inside the skin with the list declared unchanged
    (pair_e3gnn.cpp:74-289 runs exactly this every MD step; reference analogue tests/lammps_tests/test_lammps.py:201-220)."""
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic
    from sevennet_amd.neighbor_gpu import build_graph_gpu
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, seed=0)
    pos, cell = diamond_cubic(5.431, (23,) * 3, 0.05, 2)
    n, rc = len(pos), cfg['cutoff'] + SKIN
    L = np.diag(cell)
    xs, tags, shifts = [pos], [np.arange(1, n + 1)], [np.zeros((n, 3))]
    for s in np.ndindex(3, 3, 3):
        sh = (np.array(s) - 1) * L
        if not sh.any():
            continue
        img = pos + sh
        sel = np.all((img > -rc) & (img < L + rc), axis=1)
        xs.append(img[sel]); tags.append(np.nonzero(sel)[0] + 1); shifts.append(np.broadcast_to(sh, (int(sel.sum()), 3)))
    x, tag, shift = np.concatenate(xs), np.concatenate(tags), np.concatenate(shifts)
    nb = cKDTree(x).query_ball_point(x[:n], rc, workers=-1)
    rows = [np.asarray([j for j in r if j != i], np.int32) for i, r in enumerate(nb)]
    ty_all = np.zeros(len(x), np.int64)
    host = MdHost(cfg, sd)
    eng = HipForceEngine(cfg, sd, device='cuda:0')

    def check(p, unchanged):
        out = host.compute(p[tag - 1] + shift, tag, n, rows, ty_all, vflag_atom=0, unchanged=unchanged)
        g = build_graph_gpu(np.zeros(n, np.int64), p, cell, cfg['cutoff'], device='cuda:0')
        ref = eng.compute(g)
        torch.cuda.synchronize()
        assert out['n_nodes'] == n and out['n_edges'] == g.n_edges
        e_ref, f_ref = float(ref['energy'].cpu()), ref['forces'].cpu().numpy().astype(np.float64)
        assert abs(out['energy'] - e_ref) <= 1e-7 * abs(e_ref), (out['energy'], e_ref)
        assert np.abs(out['f'][:n] - f_ref).max() <= 5e-6 * np.abs(f_ref).max(), np.abs(out['f'][:n] - f_ref).max()
        assert np.abs(out['f'][n:]).max() == 0.0
        assert np.abs(out['eatom'][:n] - ref['atomic_energy'].cpu().numpy()).max() <= 1e-6
        v = ref['virial'].cpu().numpy()[[0, 1, 2, 3, 5, 4]]      # model xx yy zz xy yz zx -> LAMMPS xx yy zz xy xz yz
        assert np.abs(out['virial'] - v).max() <= 1e-6 * np.abs(v).max()

    check(pos, False)
    moved = pos + np.random.default_rng(8).uniform(-0.2, 0.2, pos.shape)    # inside half the skin: the list stays valid
    check(moved, True)
