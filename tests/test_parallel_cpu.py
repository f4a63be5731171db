"""CPU, world_size 2/4 over gloo: the brick decomposition + halo exchange reproduce the
single-process result (the reference asserts the same for `e3gnn/parallel`,
tests/lammps_tests/test_lammps.py:540-578)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import oracle_model, synthetic_system


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_case():
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = unit_test_config(conv_denominator=8.0, shift=0.0, scale=1.0)
    sd = random_state_dict(cfg, seed=5)
    types, pos, cell, ei, ev = synthetic_system((3, 2, 2), sigma=0.08, seed=2, cutoff=4.0, n_species=4)
    return cfg, sd, types, pos, cell, ei, ev


def _host_halo(send_lists, recv_counts):
    """The product HaloExchange packs with HIP kernels only; for the CPU/gloo test of the exchange PLAN
    the two row primitives are replaced by torch index ops on host tensors."""
    from sevennet_amd.parallel import HaloExchange

    class HostHalo(HaloExchange):
        def _pack(self, x, idx):
            return x.index_select(0, idx.long())

        def _unpack_add(self, y, idx, rows):
            y.index_add_(0, idx.long(), rows)

        def _reduce_add(self, y, recv):  # same grouping and order as the device version
            for s in range(self.red_rows.numel()):
                k0, k1 = int(self.red_ptr[s]), int(self.red_ptr[s + 1])
                acc = recv[int(self.red_perm[k0])].clone()
                for k in range(k0 + 1, k1):
                    acc += recv[int(self.red_perm[k])]
                y[int(self.red_rows[s])] += acc

    return HostHalo(send_lists, recv_counts, 'cpu')


class _Exchange(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_local, halo, n_local, n_total):
        ctx.halo, ctx.n_local = halo, n_local
        x = torch.zeros(n_total, h_local.shape[1], dtype=h_local.dtype)
        x[:n_local] = h_local
        halo.forward(x, n_local)
        return x

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.halo.reverse(g, ctx.n_local)
        return g[:ctx.n_local], None, None, None


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sevennet_amd.parallel import build_brick_graph
        torch.set_num_threads(1)
        cfg, sd, types, pos, cell, ei, ev = _make_case()
        bg = build_brick_graph(pos, cell, types, cfg['cutoff'], world, rank)
        halo = _host_halo(bg.send_lists, bg.recv_counts)
        m = oracle_model(cfg, sd)
        nt = len(bg.types)
        out = m.forward_brick(bg.types, bg.edge_index, bg.edge_vec, bg.n_local,
                              lambda h: _Exchange.apply(h, halo, bg.n_local, nt))
        f = out['forces'].contiguous().clone()
        halo.reverse(f, bg.n_local)  # fold ghost-atom force rows into their owners
        e = out['energy'].reshape(1).clone()
        dist.all_reduce(e)
        vir = out['virial'].clone()
        dist.all_reduce(vir)
        q.put((rank, bg.global_ids[:bg.n_local], f[:bg.n_local].numpy(), float(e), vir.numpy(),
               out['atomic_energy'].numpy(), sum(bg.recv_counts)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_bricks_equal_single_process(world):
    cfg, sd, types, pos, cell, ei, ev = _make_case()
    ref = oracle_model(cfg, sd).forward(types, ei, ev)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    F = np.zeros((len(types), 3))
    Ea = np.zeros(len(types))
    for rank, gids, f, e, vir, ea, ng in res:
        F[gids] = f
        Ea[gids] = ea
        assert ng > 0  # the case really has ghosts
        assert abs(e - float(ref['energy'])) < 1e-9 * abs(float(ref['energy']))
        assert np.abs(vir - ref['virial'].numpy()).max() < 1e-9 * np.abs(ref['virial'].numpy()).max()
    assert np.abs(F - ref['forces'].numpy()).max() < 1e-10 * max(1.0, np.abs(ref['forces'].numpy()).max())
    assert np.abs(Ea - ref['atomic_energy'].numpy()).max() < 1e-10


def test_brick_plan_consistency():
    """send/recv lists of all ranks are mutually consistent and cover every cross-brick source"""
    from sevennet_amd.parallel import build_brick_graph, processor_grid
    assert processor_grid(8) == (2, 2, 2) and processor_grid(4) == (2, 2, 1) and processor_grid(2) == (2, 1, 1)
    cfg, sd, types, pos, cell, ei, ev = _make_case()
    world = 4
    bricks = [build_brick_graph(pos, cell, types, 4.0, world, r, neighbors=(ei, ev)) for r in range(world)]
    assert sum(b.n_local for b in bricks) == len(types)
    assert sum(b.edge_index.shape[1] for b in bricks) == ei.shape[1]
    for a in bricks:
        o = a.n_local
        for p, b in enumerate(bricks):
            c = a.recv_counts[p]
            assert c == len(b.send_lists[a.rank])
            # ghost rows of `a` that come from `b` are exactly b's send list, in order
            assert (a.global_ids[o:o + c] == b.global_ids[b.send_lists[a.rank]]).all()
            o += c
        assert a.recv_counts[a.rank] == 0
        assert (a.edge_index[0] < a.n_local).all()
