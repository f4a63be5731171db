#!/usr/bin/env python3
"""Compute cost of ONE brick of a W-way decomposition of the benchmark cell, on one GPU.

    python tools/brick_cost.py --world 8 [--rank 0] [--reps 23]

Builds rank R's brick graph exactly as `bench.py --gpus W` does and times the step with a no-op
halo (ghost rows keep whatever they hold): the per-rank kernel time of a W-GPU run without the
exchange.  total/W of the single-GPU step vs this number = the scaling the kernels alone allow.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NoHalo:
    def forward(self, x, n_local):
        pass

    def reverse(self, gx, n_local):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--reps', type=int, default=23)
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    from bench import model_config, species_of
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.neighbor import diamond_cubic
    from sevennet_amd.parallel import build_brick_graph
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config('sevennet_0')
    eng = HipForceEngine(cfg, random_state_dict(cfg, 0))
    pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.05, 2)
    bg = build_brick_graph(pos, cell, species_of(cfg, len(pos)), cfg['cutoff'], a.world, a.rank)
    g = build_graph(bg.types, bg.edge_index, bg.edge_vec, n_local=bg.n_local, n_interior=bg.n_interior, device='cuda:0')
    halo = NoHalo()
    for _ in range(3):
        eng.compute(g, halo=halo)
    torch.cuda.synchronize()
    eng.events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.compute(g, halo=halo)
    t_host = (time.perf_counter() - t0) / a.steps * 1e3   # host time to enqueue a step (the GPU runs behind)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps * 1e3
    times = {k: round(float(np.sum(v)) / a.steps, 3) for k, v in eng.kernel_times_ms().items()}
    ghosts = g.n_total - g.n_local
    print(f'world {a.world} rank {a.rank}: {g.n_local} local atoms, {ghosts} ghost rows, {g.n_edges} edges, '
          f'{g.n_pairs} radial-weight rows; {dt:.2f} ms/step (no exchange), host enqueue {t_host:.2f} ms/step; '
          f'halo bytes/exchange {ghosts * 480 * 4 / 1e6:.1f} MB')
    print(dict(sorted(times.items(), key=lambda kv: -kv[1])))


if __name__ == '__main__':
    main()
