"""Shared test helpers: golden fixtures, oracle construction, layout conversion."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_ts_golden(name, parallel=False):
    d = np.load(os.path.join(GOLDEN, f'ts_oracle_{name}.npz'))
    cfg = json.loads(str(d['__pconfig__' if parallel else '__config__']))
    pre = 'pw::' if parallel else 'w::'
    sd = {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}
    return d, cfg, sd


def oracle_model(cfg, sd, dtype=torch.float64):
    from oracle.model import OracleModel
    return OracleModel(cfg, sd, dtype=dtype)


def irmul_to_mulir(x, irreps):
    """engine (ir_mul) rows -> reference (mul_ir) rows; irreps = sevennet_amd Irreps."""
    from sevennet_amd.irreps import irmul_to_mulir_index
    x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    return x[:, irmul_to_mulir_index(irreps)]


def synthetic_system(n_rep=(2, 2, 2), a=5.431, sigma=0.05, seed=0, cutoff=5.0, n_species=1):
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    pos, cell = diamond_cubic(a, n_rep, sigma, seed)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cutoff)
    rng = np.random.default_rng(seed + 1)
    types = rng.integers(0, n_species, len(pos)) if n_species > 1 else np.zeros(len(pos), np.int64)
    return types, pos, cell, ei, ev


def packed_tiles_expected(row_ptr, lo, hi, group=8):
    """restatement of snet_edge_tiles_packed: greedy windows of <= 16 consecutive edges over <= 2 rows, per group of rows"""
    rp = [int(v) for v in row_ptr]
    e0, nodes = [], []
    for a in range(lo, hi, group):
        b = min(a + group, hi)
        e, n0 = rp[a], a
        while e < rp[b]:
            while rp[n0 + 1] <= e:
                n0 += 1
            n1 = n0 + 1
            while n1 < b and rp[n1 + 1] == rp[n1]:
                n1 += 1
            lim = rp[n1 + 1] if n1 < b else rp[n0 + 1]
            end = min(e + 16, lim)
            e0.append(e)
            nodes += [n0, n1 if end > rp[n0 + 1] else n0]
            e = end
    return e0 + [rp[hi]], nodes
