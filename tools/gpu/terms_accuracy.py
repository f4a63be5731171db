#!/usr/bin/env python3
"""Force error of the fused engine vs the fp64 CPU oracle for 1 / 2 / 3 bf16 terms (SevenNet-0 shape, 64 and 216 atoms)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.model import OracleModel  # noqa: E402
from sevennet_amd.engine import HipForceEngine, build_graph  # noqa: E402
from sevennet_amd.model_spec import sevennet_0_config  # noqa: E402
from sevennet_amd.neighbor import diamond_cubic, neighbor_list  # noqa: E402
from sevennet_amd.synthetic import random_state_dict  # noqa: E402

cfg = sevennet_0_config()
for seed, reps, sigma in ((0, 2, 0.05), (1, 3, 0.15)):
    sd = random_state_dict(cfg, seed=seed)
    pos, cell = diamond_cubic(5.431, (reps,) * 3, sigma, seed)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    types = np.zeros(len(pos), np.int64)
    ref = OracleModel(cfg, sd, dtype=torch.float64).forward(types, ei, ev)
    g = build_graph(types, ei, ev, device='cuda:0')
    fs = float(ref['forces'].abs().max())
    for name, kw in (('separate', dict(fused=False)), ('fused x6', dict(fused=True, fused_terms=3)),
                     ('fused x3', dict(fused=True, fused_terms=2)), ('fused bf16', dict(fused=True, fused_terms=1))):
        out = HipForceEngine(cfg, sd, device='cuda:0', **kw).compute(g)
        torch.cuda.synchronize()
        dF = float(np.abs(out['forces'].cpu().numpy() - ref['forces'].numpy()).max())
        dE = abs(float(out['energy'].cpu()) - float(ref['energy'])) / len(pos)
        print(f'N={len(pos)} seed={seed} max|F|={fs:.3f} {name:10s} max|dF|={dF:.3e} eV/A  |dE|/N={dE:.3e} eV')
