#!/usr/bin/env python3
"""Force error of the fused engine vs the fp64 CPU oracle per in-kernel precision mode, at MD-SCALE forces.

The synthetic weights of the parity tests give max|F| ~ 0.03 eV/A with rescale scale = 1, where the north-star bar
of 1e-4 eV/A absolute says little.  Here `rescale_atomic_energy.scale` is chosen per system so that max|F| of the
fp64 oracle is --fmax eV/A (default 8), and absolute AND relative errors are printed for every mode."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.model import OracleModel  # noqa: E402
from sevennet_amd.engine import HipForceEngine, build_graph  # noqa: E402
from sevennet_amd import model_spec  # noqa: E402
from sevennet_amd.neighbor import diamond_cubic, neighbor_list  # noqa: E402
from sevennet_amd.synthetic import random_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--fmax', type=float, default=8.0)
ap.add_argument('--models', default='sevennet_0,sevennet_l3i5,sevennet_mf_ompa')
ap.add_argument('--modes', default='separate,4,3,2')
a = ap.parse_args()

for mname in a.models.split(','):
    cfg = getattr(model_spec, mname + '_config')()
    modal = 'mpa' if cfg.get('use_modality') else None
    cases = ((0, 2, 0.05), (1, 3, 0.15)) if mname == 'sevennet_0' else ((0, 2, 0.1),)
    for seed, reps, sigma in cases:
        sd = random_state_dict(cfg, seed=seed)
        pos, cell = diamond_cubic(5.431, (reps,) * 3, sigma, seed)
        ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
        nsp = int(cfg.get('_number_of_species', 1))
        types = np.zeros(len(pos), np.int64) if nsp < 23 else np.random.default_rng(4).choice(np.array([3, 8, 14, 22]), size=len(pos)).astype(np.int64)
        ref = OracleModel(cfg, sd, dtype=torch.float64, modal=modal).forward(types, ei, ev)
        f1 = float(ref['forces'].abs().max())
        k = a.fmax / f1
        sd['rescale_atomic_energy.scale'] = (sd['rescale_atomic_energy.scale'] * k).astype(np.float32)
        ref = OracleModel(cfg, sd, dtype=torch.float64, modal=modal).forward(types, ei, ev)
        fs = float(ref['forces'].abs().max())
        g = build_graph(types, ei, ev, device='cuda:0', num_species=nsp)
        for mode in a.modes.split(','):
            kw = dict(fused=False) if mode == 'separate' else dict(fused=True, fused_terms=int(mode))
            try:
                out = HipForceEngine(cfg, sd, device='cuda:0', modal=modal, **kw).compute(g)
            except Exception as e:  # noqa: BLE001
                print(f'{mname} N={len(pos)} mode={mode}: {type(e).__name__}: {e}')
                continue
            torch.cuda.synchronize()
            dF = float(np.abs(out['forces'].cpu().numpy() - ref['forces'].numpy()).max())
            dE = abs(float(out['energy'].cpu()) - float(ref['energy'])) / len(pos)
            print(f'{mname} N={len(pos)} seed={seed} scale x{k:.1f} max|F|={fs:.3f} eV/A  mode={mode:8s} '
                  f'max|dF|={dF:.3e} eV/A (rel {dF / fs:.2e})  |dE|/N={dE:.3e} eV', flush=True)
