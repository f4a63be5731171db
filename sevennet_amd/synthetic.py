"""Seeded synthetic weights for a model config (SURVEY.md §8d "Synthetic weights").

No pretrained checkpoints are available offline, so benchmarks and parity tests
draw every o3.Linear / FCTP / FullyConnectedNet weight from N(0,1) (e3nn's own
initialisation), Bessel c_n = n*pi/rc (sevenn/nn/edge_embedding.py:95-99), and
take denominator / shift / scale from the config.  Keys are the reference's
state_dict names, so the same dict feeds the HIP engine and the CPU oracle.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from .model_spec import build_model_spec


def random_state_dict(config: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    spec = build_model_spec(config)
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for k, shp in spec.param_shapes().items():
        if k == 'edge_embedding.basis_function.coeffs':
            sd[k] = np.array([n * math.pi / spec.cutoff for n in range(1, spec.n_basis + 1)], dtype=np.float32)
        elif k.endswith('convolution.denominator'):
            t = int(k.split('_')[0])
            sd[k] = np.array([spec.layers[t].denominator], dtype=np.float32)
        elif k == 'rescale_atomic_energy.shift':
            sd[k] = np.broadcast_to(np.asarray(spec.config['shift'], dtype=np.float32).reshape(-1), shp).copy()
        elif k == 'rescale_atomic_energy.scale':
            sd[k] = np.broadcast_to(np.asarray(spec.config['scale'], dtype=np.float32).reshape(-1), shp).copy()
        else:
            sd[k] = rng.standard_normal(shp).astype(np.float32)
    return sd
