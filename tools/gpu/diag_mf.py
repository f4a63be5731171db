import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import model_config
from oracle.model import OracleModel
from sevennet_amd.engine import HipForceEngine
from sevennet_amd.neighbor import neighbor_list
from sevennet_amd.neighbor_gpu import build_graph_gpu
from sevennet_amd.synthetic import random_state_dict
cfg = model_config('sevennet_mf_ompa'); sd = random_state_dict(cfg, seed=0)
a = 5.431
basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0], [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]]) * a
unit = basis + np.random.default_rng(11).normal(0.0, 0.08, basis.shape)
z_unit = np.array([3, 8, 14, 22, 8, 3, 22, 14])
def tile(n):
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), -1).reshape(-1, 3) * a
    return (g[:, None, :] + unit[None, :, :]).reshape(-1, 3), np.eye(3) * n * a, np.tile(z_unit, n ** 3)
pos_s, cell_s, ty_s = tile(2)
ei, ev, _ = neighbor_list(pos_s, cell_s, [True] * 3, cfg['cutoff'])
ref = OracleModel(cfg, sd, dtype=torch.float64, modal='mpa').forward(ty_s, ei, ev)
f_unit = ref['forces'].numpy()[:8]
print('max|F| unit', np.abs(f_unit).max())
for fused, terms in ((False, 2), (True, 3), (True, 2)):
    eng = HipForceEngine(cfg, sd, device='cuda:0', modal='mpa', fused=fused, fused_terms=terms)
    for n in (2, 4, 8, 15):
        pos, cell, ty = tile(n)
        g = build_graph_gpu(ty, pos, cell, cfg['cutoff'], device='cuda:0', num_species=119)
        out = eng.compute(g); torch.cuda.synchronize()
        F = out['forces'].cpu().numpy().reshape(-1, 8, 3)
        err = np.abs(F - f_unit[None]).max(axis=(1, 2))
        print(f'fused={fused} terms={terms} n_tile={n} atoms={len(pos)} max err {err.max():.3e} median {np.median(err):.3e} worst replica {err.argmax()} n>1e-4: {(err > 1e-4).sum()}')
