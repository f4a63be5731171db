import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import model_config
from sevennet_amd.engine import HipForceEngine, build_graph
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
eng = HipForceEngine(cfg, sd, device='cuda:0', modal='mpa', fused=False)
for n in (6, 7):
    pos, cell, ty = tile(n)
    g = build_graph_gpu(ty, pos, cell, cfg['cutoff'], device='cuda:0', num_species=119)
    out = eng.compute(g, keep=True); torch.cuda.synchronize()
    it = out['inter']
    for t in range(5):
        for key in ('si1', 'conv', 'gate_in', 'x'):
            x = it[f'{t}_{key}'].cpu().numpy()
            x = x[:len(pos)].reshape(n ** 3, 8, -1)
            d = np.abs(x - x[0][None])
            bad = np.argwhere(d > 1e-5 + 1e-4 * np.abs(x).max())
            if len(bad):
                reps = np.unique(bad[:, 0]); atoms = np.unique(bad[:, 1]); cols = np.unique(bad[:, 2])
                print(f'n={n} layer {t} {key}: dim {x.shape[-1]} bad replicas {len(reps)} first {reps[:10]} atoms-in-cell {atoms} cols {len(cols)} [{cols.min()}..{cols.max()}] first {cols[:12]}')
                rows = np.unique(bad[:, 0] * 8 + bad[:, 1])
                print('    bad rows', len(rows), rows[:20], '...', rows[-5:])
                r0 = rows[0]
                print('    row', r0, 'vals', x.reshape(-1, x.shape[-1])[r0, cols[:6]], 'expected', x[0, r0 % 8, cols[:6]])
                break
        else:
            continue
        break
    sr = g.species_rows
    for s in (3, 8, 14, 22):
        r = sr[s].cpu().numpy()
        print('   species', s, 'rows', len(r), r[:6], 'sorted', bool((np.diff(r) > 0).all()), 'types ok', bool((ty[r] == s).all()), 'ptr%16', sr[s].data_ptr() % 16)
