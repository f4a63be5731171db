import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import md_loop as M
from test_md_host_gpu import MdHost, lammps_domain
cfg, sd, pos, cell, types, vel, mass = M.setup(3, 300.0, 0.5)
n = len(pos)
fn = M.oracle_force_fn(cfg, sd, cell, types)
e0, f0 = fn(pos)
host = MdHost(cfg, sd)
x, tag, nloc, rows = lammps_domain(pos, cell, np.ones(n, bool), cfg['cutoff'] + 1.0)
out = host.compute(x, tag, n, rows, np.asarray(types)[tag - 1], eflag_atom=0, vflag_atom=0)
f = np.zeros((n, 3)); np.add.at(f, tag - 1, out['f'])
print('step0: E oracle', e0 / n, 'host', out['energy'] / n, 'dE/atom', (out['energy'] - e0) / n, 'max|dF|', np.abs(f - f0).max(), 'max|F|', np.abs(f0).max(), 'edges', out['n_edges'])
# move atoms a little, list unchanged
rng = np.random.default_rng(0)
pos2 = pos + rng.normal(0, 0.05, pos.shape)
e1, f1 = fn(pos2)
x2 = pos2[tag - 1] + (x - pos[tag - 1])
out = host.compute(x2, tag, n, rows, np.asarray(types)[tag - 1], eflag_atom=0, vflag_atom=0, unchanged=True)
f = np.zeros((n, 3)); np.add.at(f, tag - 1, out['f'])
print('moved: E oracle', e1 / n, 'host', out['energy'] / n, 'dE/atom', (out['energy'] - e1) / n, 'max|dF|', np.abs(f - f1).max(), 'edges', out['n_edges'])
