import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import md_loop as M
from test_md_host_gpu import MdHost, lammps_domain
cfg, sd, pos, cell, types, vel, mass = M.setup(3, 300.0, 0.5)
n = len(pos)
r = M.run_md(cfg, sd, pos, cell, types, vel, mass, 0.25, 80, every=10)
fn = M.oracle_force_fn(cfg, sd, cell, types)
host = MdHost(cfg, sd)
for k in (0, 20, 40, 60, 80):
    p = r['traj'][k]
    e0, f0 = fn(p)
    x, tag, nloc, rows = lammps_domain(p, cell, np.ones(n, bool), cfg['cutoff'] + 1.0)
    out = host.compute(x, tag, n, rows, np.asarray(types)[tag - 1], eflag_atom=0, vflag_atom=0)
    f = np.zeros((n, 3)); np.add.at(f, tag - 1, out['f'])
    d = f - f0
    v = (r['traj'][min(k + 1, 80)] - r['traj'][max(k - 1, 0)])
    print(f'step {k}: max|F| {np.abs(f0).max():.3f} max|dF| {np.abs(d).max():.3e} rms dF {np.sqrt((d**2).mean()):.3e} dE/atom {(out["energy"] - e0) / n:+.3e} '
          f'sum dF.v / |v| {float((d * v).sum() / np.sqrt((v ** 2).sum())):+.3e}  E_tot_run-E_pot_oracle check: {r["e_tot"][k] / n:+.6f}')
