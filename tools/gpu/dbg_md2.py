import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import md_loop as M
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg, sd, pos, cell, types, vel, mass = M.setup(reps, 300.0, 0.5)
n = len(pos)
for every in (1, 10):
    r = M.run_md(cfg, sd, pos, cell, types, vel, mass, 0.25, 80, every=every)
    e = r['e_tot'] / n
    print('every', every, 'n', n, 'dE(t) x1e6 every 4 steps:', ' '.join(f'{(e[k] - e[0]) * 1e6:+.1f}' for k in range(0, 81, 4)))
if reps <= 3:
    o = M.run_md(cfg, sd, pos, cell, types, vel, mass, 0.25, 40, force_fn=M.oracle_force_fn(cfg, sd, cell, types))
    eo = o['e_tot'] / n
    print('oracle', 'dE(t) x1e6 every 4 steps:', ' '.join(f'{(eo[k] - eo[0]) * 1e6:+.1f}' for k in range(0, 41, 4)))
    print('host - oracle E_tot x1e6:', ' '.join(f'{(e[k] - eo[k]) * 1e6:+.1f}' for k in range(0, 41, 4)))
