#!/usr/bin/env python3
"""End-to-end MD through the LAMMPS-style host: NVE velocity Verlet driven by snet_md_compute.

    python tools/md_loop.py [--reps 4] [--steps 300] [--oracle-steps 20]

What a LAMMPS run does around pair_e3gnn (pair_e3gnn.cpp:74-289), restated on the host in numpy: owned atoms + periodic ghost images
with their owners' tags, a FULL neighbor list with a skin, rebuilt every `--every` steps on the host (that is where LAMMPS builds
it); between rebuilds `snet_md_list_unchanged` + new positions only.  Reported: total-energy drift per atom per ps (a force that is
not the gradient of the energy, a stale list, a sign error in the force fold all show up here as a drift orders of magnitude above
the fp32 noise floor), steps per second host to host, and the deviation of the first steps' trajectory from the fp64 CPU oracle
integrated with the same scheme.  `SNET_FUSED_TERMS=2` in the environment runs the bf16x3 throughput mode for comparison.
Units: eV, A, fs, amu (1 eV / (A amu) = 9.648533e-3 A / fs^2)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

ACC = 9.648533212e-3    # eV / (A amu) in A / fs^2
KB = 8.617333262e-5     # eV / K


def md_scale_state(cfg, sd, types, ei, ev, f_max):
    """the synthetic weights with `rescale_atomic_energy.scale` chosen so that the largest force component of the start
    configuration is f_max eV/A (forces of the unit-scale synthetic network are ~0.03 eV/A: not an MD-like potential)"""
    import torch
    from oracle.model import OracleModel
    ref = OracleModel(cfg, sd, dtype=torch.float64).forward(types, ei, ev)
    k = f_max / float(ref['forces'].abs().max())
    out = dict(sd)
    out['rescale_atomic_energy.scale'] = (np.asarray(sd['rescale_atomic_energy.scale'], np.float64) * k).astype(np.float32)
    return out


def run_md(cfg, sd, pos0, cell, types, vel0, mass, dt, steps, every=10, skin=1.0, force_fn=None):
    """velocity Verlet; force_fn(pos) -> (E, F) overrides the GPU host (the oracle leg).  Returns dict(e_tot[steps + 1], pos, ...)"""
    from test_md_host_gpu import MdHost, lammps_domain
    n = len(pos0)
    pos, vel = pos0.copy(), vel0.copy()
    rc = cfg['cutoff'] + skin
    host = None if force_fn is not None else MdHost(cfg, sd)
    state, rebuilt = {}, []

    def forces(step):
        if force_fn is not None:
            return force_fn(pos)
        # LAMMPS' `neigh_modify every N check yes`: rebuild every N steps, and at once when an atom has moved half the skin
        rebuild = step % every == 0 or np.sqrt(((pos - state['pos_at_build']) ** 2).sum(1).max()) >= 0.5 * skin
        if rebuild:
            x, tag, nloc, rows = lammps_domain(pos, cell, np.ones(n, bool), rc)
            state.update(tag=tag, rows=rows, shift=x - pos[tag - 1], pos_at_build=pos.copy(), rebuilds=state.get('rebuilds', 0) + 1)
        rebuilt.append(rebuild)   # (between rebuilds LAMMPS moves the ghosts with their owners -- forward_comm of positions -- and keeps the list)
        x = pos[state['tag'] - 1] + state['shift']
        out = host.compute(x, state['tag'], n, state['rows'], np.asarray(types)[state['tag'] - 1], eflag_atom=0, vflag_atom=0, unchanged=not rebuild)
        f = np.zeros((n, 3))
        np.add.at(f, state['tag'] - 1, out['f'])        # newton on: LAMMPS folds the ghost forces into their owners (reverse_comm)
        return out['energy'], f

    e_pot, f = forces(0)
    e_tot = [e_pot + 0.5 * mass * (vel ** 2).sum() / ACC]
    traj = [pos.copy()]
    t_steps = []
    for s in range(1, steps + 1):
        t0 = time.perf_counter()
        vel += 0.5 * dt * ACC * f / mass
        pos += dt * vel
        e_pot, f = forces(s)
        vel += 0.5 * dt * ACC * f / mass
        t_steps.append(time.perf_counter() - t0)
        e_tot.append(e_pot + 0.5 * mass * (vel ** 2).sum() / ACC)
        traj.append(pos.copy())
    return dict(e_tot=np.asarray(e_tot), traj=traj, step_s=np.asarray(t_steps), every=every, rebuilt=np.asarray(rebuilt[1:], bool),
                rebuilds=state.get('rebuilds', 0), max_disp=float(np.sqrt(((pos - pos0) ** 2).sum(1).max())))


def drift_per_atom_per_ps(e_tot, n, dt):
    """slope of a least-squares line through E_total(t) (eV per atom per ps) and the rms fluctuation around it (eV per atom)"""
    t = np.arange(len(e_tot)) * dt * 1e-3
    a, b = np.polyfit(t, e_tot / n, 1)
    return float(a), float(np.sqrt(np.mean((e_tot / n - (a * t + b)) ** 2)))


def setup(reps, temperature, f_max, seed=7):
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    pos, cell = diamond_cubic(5.431, (reps,) * 3, 0.05, 2)
    types = np.zeros(len(pos), np.int64)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    sd = md_scale_state(cfg, random_state_dict(cfg, seed=0), types, ei, ev, f_max)
    mass = 28.0855
    rng = np.random.default_rng(seed)
    vel = rng.normal(0.0, np.sqrt(KB * temperature * ACC / mass), (len(pos), 3))
    vel -= vel.mean(0)
    return cfg, sd, pos, np.asarray(cell, float), types, vel, mass


def oracle_force_fn(cfg, sd, cell, types):
    import torch
    from oracle.model import OracleModel
    from sevennet_amd.neighbor import neighbor_list
    m = OracleModel(cfg, sd, dtype=torch.float64)

    def fn(pos):
        ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
        out = m.forward(types, ei, ev)
        return float(out['energy']), out['forces'].numpy().astype(np.float64)
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=4, help='diamond cells per axis (4 -> 512 atoms)')
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--dt', type=float, default=0.25, help='fs (the synthetic-weight potential is stiff: its atoms gain ~0.1 eV each in 100 fs)')
    ap.add_argument('--every', type=int, default=10, help='neighbor-list rebuild interval (steps)')
    ap.add_argument('--temperature', type=float, default=300.0)
    ap.add_argument('--fmax', type=float, default=0.5, help='largest force component of the start configuration (eV/A)')
    ap.add_argument('--oracle-steps', type=int, default=20)
    a = ap.parse_args()
    cfg, sd, pos, cell, types, vel, mass = setup(a.reps, a.temperature, a.fmax)
    n = len(pos)
    mode = os.environ.get('SNET_FUSED_TERMS', '4')
    r = run_md(cfg, sd, pos, cell, types, vel, mass, a.dt, a.steps, a.every)
    slope, rms = drift_per_atom_per_ps(r['e_tot'], n, a.dt)
    ts = r['step_s']
    reb, keep = ts[r['rebuilt']], ts[~r['rebuilt']]
    print(f'NVE through snet_md_compute: {n} atoms, {a.steps} steps of {a.dt} fs, T0 {a.temperature} K, max|F0| {a.fmax} eV/A, list rebuilt '
          f'every {a.every} steps or when an atom has moved half the skin ({r["rebuilds"]} rebuilds; host KD-tree, skin 1.0 A; largest displacement {r["max_disp"]:.2f} A), in-kernel products mode {mode} ({ {"4": "f16x3, fp32 class", "2": "bf16x3"}.get(mode, mode)})')
    print(f'  total energy: drift {slope:+.3e} eV/atom/ps, rms fluctuation {rms:.3e} eV/atom, E_tot(0) {r["e_tot"][0] / n:+.6f} '
          f'E_tot(end) {r["e_tot"][-1] / n:+.6f} eV/atom; kinetic energy start {0.5 * mass * (vel ** 2).sum() / ACC / n:.4f} eV/atom')
    print(f'  host to host: {1.0 / np.median(keep):.1f} steps/s between rebuilds (median {np.median(keep) * 1e3:.2f} ms per step), '
          f'{np.median(reb) * 1e3:.1f} ms on a rebuild step (numpy / KD-tree list build included)')
    if a.oracle_steps > 0:
        t0 = time.perf_counter()
        o = run_md(cfg, sd, pos, cell, types, vel, mass, a.dt, a.oracle_steps, force_fn=oracle_force_fn(cfg, sd, cell, types))
        dev = max(np.abs(r['traj'][k] - o['traj'][k]).max() for k in range(a.oracle_steps + 1))
        de = np.abs(r['e_tot'][:a.oracle_steps + 1] - o['e_tot']).max() / n
        print(f'  first {a.oracle_steps} steps against the fp64 CPU oracle integrated with the same scheme: max |dx| {dev:.3e} A, '
              f'max |dE_tot| {de:.3e} eV/atom ({time.perf_counter() - t0:.0f} s of CPU)')


if __name__ == '__main__':
    main()
