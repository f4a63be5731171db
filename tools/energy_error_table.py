"""CPU half of the energy-error attribution (see tools/gpu/energy_error_dump.py).

For every intermediate the engine dumped, the fp64 oracle is continued FROM that tensor (all other inputs at their fp64
values); the resulting energy error per atom is what the engine had accumulated up to that stage.  The difference
between consecutive rows is the share each stage adds.  The same is done for the fp32 PyTorch oracle, which is the
arithmetic the engine replaces.

    python tools/energy_error_table.py [gpurun_out] > profiles/r04_energy_error_attribution.txt
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def continue_from(m, types, ei, ref_inter, stage, value):
    """fp64 energy per atom with `stage` replaced by `value` (torch fp64)"""
    types_t = torch.as_tensor(types, dtype=torch.long)
    src, dst = torch.as_tensor(ei[1]), torch.as_tensor(ei[0])
    N = len(types)
    emb, sh = ref_inter['edge_embedding'], ref_inter['edge_attr']
    if stage == 'edge_embedding':
        emb = value
    if stage == 'edge_attr':
        sh = value
    onehot, x = m.node_embed(types_t)
    t0, kind = (-1, None) if stage in ('edge_embedding', 'edge_attr') else (int(stage.split('_')[0]), stage.split('_', 1)[1])
    for ls in m.layers:
        if ls.t < t0:
            continue
        if ls.t == t0:
            x_in = ref_inter[f'{ls.t - 1}_x'] if ls.t > 0 else x
            sc = m.sc_intro(ls, x_in, onehot)
            if kind == 'x':
                x = value
                continue
            if kind == 'gate_in':
                x = ls.gate.apply(value)
                continue
            h = value if kind == 'si1' else None
            mm = value if kind == 'conv' else m.conv(ls, h, emb, sh, src, dst, N)
        else:
            sc = m.sc_intro(ls, x, onehot)
            h = m.si1(ls, x)
            mm = m.conv(ls, h, emb, sh, src, dst, N)
        y = m.si2(ls, mm)
        if sc is not None:
            y = y + sc
        x = ls.gate.apply(y)
    e_atom = m.readout(x, types_t)
    return e_atom.squeeze(-1)


def main():
    from oracle.model import OracleModel
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out')
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, seed=0)
    pos, cell = diamond_cubic(5.431, (2, 2, 2), 0.05, 0)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    types = np.zeros(len(pos), np.int64)
    m64 = OracleModel(cfg, sd, dtype=torch.float64)
    with torch.enable_grad():
        r64 = m64.forward(types, ei, ev, keep=True)
        r32 = OracleModel(cfg, sd, dtype=torch.float32).forward(types, ei, ev, keep=True)
    e64 = r64['atomic_energy']
    runs = {'fp32 PyTorch oracle': dict(inter={k: v.double() for k, v in r32['inter'].items()},
                                        atomic_energy=r32['atomic_energy'].double(), forces=r32['forces'].double(),
                                        energy=float(r32['energy']))}
    for f in sorted(glob.glob(os.path.join(d, 'energy_error_*.npz'))):
        z = np.load(f)
        runs['engine ' + os.path.basename(f)[13:-4]] = dict(
            inter={k[3:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith('i::')},
            atomic_energy=torch.from_numpy(z['atomic_energy']).double(), forces=torch.from_numpy(z['forces']).double(),
            energy=float(z['energy'][0]))
    stages = [k for k in r64['inter'] if k != 'x_embed']
    print(f'smoke system: {len(pos)} atoms, {ei.shape[1]} edges, E/N = {float(r64["energy"]) / len(pos):.6g} (unit rescale), '
          f'max|F| = {float(r64["forces"].abs().max()):.4g}')
    print('rows: mean over atoms of (atomic energy continued in fp64 from the run\'s tensor at that stage) - fp64, in units of 1e-7;')
    print('      in brackets the rms relative error of the tensor itself\n')
    names = list(runs)
    print(f'{"stage":16s}' + ''.join(f'{n[:26]:>28s}' for n in names))
    with torch.no_grad():
        for st in stages:
            row = f'{st:16s}'
            for n in names:
                v = runs[n]['inter'][st]
                e = continue_from(m64, types, ei, r64['inter'], st, v)
                rel = float((v - r64['inter'][st]).norm() / r64['inter'][st].norm())
                row += f'{float((e - e64).mean()) * 1e7:14.3f} [{rel:9.2e}] '
            print(row)
    row = f'{"final (run)":16s}'
    for n in names:
        row += f'{float((runs[n]["atomic_energy"] - e64).mean()) * 1e7:14.3f} [{"":9s}] '
    print(row)
    print()
    for n in names:
        r = runs[n]
        de = abs(r['energy'] - float(r64['energy'])) / len(pos)
        ea = (r['atomic_energy'] - e64)
        df = float((r['forces'] - r64['forces']).abs().max())
        print(f'{n:32s} |dE|/N {de:.3e} ({de / abs(float(r64["energy"]) / len(pos)):.2e} rel)  atomic-energy err: mean {float(ea.mean()):+.2e} '
              f'std {float(ea.std()):.2e} max {float(ea.abs().max()):.2e}   max|dF| {df:.3e} ({df / float(r64["forces"].abs().max()):.2e} rel)')


if __name__ == '__main__':
    main()
