"""Where does the engine's energy error come from?  

GPU half: runs the smoke system (SevenNet-0 shape, 64-atom Si, seeded weights) through the Python-hosted engine in
several precision configurations with keep=True and writes every intermediate (reference mul_ir layout) plus energies
and forces to gpurun_out/energy_error_<mode>.npz.  The CPU half (tools/energy_error_table.py) continues each dumped
intermediate in the fp64 oracle, so the table attributes the final energy error to the stage that introduced it.

    gpurun --timeout 600 -- python tools/gpu/energy_error_dump.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

MODES = {
    'default': dict(),
    'no_species_tables': dict(species_tables=False),   # round 3's layer 0: per-atom GEMMs
    'linear_fp32': dict(linear_mode='fp32'),
    'unfused': dict(fused=False),
    'unfused_all_fp32': dict(fused=False, linear_mode='fp32', mlp_mode='fp32'),
    'fused_bf16x6': dict(fused_terms='bf16x6'),
}


def main():
    from helpers import irmul_to_mulir
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, seed=0)
    pos, cell = diamond_cubic(5.431, (2, 2, 2), 0.05, 0)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    types = np.zeros(len(pos), np.int64)
    g = build_graph(types, ei, ev, device='cuda:0')
    order = g.order.cpu().numpy() if g.order is not None else np.arange(ei.shape[1])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    modes = dict(MODES)
    for extra in sys.argv[1:]:   # name=kw:value,kw:value  (ad-hoc engine keywords)
        name, kws = extra.split('=')
        modes[name] = {k: (v if not v.replace('.', '').isdigit() else int(v)) for k, v in (p.split(':') for p in kws.split(','))}
    for name, kw in modes.items():
        eng = HipForceEngine(cfg, sd, device='cuda:0', **kw)
        out = eng.compute(g, keep=True)
        torch.cuda.synchronize()
        d = dict(energy=out['energy'].cpu().numpy(), atomic_energy=out['atomic_energy'].cpu().numpy(),
                 forces=out['forces'].cpu().numpy(), dE_dr=out['dE_dr'].cpu().numpy(), edge_order=order)
        inter = out['inter']
        # per-edge arrays are in the engine's (center-sorted) edge order: back to the caller's
        for key in ('edge_embedding', 'edge_attr'):
            a = inter[key].cpu().numpy()
            b = np.empty_like(a)
            b[order] = a
            d['i::' + key] = b
        for t, L in enumerate(eng.layers):
            ls = L.spec
            for key, irr in ((f'{t}_si1', ls.si1.irreps_out), (f'{t}_conv', ls.conv.irreps_out),
                             (f'{t}_gate_in', ls.gate.irreps_in), (f'{t}_x', ls.gate.irreps_out)):
                d['i::' + key] = irmul_to_mulir(inter[key], irr)
        np.savez(os.path.join(ROOT, 'gpurun_out', f'energy_error_{name}.npz'), **d)
        print(name, 'E/N', float(d['energy'][0]) / len(pos), flush=True)


if __name__ == '__main__':
    main()
