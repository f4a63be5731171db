#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference (run in the BUILD container only).

Nothing here travels as code to the GPU box: the outputs are data fixtures
(inputs + expected outputs + weight arrays).  Sources, all read-only:

  /root/reference/tests/data/checkpoints/cp_0.pth
      -> w3j_cp0.npz        the 8 real Wigner-3j buffers e3nn stored in the
                            reference's test checkpoint (pins l<=2 CG incl. sign)
      -> cp0_state.npz      its full state_dict + config (pins tensor shapes /
                            the checkpoint layout the engine ingests)
  /root/reference/example_inputs/md_serial_example/deployed_serial.pt
  /root/reference/example_inputs/md_parallel_example/deployed_parallel/*.pt
      -> ts_oracle_*.npz    weights recovered from the frozen TorchScript
                            constants (re-expressed in the reference's
                            state_dict naming) + inputs + the outputs the
                            reference model itself produces on CPU here.

Usage:  python oracle/tools/make_golden.py
"""
import json
import math
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle.e3 import Irreps  # noqa: E402
from oracle.model import (OracleModel, fctp_instructions,  # noqa: E402
                          linear_instructions)
from sevennet_amd.neighbor import neighbor_list  # noqa: E402

TS_CONFIG = {  # deployed example model: sevenn 0.8.6, Hf/O
    'cutoff': 4.0, 'channel': 4, 'lmax': 1, 'is_parity': True,
    'num_convolution_layer': 4, 'self_connection_type': 'nequip',
    'cutoff_function': {'cutoff_function_name': 'poly_cut', 'poly_cut_p_value': 6},
    'radial_basis': {'radial_basis_name': 'bessel', 'bessel_basis_num': 8},
    'weight_nn_hidden_neurons': [64, 64], 'act_radial': 'silu',
    '_normalize_sph': False, 'version': '0.8.6', '_number_of_species': 2,
    '_legacy_v08': True, 'chemical_species': ['Hf', 'O'],
}


# --------------------------------------------------------------------------- #
def dump_cp0():
    cp = torch.load(f'{REF}/tests/data/checkpoints/cp_0.pth', map_location='cpu', weights_only=False)
    sd = cp['model_state_dict']
    w3j = {k.split('._w3j_')[1]: v.numpy() for k, v in sd.items() if '_w3j_' in k}
    np.savez(os.path.join(OUT, 'w3j_cp0.npz'), **w3j)
    cfg = {k: v for k, v in cp['config'].items()
           if isinstance(v, (int, float, str, bool, list, dict)) and k not in ('data_format_args', 'continue')}
    cfg['_type_map'] = {str(k): v for k, v in cp['config']['_type_map'].items()}
    arrays = {k: v.numpy() for k, v in sd.items() if '_w3j_' not in k and 'output_mask' not in k}
    np.savez(os.path.join(OUT, 'cp0_state.npz'), __config__=json.dumps(cfg), **arrays)
    print('cp0:', len(w3j), 'w3j tensors,', len(arrays), 'state tensors')


# --------------------------------------------------------------------------- #
def constant_uses(module):
    code, consts = module.code_with_constants
    cm = consts.const_mapping
    defs = {}
    lines = code.split('\n')
    for ln in lines:
        mt = re.match(r'\s*(\w+) = (.*)$', ln)
        if mt:
            defs.setdefault(mt.group(1), mt.group(2))
    uses = []
    for ln in lines:
        for mt in re.finditer(r'CONSTANTS\.(c\d+)', ln):
            t = cm[mt.group(1)]
            if t.dim() == 0:
                continue
            explicit = None
            mm = re.search(r'torch\.(?:tensordot|matmul)\((\w+), CONSTANTS', ln)
            if mm:
                d = defs.get(mm.group(1), '')
                m2 = re.match(r'torch\.mul\(\w+, ([0-9.eE+-]+)\)', d)
                if m2:
                    explicit = float(m2.group(1))
            uses.append((mt.group(1), t.detach().clone(), explicit, ln.strip()))
    return uses


class Walker:
    def __init__(self, uses):
        self.uses, self.i = uses, 0

    def take(self, shape):
        while self.i < len(self.uses):
            name, t, explicit, ln = self.uses[self.i]
            self.i += 1
            if tuple(t.shape) == tuple(shape):
                return t, explicit, ln
            if t.dim() == 3 and t.shape[0] in (1, 3, 5):  # folded w3j constant
                continue
            raise RuntimeError(f'unexpected constant {name}{tuple(t.shape)} while looking for {shape}: {ln}')
        raise RuntimeError(f'ran out of constants looking for {shape}')


def skip_ghost_copy(w: Walker, n_blocks: int):
    """ghost_* twin modules (model_build.py:383-421) reuse the same constants."""
    for _ in range(n_blocks):
        assert w.uses[w.i][0] == w.uses[w.i - n_blocks][0]
        w.i += 1


def take_linear(w: Walker, irreps_in, irreps_out):
    ins = linear_instructions(irreps_in, irreps_out)
    fan = [0] * len(irreps_out)
    for i, j in ins:
        fan[j] += irreps_in[i][0]
    blocks = []
    for i, j in ins:
        t, explicit, ln = w.take((irreps_in[i][0], irreps_out[j][0]))
        alpha = 1.0 / math.sqrt(fan[j])
        if explicit is not None:
            assert abs(explicit - alpha) < 1e-6, (explicit, alpha, ln)
        elif abs(alpha - 1.0) > 1e-12:
            t = t / alpha  # scalar was fused into the (smaller) weight operand
        blocks.append(t.reshape(-1))
    return torch.cat(blocks) if blocks else torch.zeros(0)


def take_fctp(w: Walker, irreps_in, n_sp, irreps_out):
    blocks = []
    for i, j in fctp_instructions(irreps_in, irreps_out):
        t, explicit, ln = w.take((irreps_in[i][0], n_sp, irreps_out[j][0]))
        blocks.append(t.reshape(-1))  # alpha always rides on the one-hot operand
    return torch.cat(blocks)


def extract_state(uses, model: OracleModel, layers, first=True, last=True):
    """Walk constants in program order, mirroring the module order of
    sevenn/nn/interaction_blocks.py:41-76."""
    w = Walker(uses)
    sd = {}
    ns = model.num_species

    def conv_part(ls):
        t = ls.t
        for i in range(len(ls.mlp_dims) - 1):
            c, _, _ = w.take((ls.mlp_dims[i], ls.mlp_dims[i + 1]))
            sd[f'{t}_convolution.weight_nn.layer{i}.weight'] = c * math.sqrt(ls.mlp_dims[i])
        sd[f'{t}_convolution.denominator'] = w.take((1,))[0]
        sd[f'{t}_self_interaction_2.linear.weight'] = take_linear(w, ls.irreps_out_tp, ls.irreps_gate_in)

    def pre_part(ls):
        t = ls.t
        sd[f'{t}_self_connection_intro.fc_tensor_product.weight'] = take_fctp(w, ls.irreps_x, ns, ls.irreps_gate_in)
        sd[f'{t}_self_interaction_1.linear.weight'] = take_linear(w, ls.irreps_x, ls.irreps_x)

    sd['edge_embedding.basis_function.coeffs'] = w.take((model.n_basis,))[0]
    if first:
        sd['onehot_to_feature_x.linear.weight'] = take_linear(w, Irreps(f'{ns}x0e'), model.irreps_embed)
    return w, sd, conv_part, pre_part


def extract_serial():
    m = torch.jit.load(f'{REF}/example_inputs/md_serial_example/deployed_serial.pt', map_location='cpu')
    uses = constant_uses(m)
    cfg = dict(TS_CONFIG)
    shell = OracleModel(dict(cfg, shift=0.0, scale=1.0, conv_denominator=1.0), None)
    w, sd, conv_part, pre_part = extract_state(uses, shell, shell.layers)
    for ls in shell.layers:
        pre_part(ls)
        conv_part(ls)
    sd['reduce_input_to_hidden.linear.weight'] = take_linear(w, shell.irreps_final, shell.irreps_hidden)
    sd['reduce_hidden_to_energy.linear.weight'] = take_linear(w, shell.irreps_hidden, Irreps('1x0e'))
    sd['rescale_atomic_energy.scale'] = w.take((1,))[0]
    sd['rescale_atomic_energy.shift'] = w.take((1,))[0]
    assert w.i == len(uses), (w.i, len(uses))
    cfg['conv_denominator'] = [float(sd[f'{t}_convolution.denominator']) for t in range(4)]
    cfg['shift'] = float(sd['rescale_atomic_energy.shift'])
    cfg['scale'] = float(sd['rescale_atomic_energy.scale'])
    return m, cfg, {k: v.numpy().astype(np.float32) for k, v in sd.items()}


def extract_parallel():
    """Same walk over the four deployed_parallel segments
    (segment boundaries: model_build.py:423-431)."""
    segs = [torch.jit.load(f'{REF}/example_inputs/md_parallel_example/deployed_parallel/deployed_parallel_{i}.pt',
                           map_location='cpu') for i in range(4)]
    cfg = dict(TS_CONFIG)
    shell = OracleModel(dict(cfg, shift=0.0, scale=1.0, conv_denominator=1.0), None)
    L = shell.L
    sd = {}
    for k, seg in enumerate(segs):
        uses = constant_uses(seg)
        w, sdk, conv_part, pre_part = extract_state(uses, shell, shell.layers, first=(k == 0))
        if k == 0:
            skip_ghost_copy(w, 1)          # ghost_onehot_to_feature_x
            ls = shell.layers[0]
            sdk['0_self_connection_intro.fc_tensor_product.weight'] = take_fctp(
                w, ls.irreps_x, shell.num_species, ls.irreps_gate_in)
            sdk['0_self_interaction_1.linear.weight'] = take_linear(w, ls.irreps_x, ls.irreps_x)
            skip_ghost_copy(w, 1)          # ghost_0_self_interaction_1
        else:
            assert np.allclose(sdk['edge_embedding.basis_function.coeffs'], sd['edge_embedding.basis_function.coeffs'])
        conv_part(shell.layers[k])
        if k + 1 < L:
            pre_part(shell.layers[k + 1])
        else:
            sdk['reduce_input_to_hidden.linear.weight'] = take_linear(w, shell.irreps_final, shell.irreps_hidden)
            sdk['reduce_hidden_to_energy.linear.weight'] = take_linear(w, shell.irreps_hidden, Irreps('1x0e'))
            sdk['rescale_atomic_energy.scale'] = w.take((1,))[0]
            sdk['rescale_atomic_energy.shift'] = w.take((1,))[0]
        assert w.i == len(uses), (k, w.i, len(uses))
        sd.update(sdk)
    cfg['conv_denominator'] = [float(sd[f'{t}_convolution.denominator']) for t in range(L)]
    cfg['shift'] = float(sd['rescale_atomic_energy.shift'])
    cfg['scale'] = float(sd['rescale_atomic_energy.scale'])
    return cfg, {k: v.numpy().astype(np.float32) for k, v in sd.items()}


# --------------------------------------------------------------------------- #
def read_lammps_data(path):
    with open(path) as f:
        lines = [ln.strip() for ln in f]
    n = int([ln for ln in lines if ln.endswith('atoms')][0].split()[0])
    g = lambda key: [ln for ln in lines if ln.endswith(key)][0].split()  # noqa: E731
    xlo, xhi = map(float, g('xlo xhi')[:2])
    ylo, yhi = map(float, g('ylo yhi')[:2])
    zlo, zhi = map(float, g('zlo zhi')[:2])
    xy, xz, yz = map(float, g('xy xz yz')[:3])
    cell = np.array([[xhi - xlo, 0, 0], [xy, yhi - ylo, 0], [xz, yz, zhi - zlo]])
    i0 = lines.index('Atoms') + 2
    rows = [ln.split() for ln in lines[i0:i0 + n]]
    types = np.array([int(r[1]) - 1 for r in rows])
    pos = np.array([[float(v) for v in r[2:5]] for r in rows])
    return types, pos, cell


def read_extxyz_frame(path, frame=0, type_map=None):
    with open(path) as f:
        lines = f.read().split('\n')
    i = 0
    for _ in range(frame):
        i += int(lines[i]) + 2
    n = int(lines[i])
    lat = re.search(r'Lattice="([^"]+)"', lines[i + 1]).group(1)
    cell = np.array([float(v) for v in lat.split()]).reshape(3, 3)
    sym, pos = [], []
    for ln in lines[i + 2:i + 2 + n]:
        t = ln.split()
        sym.append(t[0])
        pos.append([float(v) for v in t[1:4]])
    return np.array([type_map[s] for s in sym]), np.array(pos), cell


def rocksalt_hfo(reps=(2, 2, 2), a=4.6, sigma=0.1, seed=0):
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.5, .5, .5], [.5, 0, 0], [0, .5, 0], [0, 0, .5]])
    ty = np.array([0, 0, 0, 0, 1, 1, 1, 1])
    g = np.stack(np.meshgrid(*[np.arange(r) for r in reps], indexing='ij'), -1).reshape(-1, 3)
    pos = (g[:, None, :] + basis[None]).reshape(-1, 3) * a
    types = np.tile(ty, len(g))
    rng = np.random.default_rng(seed)
    return types, pos + rng.normal(0, sigma, pos.shape), np.diag(np.array(reps) * a).astype(float)


def run_serial(m, types, pos, cell, cutoff):
    ei, ev, S = neighbor_list(pos, cell, [True] * 3, cutoff)
    data = {
        'x': torch.tensor(types, dtype=torch.int64),
        'edge_index': torch.tensor(ei, dtype=torch.int64),
        'pos': torch.tensor(pos, dtype=torch.float32).requires_grad_(True),
        'cell_lattice_vectors': torch.tensor(cell, dtype=torch.float32),
        'pbc_shift': torch.tensor(S, dtype=torch.float32),
        'cell_volume': torch.tensor(abs(np.linalg.det(cell)), dtype=torch.float32),
        'num_atoms': torch.tensor(len(types), dtype=torch.int64),
        'batch': torch.zeros(len(types), dtype=torch.int64),
    }
    out = m(data)
    return ei, S, {
        'energy': out['inferred_total_energy'].detach().numpy(),
        'atomic_energy': out['atomic_energy'].detach().numpy().reshape(-1),
        'forces': out['inferred_force'].detach().numpy(),
        'stress': out['inferred_stress'].detach().numpy(),
        'edge_vec': out['edge_vec'].detach().numpy(),
        'edge_embedding': out['edge_embedding'].detach().numpy(),
        'edge_attr': out['edge_attr'].detach().numpy(),
        'x_final': out['x'].detach().numpy(),
    }


def run_parallel_chain(types, ei, edge_vec):
    """Chain the four deployed_parallel segments with zero ghosts, keeping every
    segment's `x` / `self_cont_tmp` -- per-layer intermediates of the reference
    (scheme of pair_e3gnn_parallel.cpp:358-390)."""
    segs = [torch.jit.load(f'{REF}/example_inputs/md_parallel_example/deployed_parallel/deployed_parallel_{i}.pt',
                           map_location='cpu') for i in range(4)]
    n = len(types)
    ev = torch.tensor(edge_vec, dtype=torch.float32).requires_grad_(True)
    data = {
        'x': torch.tensor(types, dtype=torch.int64),
        'x_ghost': torch.zeros(0, dtype=torch.int64),
        'edge_index': torch.tensor(ei, dtype=torch.int64),
        'edge_vec': ev,
        # 1-element 1-D tensors as in pair_e3gnn_parallel.cpp:311-340
        # (tensor_split(x, nlocal) must see split *indices*, not a section count)
        'num_atoms': torch.tensor([n], dtype=torch.int64),
        'nlocal': torch.tensor([n], dtype=torch.int64),
    }
    inter = {}
    for k, seg in enumerate(segs):
        data = seg(data)
        if k < 3:
            inter[f'seg{k}_x'] = data['x'].detach().numpy().copy()
            inter[f'seg{k}_self_cont_tmp'] = data['self_cont_tmp'].detach().numpy().copy()
            data['x_ghost'] = torch.zeros(0, data['x'].shape[1])
            data['edge_vec'] = ev
    e = data['inferred_total_energy']
    (g,) = torch.autograd.grad(e, ev)
    inter['par_energy'] = e.detach().numpy()
    inter['par_atomic_energy'] = data['atomic_energy'].detach().numpy().reshape(-1)
    inter['par_dE_dr'] = g.numpy()
    return inter


def main():
    os.makedirs(OUT, exist_ok=True)
    dump_cp0()
    m, cfg, sd = extract_serial()
    tm = {'Hf': 0, 'O': 1}
    systems = {
        'hfo2_96': read_lammps_data(f'{REF}/example_inputs/md_serial_example/res.dat'),
        'hfo2_12': read_extxyz_frame(f'{REF}/tests/data/systems/hfo2.extxyz', 0, tm),
        'hfo_rs64': rocksalt_hfo(),
    }
    pcfg, psd = extract_parallel()
    model = OracleModel(cfg, sd, dtype=torch.float64)
    pmodel = OracleModel(pcfg, psd, dtype=torch.float64)
    for name, (types, pos, cell) in systems.items():
        ei, S, out = run_serial(m, types, pos, cell, cfg['cutoff'])
        inter = run_parallel_chain(types, ei, out['edge_vec'])
        mine = model.forward(types, ei, out['edge_vec'].astype(np.float64))
        pmine = pmodel.forward(types, ei, out['edge_vec'].astype(np.float64), keep=True)
        dE = abs(float(mine['energy']) - float(out['energy']))
        dF = np.abs(mine['forces'].numpy() - out['forces']).max()
        dPE = abs(float(pmine['energy']) - float(inter['par_energy']))
        dP = np.abs(inter['par_dE_dr'] - pmine['dE_dr'].numpy()).max()
        dX = max(np.abs(pmodel.si1(pmodel.layers[k + 1], pmine['inter'][f'{k}_x']).numpy() - inter[f'seg{k}_x']).max()
                 for k in range(3))
        print(f'{name}: N={len(types)} E={ei.shape[1]} serial |dE|={dE:.3e} max|dF|={dF:.3e} ; '
              f'segments |dE|={dPE:.3e} max|d(dE/dr)|={dP:.3e} max|dx|={dX:.3e}')
        np.savez_compressed(
            os.path.join(OUT, f'ts_oracle_{name}.npz'),
            __config__=json.dumps(cfg), __pconfig__=json.dumps(pcfg),
            types=types.astype(np.int64), pos=pos, cell=cell,
            edge_index=ei.astype(np.int64), shifts=S,
            **{f'out_{k}': v for k, v in out.items()}, **inter,
            **{f'w::{k}': v for k, v in sd.items()}, **{f'pw::{k}': v for k, v in psd.items()})


if __name__ == '__main__':
    main()
