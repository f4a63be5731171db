"""`.snet` model file: the engine-side analogue of the reference's deployed TorchScript model.

The reference deploys a checkpoint to a frozen TorchScript archive that the C++ LAMMPS pair styles
load (sevenn/scripts/deploy.py:16-76, pair_e3gnn.cpp:308-411).  Here the deployable artefact is a
flat little-endian binary consumed by `snet_model_load()` of libsnet_hip.so (csrc/snet_model.cpp):
all tables of `ModelSpec` plus the weights with every normalisation already folded, so a native
host needs no Python, no torch and no e3nn.

Layout (all ints int32, floats float32 unless noted):
    magic 'SNETMDL4'   (v3: species tables of layer 0; unread tensor-product paths are not stored; v4: readout kind, `readout_as_fcn`)
    header  : n_species n_layers lmax normalize n_basis cutoff_kind poly_p act_radial n_scale d0
              cutoff(f32) cutoff_on(f32) act_cst(f32)
    coeffs[n_basis]  embed[n_species*d0]  scale[n_scale]  shift[n_scale]
    per layer: dx dmid gin dout wn  tag[12 bytes]  conv_scale(f32)
               mlp dims[4]  W0 W1 W2 (row-major, 1/sqrt(fan_in) folded)
               linear sc (all-zero header if absent), si1, si2   (see _write_linear)
               n_gate_segs, segs[kind in_off out_off mul l gate_off act | cst f32]
    readout kind (0 / 1); kind 0: readout linears ro1, ro2
    species tables of layer 0 (model_spec.species_only_tables): dx0, gin0 (0 = no self-connection),
               h0[n_species*dx0], sc0[n_species*gin0]  -- SI1(x) and sc(x) of the species-only layer-0 inputs, fp64-evaluated
    kind 0: folded readout (model_spec.folded_readout): d_ro, c (float64), v[d_ro] (float64)  -- e_i = x_i . v + c
    kind 1: n_fcn_layers, widths[n + 1], act id, act_cst (f32), then W_i[d_i * d_{i+1}] with 1/sqrt(d_i) folded
    metadata : n_bytes, then `key=value` lines (utf-8) -- the `_extra_files` of the reference's deployed
               model (deploy.py:56-72): chemical_symbols_to_index, cutoff, num_species, model_type,
               version, dtype
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

from .model_spec import (ACT_CST, ACT_ID, LinearSpec, build_model_spec, folded_readout, linear_modal_bias,
                         linear_weight_matrices, species_only_tables)

MAGIC = b'SNETMDL4'

_SYMBOLS = ('X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr '
            'Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W '
            'Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt '
            'Ds Rg Cn Nh Fl Mc Lv Ts Og').split()


def species_symbols(config: dict, n_species: int):
    """element symbol of every species index (deploy.py:57-61: symbols in `_type_map` key order)"""
    tm = config.get('_type_map')
    if tm:
        by_index = sorted(((int(v), int(z)) for z, v in tm.items()))
        return [_SYMBOLS[z] for _, z in by_index]
    if config.get('chemical_species'):
        return list(config['chemical_species'])[:n_species]
    return [f'X{i}' for i in range(n_species)]


def _i(*v):
    return struct.pack('<%di' % len(v), *[int(x) for x in v])


def _f(*v):
    return struct.pack('<%df' % len(v), *[float(x) for x in v])


def _arr(a):
    return np.ascontiguousarray(a, dtype='<f4').tobytes()


def _write_linear(spec: LinearSpec, flat, modal_idx: int = -1, bias_flat=None) -> bytes:
    """dim_in dim_out n_species n_blocks n_zero n_zero_in
    | blocks[l in_off mul_in out_off mul_out species accumulate]
    | zero[off len] (output columns no block writes) | zero_in[off len] (input columns no block reads)
    | weights of every block, [K,N] row-major with alpha folded
    | has_bias, then bias[dim_out] if 1 (multi-modal linear, fidelity channel fixed at deploy time)"""
    if spec is None:
        return _i(0, 0, 0, 0, 0, 0) + _i(0)
    mats = linear_weight_matrices(spec, flat)
    fed = {b.in_off for b in spec.blocks}
    zero_in = [(off, m * (2 * l + 1)) for off, (m, l, _) in zip(spec.irreps_in.offsets(), spec.irreps_in)
               if off not in fed]
    out = [_i(spec.dim_in, spec.dim_out, spec.n_species, len(spec.blocks), len(spec.zero_out), len(zero_in))]
    for b in spec.blocks:
        out.append(_i(b.l, b.in_off, b.mul_in, b.out_off, b.mul_out, b.species, int(b.accumulate)))
    for off, ln in list(spec.zero_out) + zero_in:
        out.append(_i(off, ln))
    for m in mats:
        out.append(_arr(m))
    bias = linear_modal_bias(spec, flat, modal_idx, bias_flat)   # modal one-hot share + o3.Linear bias: one constant row
    out.append(_i(0 if bias is None else 1))
    if bias is not None:
        out.append(_arr(bias))
    return b''.join(out)


def write_model_file(path: str, config: dict, state_dict: Dict[str, np.ndarray], modal=None) -> None:
    """modal: fidelity channel of a multi-modal model, fixed in the file like the reference's
    `prepare_modal_deploy` (sevenn/scripts/deploy.py:42-47)."""
    sp = build_model_spec(config)
    mi = sp.modal_index(modal)
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v, dtype=np.float64)
          for k, v in state_dict.items()}
    for k, shp in sp.param_shapes().items():
        if k not in sd:
            raise KeyError(f'state_dict is missing {k}')
        if not k.startswith('rescale_atomic_energy.'):  # rescale shapes come from the tensors (rescale_vectors)
            sd[k] = sd[k].reshape(shp)
    scale_v, shift_v = sp.rescale_vectors(sd, mi)
    n_scale = len(scale_v)
    embed = linear_weight_matrices(sp.embed, sd[sp.embed.name])[0]
    eb = linear_modal_bias(sp.embed, sd[sp.embed.name], mi, sd.get(sp.embed.bias_name))
    if eb is not None:
        embed = embed + eb[None, :]
    out = [MAGIC,
           _i(sp.num_species, len(sp.layers), sp.lmax_edge, int(sp.normalize_sph), sp.n_basis, sp.cutoff_kind,
              sp.cutoff_p, ACT_ID[sp.act_radial], n_scale, sp.embed.dim_out),
           _f(sp.cutoff, sp.cutoff_on, ACT_CST[sp.act_radial]),
           _arr(sd['edge_embedding.basis_function.coeffs']), _arr(embed),
           _arr(scale_v), _arr(shift_v)]
    inv_act = {v: k for k, v in ACT_ID.items()}
    for ls in sp.layers:
        d = ls.mlp_dims
        if len(d) != 4:
            raise NotImplementedError('.snet files need a radial MLP with two hidden layers')
        out.append(_i(ls.si1.dim_out, ls.conv.irreps_out.dim, ls.gate.irreps_in.dim, ls.gate.irreps_out.dim,
                      ls.conv.weight_numel))
        out.append(ls.conv.tag.encode()[:12].ljust(12, b'\0'))
        out.append(_f(1.0 / float(sd[f'{ls.t}_convolution.denominator'][0])))
        out.append(_i(*d))
        rw = ls.radial_weights(sd)   # the last layer's columns of paths nothing reads are not stored
        for i in range(3):
            out.append(_arr(rw[i] / np.sqrt(d[i])))
        out.append(_write_linear(ls.sc, sd[ls.sc.name] if ls.sc is not None else None))
        out.append(_write_linear(ls.si1, sd[ls.si1.name], mi, sd.get(ls.si1.bias_name)))
        out.append(_write_linear(ls.si2, sd[ls.si2.name], mi, sd.get(ls.si2.bias_name)))
        out.append(_i(len(ls.gate.segs)))
        for s in ls.gate.segs:
            out.append(_i(s.kind, s.in_off, s.out_off, s.mul, s.l, s.gate_off, s.act))
            out.append(_f(ACT_CST[inv_act[s.act]]))
    fcn = sp.readout_fcn_dims
    out.append(_i(1 if fcn else 0))   # readout kind: 0 = two linears (stored folded below), 1 = `readout_as_fcn` (nn/linear.py:145-180)
    if not fcn:
        out.append(_write_linear(sp.readout1, sd[sp.readout1.name], mi, sd.get(sp.readout1.bias_name)))
        out.append(_write_linear(sp.readout2, sd[sp.readout2.name], -1, sd.get(sp.readout2.bias_name)))
    h0, sc0 = species_only_tables(sp, sd, mi)
    out.append(_i(h0.shape[1], 0 if sc0 is None else sc0.shape[1]))
    out.append(_arr(h0))
    if sc0 is not None:
        out.append(_arr(sc0))
    if fcn:   # layer count, widths, activation, then W_i / sqrt(fan_in) as [d_i, d_{i+1}] (what engine.py multiplies with)
        out.append(_i(len(fcn) - 1, *fcn, ACT_ID[sp.readout_fcn_act]))
        out.append(_f(ACT_CST[sp.readout_fcn_act]))
        for i in range(len(fcn) - 1):
            out.append(_arr(sd[f'readout_FCN.fcn.layer{i}.weight'].reshape(fcn[i], fcn[i + 1]) / np.sqrt(fcn[i])))
    else:
        v, c = folded_readout(sp, sd, mi)
        out.append(_i(len(v)))
        out.append(struct.pack('<d', c))
        out.append(np.ascontiguousarray(v, dtype='<f8').tobytes())
    meta = {'chemical_symbols_to_index': ' '.join(species_symbols(config, sp.num_species)),
            'cutoff': repr(float(sp.cutoff)), 'num_species': str(sp.num_species),
            'model_type': str(config.get('model_type', 'E3_equivariant_model')),
            'version': str(config.get('version', '')), 'dtype': 'single'}
    text = ''.join(f'{k}={v}\n' for k, v in meta.items()).encode()
    out.append(_i(len(text)))
    out.append(text)
    with open(path, 'wb') as f:
        f.write(b''.join(out))
