"""CPU restatement of the SevenNet energy/force model.  TEST INFRASTRUCTURE ONLY.

Each function cites the reference code it follows.  Weight tensors use the
reference's own state-dict names (tests/data/checkpoints/cp_0.pth), so a
reference checkpoint's `model_state_dict` drops in unchanged.

Arithmetic mirrors e3nn's lowering (gather -> per-path einsum with dense real
Wigner-3j -> scatter_add -> torch.autograd for forces), which is what the
reference executes on CPU; this is also the "reference CPU PyTorch path" timed
by bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from .e3 import (Irreps, infer_irreps_out, irrep_product, normalize2mom_const,
                 spherical_harmonics, wigner_3j)

DEFAULTS = dict(  # sevenn/_const.py:95-135
    cutoff=4.5, channel=32, irreps_manual=False, lmax=1, lmax_edge=-1, lmax_node=-1,
    is_parity=True, num_convolution_layer=3,
    radial_basis={'radial_basis_name': 'bessel'},
    cutoff_function={'cutoff_function_name': 'poly_cut'},
    act_radial='silu', act_scalar={'e': 'silu', 'o': 'tanh'},
    act_gate={'e': 'silu', 'o': 'tanh'}, weight_nn_hidden_neurons=[64, 64],
    conv_denominator=1.0, self_connection_type='nequip', _normalize_sph=True,
    shift=0.0, scale=1.0, version='0.12.0', use_bias_in_linear=False, readout_as_fcn=False,
    readout_fcn_hidden_neurons=[30, 30], readout_fcn_activation='relu',
)


def _version_tuple(v: str):
    return tuple(int(t) for t in v.split('.')[:3] if t.isdigit())


def _act(name: str):
    """sevenn/_const.py:33-47 (ssp: sevenn/nn/activation.py:7)"""
    return {'silu': torch.nn.functional.silu, 'tanh': torch.tanh, 'relu': torch.relu, 'abs': torch.abs,
            'ssp': lambda x: torch.nn.functional.softplus(x) - math.log(2.0), 'sigmoid': torch.sigmoid,
            'elu': torch.nn.functional.elu}[name]


# --------------------------------------------------------------------------- #
# layer arithmetic
# --------------------------------------------------------------------------- #
def bessel_basis(r, coeffs, rc):
    """sevenn/nn/edge_embedding.py:101-103"""
    ur = r.unsqueeze(-1)
    return (2.0 / rc) * torch.sin(coeffs * ur) / ur


def poly_cutoff(r, rc, p=6):
    """sevenn/nn/edge_embedding.py:125-132"""
    x = r / rc
    return (1 - (p + 1.0) * (p + 2.0) / 2.0 * torch.pow(x, p)
            + p * (p + 2.0) * torch.pow(x, p + 1.0)
            - p * (p + 1.0) / 2.0 * torch.pow(x, p + 2.0))


def xplor_cutoff(r, rc, r_on):
    """sevenn/nn/edge_embedding.py:150-160"""
    r_sq, on_sq, c_sq = r * r, r_on * r_on, rc * rc
    return torch.where(r < r_on, torch.ones_like(r),
                       (c_sq - r_sq) ** 2 * (c_sq + 2 * r_sq - 3 * on_sq) / (c_sq - on_sq) ** 3)


def fcn_apply(x, weights: List[torch.Tensor], act_name: str):
    """e3nn.nn.FullyConnectedNet as used at sevenn/nn/convolution.py:93-95,121."""
    cst = normalize2mom_const(act_name)
    act = _act(act_name)
    for i, w in enumerate(weights):
        x = x @ (w / math.sqrt(w.shape[0]))
        if i + 1 < len(weights):
            x = act(x) * cst
    return x


def linear_instructions(irreps_in: Irreps, irreps_out: Irreps):
    return [(i, j) for i, (_, a) in enumerate(irreps_in)
            for j, (_, b) in enumerate(irreps_out) if a == b]


def linear_weight_numel(irreps_in, irreps_out):
    return sum(irreps_in[i][0] * irreps_out[j][0] for i, j in linear_instructions(irreps_in, irreps_out))


def linear_bias_numel(irreps_out: Irreps) -> int:
    return sum(mo for mo, (l, p) in irreps_out if (l, p) == (0, 1))


def linear_apply(x, irreps_in: Irreps, irreps_out: Irreps, w_flat, bias=None):
    """e3nn o3.Linear (sevenn/nn/linear.py:94-100): per matching irrep pair a
    [mul_in, mul_out] block, scaled 1/sqrt(total fan-in of that output block).
    bias (o3.Linear(biases=True), `use_bias_in_linear`): one value per channel of every 0e output block, concatenated in
    irreps_out order, added unscaled."""
    ins = linear_instructions(irreps_in, irreps_out)
    sl_in, sl_out = irreps_in.slices(), irreps_out.slices()
    fan = [0] * len(irreps_out)
    for i, j in ins:
        fan[j] += irreps_in[i][0]
    outs = [None] * len(irreps_out)
    o = 0
    for i, j in ins:
        mi, (l, _) = irreps_in[i]
        mo = irreps_out[j][0]
        w = w_flat[o:o + mi * mo].reshape(mi, mo)
        o += mi * mo
        xi = x[:, sl_in[i]].reshape(-1, mi, 2 * l + 1)
        y = torch.einsum('nui,uv->nvi', xi, w) / math.sqrt(fan[j])
        outs[j] = y if outs[j] is None else outs[j] + y
    assert o == w_flat.numel()
    cols, b_off = [], 0
    for j, (mo, (l, p_)) in enumerate(irreps_out):
        if outs[j] is None:
            c = x.new_zeros(x.shape[0], mo * (2 * l + 1))
        else:
            c = outs[j].reshape(x.shape[0], -1)
        if bias is not None and (l, p_) == (0, 1):
            c = c + bias[b_off:b_off + mo]
            b_off += mo
        cols.append(c)
    assert bias is None or b_off == bias.numel()
    return torch.cat(cols, dim=1)


def fctp_instructions(irreps_in: Irreps, irreps_out: Irreps):
    return [(i, j) for i, (_, a) in enumerate(irreps_in)
            for j, (_, b) in enumerate(irreps_out) if a == b]


def fctp_weight_numel(irreps_in, n_species, irreps_out):
    return sum(irreps_in[i][0] * n_species * irreps_out[j][0]
               for i, j in fctp_instructions(irreps_in, irreps_out))


def fctp_apply(x, onehot, irreps_in: Irreps, irreps_out: Irreps, w_flat):
    """e3nn FullyConnectedTensorProduct(x, Nx0e -> out), 'uvw'
    (sevenn/nn/self_connection.py:11-67).  Coefficient 1/sqrt(sum mul_in*N)."""
    n_sp = onehot.shape[1]
    ins = fctp_instructions(irreps_in, irreps_out)
    sl_in = irreps_in.slices()
    fan = [0] * len(irreps_out)
    for i, j in ins:
        fan[j] += irreps_in[i][0] * n_sp
    outs = [None] * len(irreps_out)
    o = 0
    for i, j in ins:
        mi, (l, _) = irreps_in[i]
        mo = irreps_out[j][0]
        w = w_flat[o:o + mi * n_sp * mo].reshape(mi, n_sp, mo)
        o += mi * n_sp * mo
        xi = x[:, sl_in[i]].reshape(-1, mi, 2 * l + 1)
        y = torch.einsum('nui,nv,uvw->nwi', xi, onehot, w) / math.sqrt(fan[j])
        outs[j] = y if outs[j] is None else outs[j] + y
    assert o == w_flat.numel()
    cols = []
    for j, (mo, (l, _)) in enumerate(irreps_out):
        cols.append(x.new_zeros(x.shape[0], mo * (2 * l + 1)) if outs[j] is None
                    else outs[j].reshape(x.shape[0], -1))
    return torch.cat(cols, dim=1)


def conv_instructions(irreps_x: Irreps, irreps_filter: Irreps, irreps_out: Irreps, sort_by_out: bool):
    """sevenn/nn/convolution.py:61-82.  Returns (irreps_mid sorted, instructions
    [(i_x, i_filter, i_mid)], weight_numel).  Weight columns follow the
    instruction list order, mul_x per instruction."""
    ins, mid = [], []
    for i, (mul_x, ir_x) in enumerate(irreps_x):
        for j, (_, ir_f) in enumerate(irreps_filter):
            for ir_o in irrep_product(ir_x, ir_f):
                if ir_o in irreps_out:
                    ins.append((i, j, len(mid)))
                    mid.append((mul_x, ir_o))
    mid_sorted, p, _ = Irreps(mid).sort()
    ins = [(i, j, p[k]) for i, j, k in ins]
    if sort_by_out:  # v0.11+
        ins = sorted(ins, key=lambda t: t[2])
    wn = sum(irreps_x[i][0] for i, _, _ in ins)
    return mid_sorted, ins, wn


def tp_uvu(x_src, sh, weight, irreps_x: Irreps, irreps_sh: Irreps, irreps_mid: Irreps, ins):
    """e3nn TensorProduct('uvu', shared_weights=False) per edge
    (sevenn/nn/convolution.py:84-91,131): out[u,k] = sqrt(2l3+1) w[u] sum_ij
    C[i,j,k] x[u,i] Y[j]."""
    E = x_src.shape[0]
    sl_x, sl_sh = irreps_x.slices(), irreps_sh.slices()
    outs: List[Optional[torch.Tensor]] = [None] * len(irreps_mid)
    o = 0
    for i, j, k in ins:
        mul, (l1, _) = irreps_x[i]
        l2 = irreps_sh[j][1][0]
        l3 = irreps_mid[k][1][0]
        w = weight[:, o:o + mul]
        o += mul
        C = wigner_3j(l1, l2, l3, dtype=x_src.dtype).to(x_src.device) * math.sqrt(2 * l3 + 1)
        xi = x_src[:, sl_x[i]].reshape(E, mul, 2 * l1 + 1)
        yj = sh[:, sl_sh[j]]
        # pairwise contractions like e3nn's lowering: (x (x) Y) @ C, then the per-edge weight
        xy = (xi.unsqueeze(-1) * yj.reshape(E, 1, 1, 2 * l2 + 1)).reshape(E, mul, (2 * l1 + 1) * (2 * l2 + 1))
        outs[k] = ((xy @ C.reshape(-1, 2 * l3 + 1)) * w.unsqueeze(-1)).reshape(E, mul * (2 * l3 + 1))
    assert o == weight.shape[1]
    return torch.cat(outs, dim=1)


class GateSpec:
    """e3nn nn.Gate as configured by sevenn/nn/equivariant_gate.py:26-49."""

    def __init__(self, irreps_x: Irreps, act_scalar: Dict[str, str], act_gate: Dict[str, str]):
        pm = {1: 'e', -1: 'o'}
        self.scalars = Irreps([(m, ir) for m, ir in irreps_x if ir[0] == 0])
        self.gated = Irreps([(m, ir) for m, ir in irreps_x if ir[0] > 0])
        gp = 1 if (0, 1) in self.scalars else -1
        self.gates = Irreps([(m, (0, gp)) for m, _ in self.gated])
        self.act_scalars = [act_scalar[pm[p]] for _, (_, p) in self.scalars]
        self.act_gates = [act_gate[pm[p]] for _, (_, p) in self.gates]
        cat = Irreps(self.scalars.items + self.gates.items + self.gated.items)
        srt, p, inv = cat.sort()
        self.irreps_in = srt.simplify()
        self.irreps_out = Irreps(self.scalars.items + self.gated.items)
        # column ranges (in the sorted input) of each original block
        starts, o = [], 0
        for m, (l, _) in srt:
            starts.append(o)
            o += m * (2 * l + 1)
        self._cols = []
        for old, (m, (l, _)) in enumerate(cat):
            s = starts[p[old]]
            self._cols.append((s, s + m * (2 * l + 1)))
        self.ns, self.ng = len(self.scalars), len(self.gates)

    def apply(self, x):
        outs = []
        for b in range(self.ns):
            s, e = self._cols[b]
            a = self.act_scalars[b]
            outs.append(_act(a)(x[:, s:e]) * normalize2mom_const(a))
        for b in range(self.ng):
            s, e = self._cols[self.ns + b]
            a = self.act_gates[b]
            g = _act(a)(x[:, s:e]) * normalize2mom_const(a)
            s2, e2 = self._cols[self.ns + self.ng + b]
            m, (l, _) = self.gated[b]
            outs.append((x[:, s2:e2].reshape(-1, m, 2 * l + 1) * g.unsqueeze(-1)).reshape(x.shape[0], -1))
        return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------- #
# model
# --------------------------------------------------------------------------- #
class LayerSpec:
    pass


class OracleModel:
    """config + state_dict -> energy/forces.  Follows
    sevenn/model_build.py:448-636 and sevenn/nn/interaction_blocks.py:41-76."""

    def __init__(self, config: dict, state_dict: Optional[dict] = None, dtype=torch.float32, modal=None):
        """modal: name (key of config['_modal_map']) or index of the fidelity channel of a multi-modal
        model (sevenn/model_build.py:185-231); required iff config['use_modality']."""
        cfg = dict(DEFAULTS)
        cfg.update(config)
        self.cfg = cfg
        self.dtype = dtype
        # multi-modal models: a one-hot of the fidelity channel is appended as extra 0e inputs of the
        # flagged linears (sevenn/nn/linear.py:66-92) and shift/scale may carry a modal axis
        self.n_modal = int(cfg.get('_number_of_modalities', 0)) if cfg.get('use_modality') else 0
        self.modal_idx = None
        if self.n_modal:
            if self.n_modal < 2:
                raise ValueError('use_modality needs _number_of_modalities >= 2')
            if modal is None:
                raise ValueError('multi-modal model: a modal must be given')
            self.modal_idx = int(cfg['_modal_map'][modal]) if isinstance(modal, str) else int(modal)
            if not 0 <= self.modal_idx < self.n_modal:
                raise ValueError(f'modal index {self.modal_idx} out of range')
        self.modal_in = {k: bool(self.n_modal and cfg.get(f, False)) for k, f in (
            ('embed', 'use_modal_node_embedding'), ('si1', 'use_modal_self_inter_intro'),
            ('si2', 'use_modal_self_inter_outro'), ('out', 'use_modal_output_block'))}
        self.cutoff = float(cfg['cutoff'])
        self.num_species = int(cfg.get('_number_of_species', cfg.get('num_species', 0))
                               or len(cfg['chemical_species']))
        ch = int(cfg['channel'])
        L = int(cfg['num_convolution_layer'])
        self.L = L
        lmax_edge = cfg['lmax_edge'] if cfg['lmax_edge'] > 0 else cfg['lmax']
        lmax_node = cfg['lmax_node'] if cfg['lmax_node'] > 0 else cfg['lmax']
        self.lmax_edge = lmax_edge
        parity = -1 if cfg['is_parity'] else 1
        self.irreps_filter = Irreps.spherical_harmonics(lmax_edge, parity)
        self.normalize_sph = bool(cfg['_normalize_sph'])
        self.n_basis = int(cfg['radial_basis'].get('bessel_basis_num', 8))
        cf = cfg['cutoff_function']
        self.cut_name = cf['cutoff_function_name']
        self.cut_p = int(cf.get('poly_cut_p_value', 6))
        self.cut_on = float(cf.get('cutoff_on', 0.0))
        self.act_radial = cfg['act_radial']
        sort_by_out = _version_tuple(str(cfg['version'])) >= (0, 11, 0)
        manual = cfg['irreps_manual']
        if manual is not False:
            manual = [Irreps(s) for s in manual]
            assert len(manual) == L + 1
        sc_types = cfg['self_connection_type']
        if isinstance(sc_types, str):
            sc_types = [sc_types] * L
        denom = cfg['conv_denominator']
        if not isinstance(denom, (list, tuple)):
            denom = [denom] * L
        hidden = list(cfg['weight_nn_hidden_neurons'])

        # sevenn<=0.8 (the deployed example models): last layer keeps full
        # irreps and the readout hidden width is channel//2
        legacy = bool(cfg.get('_legacy_v08', False))
        irreps_x = Irreps(f'{ch}x0e') if manual is False else manual[0]
        self.irreps_embed = irreps_x
        self.layers: List[LayerSpec] = []
        for t in range(L):
            parity_mode = 'full'
            if t == L - 1 and not legacy:
                lmax_node = 0
                parity_mode = 'even'
            if manual is False:
                irreps_out = infer_irreps_out(irreps_x, self.irreps_filter, lmax_node, parity_mode, ch)
            else:
                irreps_out = manual[t + 1]
            irreps_out_tp = infer_irreps_out(irreps_x, self.irreps_filter, irreps_out.lmax, parity_mode, False)
            ls = LayerSpec()
            ls.t = t
            ls.irreps_x = irreps_x
            ls.irreps_out = irreps_out
            ls.irreps_out_tp = irreps_out_tp
            ls.gate = GateSpec(irreps_out, cfg['act_scalar'], cfg['act_gate'])
            ls.irreps_gate_in = ls.gate.irreps_in
            ls.irreps_mid, ls.ins, ls.wn = conv_instructions(irreps_x, self.irreps_filter, irreps_out_tp, sort_by_out)
            assert ls.irreps_mid.dim == irreps_out_tp.dim
            ls.sc_type = sc_types[t]
            ls.denominator = float(denom[t])
            ls.mlp_dims = [self.n_basis] + hidden + [ls.wn]
            self.layers.append(ls)
            irreps_x = irreps_out
        self.irreps_final = irreps_x
        self.irreps_hidden = Irreps([((ch if legacy else irreps_x.dim) // 2, (0, 1))])
        self.use_bias = bool(cfg.get('use_bias_in_linear', False))
        self.readout_fcn_dims = ([irreps_x.dim] + [int(v) for v in cfg.get('readout_fcn_hidden_neurons', [30, 30])] + [1]
                                 if cfg.get('readout_as_fcn') else None)
        self.readout_fcn_act = str(cfg.get('readout_fcn_activation', 'relu'))

        self.p: Dict[str, torch.Tensor] = OrderedDict()
        shapes = self.param_shapes()
        if state_dict is None:  # structure-only shell (shapes, irreps)
            return
        for k, shp in shapes.items():
            if k not in state_dict:
                raise KeyError(f'missing parameter {k}')
            v = torch.as_tensor(np.asarray(state_dict[k]), dtype=dtype).reshape(shp)
            self.p[k] = v

    # ---------------------------------------------------------------- modal plumbing
    def _mi(self, irreps: Irreps, which: str) -> Irreps:
        """input irreps of a (possibly modal-patched) linear: + Mx0e (linear.py:66-71)"""
        return irreps + Irreps(f'{self.n_modal}x0e') if self.modal_in[which] else irreps

    def _mx(self, x, which: str):
        """append the modal one-hot columns to every row (linear.py:85-92, unbatched branch)"""
        if not self.modal_in[which]:
            return x
        oh = torch.zeros(self.n_modal, dtype=x.dtype)
        oh[self.modal_idx] = 1.0
        return torch.cat([x, oh.expand(x.shape[0], -1)], dim=1)

    # ---------------------------------------------------------------- shapes
    def param_shapes(self) -> Dict[str, tuple]:
        s = OrderedDict()
        s['edge_embedding.basis_function.coeffs'] = (self.n_basis,)
        s['onehot_to_feature_x.linear.weight'] = (
            linear_weight_numel(self._mi(Irreps(f'{self.num_species}x0e'), 'embed'), self.irreps_embed),)
        for ls in self.layers:
            t = ls.t
            if ls.sc_type == 'nequip':
                s[f'{t}_self_connection_intro.fc_tensor_product.weight'] = (
                    fctp_weight_numel(ls.irreps_x, self.num_species, ls.irreps_gate_in),)
            elif ls.sc_type == 'linear':
                s[f'{t}_self_connection_intro.linear.weight'] = (
                    linear_weight_numel(ls.irreps_x, ls.irreps_gate_in),)
            s[f'{t}_self_interaction_1.linear.weight'] = (linear_weight_numel(self._mi(ls.irreps_x, 'si1'), ls.irreps_x),)
            s[f'{t}_convolution.denominator'] = (1,)
            for i in range(len(ls.mlp_dims) - 1):
                s[f'{t}_convolution.weight_nn.layer{i}.weight'] = (ls.mlp_dims[i], ls.mlp_dims[i + 1])
            s[f'{t}_self_interaction_2.linear.weight'] = (
                linear_weight_numel(self._mi(ls.irreps_out_tp, 'si2'), ls.irreps_gate_in),)
        if self.readout_fcn_dims:   # readout_as_fcn (model_build.py:124-138)
            for i in range(len(self.readout_fcn_dims) - 1):
                s[f'readout_FCN.fcn.layer{i}.weight'] = (self.readout_fcn_dims[i], self.readout_fcn_dims[i + 1])
        else:
            s['reduce_input_to_hidden.linear.weight'] = (
                linear_weight_numel(self._mi(self.irreps_final, 'out'), self.irreps_hidden),)
            s['reduce_hidden_to_energy.linear.weight'] = (linear_weight_numel(self.irreps_hidden, Irreps('1x0e')),)
        if self.use_bias:   # use_bias_in_linear: embedding, SI1 / SI2, readout linears (model_build.py:518,531; interaction_blocks.py:50,72)
            s['onehot_to_feature_x.linear.bias'] = (linear_bias_numel(self.irreps_embed),)
            for ls in self.layers:
                s[f'{ls.t}_self_interaction_1.linear.bias'] = (linear_bias_numel(ls.irreps_x),)
                s[f'{ls.t}_self_interaction_2.linear.bias'] = (linear_bias_numel(ls.irreps_gate_in),)
            if not self.readout_fcn_dims:
                s['reduce_input_to_hidden.linear.bias'] = (linear_bias_numel(self.irreps_hidden),)
                s['reduce_hidden_to_energy.linear.bias'] = (1,)
        if self.n_modal:  # ModalWiseRescale (scale.py:196-363): always per species, optionally per modal
            ns = self.num_species
            s['rescale_atomic_energy.shift'] = (self.n_modal, ns) if self.cfg.get('use_modal_wise_shift') else (ns,)
            s['rescale_atomic_energy.scale'] = (self.n_modal, ns) if self.cfg.get('use_modal_wise_scale') else (ns,)
            return s
        nsc = np.asarray(self.cfg['shift']).size
        nsl = np.asarray(self.cfg['scale']).size
        n = max(nsc, nsl)
        s['rescale_atomic_energy.shift'] = (n,)
        s['rescale_atomic_energy.scale'] = (n,)
        return s

    def num_weights(self) -> int:
        """Trainable-parameter count as pinned by tests/unit_tests/test_model.py:164-182
        (denominator, shift, scale are non-trainable by default)."""
        n = 0
        for k, shp in self.param_shapes().items():
            if k.endswith('denominator') or k.startswith('rescale'):
                continue
            n += int(np.prod(shp))
        return n

    # ---------------------------------------------------------------- pieces
    def edge_embedding(self, edge_vec):
        """sevenn/nn/edge_embedding.py:207-217"""
        r = torch.linalg.norm(edge_vec, dim=-1)
        basis = bessel_basis(r, self.p['edge_embedding.basis_function.coeffs'], self.cutoff)
        if self.cut_name == 'poly_cut':
            env = poly_cutoff(r, self.cutoff, self.cut_p)
        elif self.cut_name == 'XPLOR':
            env = xplor_cutoff(r, self.cutoff, self.cut_on)
        else:
            raise ValueError(self.cut_name)
        emb = basis * env.unsqueeze(-1)
        sh = spherical_harmonics(self.lmax_edge, edge_vec, self.normalize_sph)
        return emb, sh

    def node_embed(self, types):
        onehot = torch.nn.functional.one_hot(types, self.num_species).to(self.dtype)
        x = linear_apply(self._mx(onehot, 'embed'), self._mi(Irreps(f'{self.num_species}x0e'), 'embed'), self.irreps_embed,
                         self.p['onehot_to_feature_x.linear.weight'], self.p.get('onehot_to_feature_x.linear.bias'))
        return onehot, x

    def sc_intro(self, ls, x, onehot):
        t = ls.t
        if ls.sc_type == 'nequip':
            return fctp_apply(x, onehot, ls.irreps_x, ls.irreps_gate_in,
                              self.p[f'{t}_self_connection_intro.fc_tensor_product.weight'])
        if ls.sc_type == 'linear':
            return linear_apply(x, ls.irreps_x, ls.irreps_gate_in,
                                self.p[f'{t}_self_connection_intro.linear.weight'])
        return None

    def si1(self, ls, x):
        return linear_apply(self._mx(x, 'si1'), self._mi(ls.irreps_x, 'si1'), ls.irreps_x,
                            self.p[f'{ls.t}_self_interaction_1.linear.weight'], self.p.get(f'{ls.t}_self_interaction_1.linear.bias'))

    def radial_weights(self, ls, emb):
        ws = [self.p[f'{ls.t}_convolution.weight_nn.layer{i}.weight'] for i in range(len(ls.mlp_dims) - 1)]
        return fcn_apply(emb, ws, self.act_radial)

    def conv(self, ls, x_all, emb, sh, edge_src, edge_dst, n_out, weight=None):
        """sevenn/nn/convolution.py:118-141 (x_all may include ghost rows)."""
        if weight is None:
            weight = self.radial_weights(ls, emb)
        msg = tp_uvu(x_all[edge_src], sh, weight, ls.irreps_x, self.irreps_filter, ls.irreps_mid, ls.ins)
        out = x_all.new_zeros(x_all.shape[0], msg.shape[1])
        out.index_add_(0, edge_dst, msg)
        out = out / self.p[f'{ls.t}_convolution.denominator']
        return out[:n_out]

    def si2(self, ls, x):
        return linear_apply(self._mx(x, 'si2'), self._mi(ls.irreps_out_tp, 'si2'), ls.irreps_gate_in,
                            self.p[f'{ls.t}_self_interaction_2.linear.weight'], self.p.get(f'{ls.t}_self_interaction_2.linear.bias'))

    def readout(self, x, types):
        if self.readout_fcn_dims:   # FCN_e3nn (nn/linear.py:145-180): FullyConnectedNet on the final scalars
            ws = [self.p[f'readout_FCN.fcn.layer{i}.weight'] for i in range(len(self.readout_fcn_dims) - 1)]
            e = fcn_apply(x, ws, self.readout_fcn_act)
        else:
            h = linear_apply(self._mx(x, 'out'), self._mi(self.irreps_final, 'out'), self.irreps_hidden,
                             self.p['reduce_input_to_hidden.linear.weight'], self.p.get('reduce_input_to_hidden.linear.bias'))
            e = linear_apply(h, self.irreps_hidden, Irreps('1x0e'), self.p['reduce_hidden_to_energy.linear.weight'],
                             self.p.get('reduce_hidden_to_energy.linear.bias'))
        sc, sh = self.p['rescale_atomic_energy.scale'], self.p['rescale_atomic_energy.shift']
        return rescale_apply(e, types, sc, sh, self.modal_idx if self.n_modal else None)

    # ------------------------------------------------------- one brick of a decomposition
    def forward_brick(self, types_all, edge_index, edge_vec, n_local, exchange):
        """One rank of the reference's parallel scheme (model_build.py:361-431,
        pair_e3gnn_parallel.cpp:358-441): `types_all` = local then ghost species;
        edges have local centers; `exchange(h_local) -> h_all` must return the
        conv input with ghost rows filled from their owners and be differentiable
        (its backward accumulates ghost-row gradients into the owners).  Returns the
        local energy sum, local atomic energies, dE/dr of the local edges and the
        force/virial rows of all n_total atoms (ghost rows still to be folded)."""
        types_all = torch.as_tensor(types_all, dtype=torch.long)
        edge_index = torch.as_tensor(edge_index, dtype=torch.long)
        edge_vec = torch.as_tensor(edge_vec, dtype=self.dtype).clone().requires_grad_(True)
        nt = types_all.shape[0]
        emb, sh = self.edge_embedding(edge_vec)
        onehot_all, x_all = self.node_embed(types_all)
        onehot = onehot_all[:n_local]
        src, dst = edge_index[1], edge_index[0]
        x = x_all[:n_local]
        for ls in self.layers:
            sc = self.sc_intro(ls, x, onehot)
            if ls.t == 0:
                h_all = self.si1(ls, x_all)  # ghost layer-0 features depend on species only
            else:
                h_all = exchange(self.si1(ls, x))
            m = self.conv(ls, h_all, emb, sh, src, dst, n_local)
            y = self.si2(ls, m)
            if sc is not None:
                y = y + sc
            x = ls.gate.apply(y)
        e_atom = self.readout(x, types_all[:n_local])
        energy = e_atom.sum()
        (g,) = torch.autograd.grad(energy, edge_vec, allow_unused=True)
        if g is None:
            g = torch.zeros_like(edge_vec)
        out = force_virial_from_edge(g, edge_vec.detach(), edge_index, nt)
        out.update(energy=energy.detach(), atomic_energy=e_atom.detach().squeeze(-1), dE_dr=g)
        return out

    # ---------------------------------------------------------------- forward
    def forward(self, types, edge_index, edge_vec, keep=False):
        """Serial evaluation.  edge_index[0] = center (dst), [1] = neighbor (src);
        edge_vec = r_src - r_dst (+ image).  Returns dict with energy, atomic
        energies, dE/d edge_vec, forces, virial(6: xx,yy,zz,xy,yz,zx)."""
        types = torch.as_tensor(types, dtype=torch.long)
        edge_index = torch.as_tensor(edge_index, dtype=torch.long)
        edge_vec = torch.as_tensor(edge_vec, dtype=self.dtype).clone().requires_grad_(True)
        N = types.shape[0]
        inter = OrderedDict()
        emb, sh = self.edge_embedding(edge_vec)
        onehot, x = self.node_embed(types)
        src, dst = edge_index[1], edge_index[0]
        if keep:
            inter['edge_embedding'], inter['edge_attr'], inter['x_embed'] = emb, sh, x
        for ls in self.layers:
            sc = self.sc_intro(ls, x, onehot)
            h = self.si1(ls, x)
            m = self.conv(ls, h, emb, sh, src, dst, N)
            y = self.si2(ls, m)
            if sc is not None:
                y = y + sc
            x = ls.gate.apply(y)
            if keep:
                inter[f'{ls.t}_si1'], inter[f'{ls.t}_conv'], inter[f'{ls.t}_gate_in'], inter[f'{ls.t}_x'] = h, m, y, x
        e_atom = self.readout(x, types)
        energy = e_atom.sum()
        (g,) = torch.autograd.grad(energy, edge_vec, allow_unused=True)
        if g is None:
            g = torch.zeros_like(edge_vec)
        out = force_virial_from_edge(g, edge_vec.detach(), edge_index, N)
        out.update(energy=energy.detach(), atomic_energy=e_atom.detach().squeeze(-1), dE_dr=g)
        if keep:
            out['inter'] = {k: v.detach() for k, v in inter.items()}
        return out


def rescale_apply(e, types, scale, shift, modal_idx=None):
    """Rescale (sevenn/nn/scale.py:53-56), SpeciesWiseRescale (:155-162), ModalWiseRescale (:341-363):
    e[N,1] -> e * scale[(modal,) type] + shift[(modal,) type]; a 2-D table is indexed by the modal first."""
    if modal_idx is not None:
        scale = scale[modal_idx] if scale.dim() == 2 else scale
        shift = shift[modal_idx] if shift.dim() == 2 else shift
        return e * scale[types].view(-1, 1) + shift[types].view(-1, 1)
    if scale.numel() == 1:
        return e * scale + shift
    return e * scale[types].view(-1, 1) + shift[types].view(-1, 1)


def force_virial_from_edge(g, rij, edge_index, n_atoms):
    """sevenn/nn/force_output.py:171-230: F = scatter(idx0, g) - scatter(idx1, g);
    atomic virial = -sum_{e: idx1=i} (r*g, r_x g_y, r_y g_z, r_z g_x)."""
    F = g.new_zeros(n_atoms, 3)
    F.index_add_(0, edge_index[0], g)
    F.index_add_(0, edge_index[1], -g)
    vir = torch.cat([rij * g, (rij[:, 0] * g[:, 1]).unsqueeze(-1),
                     (rij[:, 1] * g[:, 2]).unsqueeze(-1), (rij[:, 2] * g[:, 0]).unsqueeze(-1)], dim=-1)
    s = g.new_zeros(n_atoms, 6)
    s.index_add_(0, edge_index[1], vir)
    return dict(forces=F, atomic_virial=-s, virial=-s.sum(0))
