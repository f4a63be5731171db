"""Model description for the HIP force engine: reference config -> op specs.

Reproduces the layer/irreps structure the reference builds in
sevenn/model_build.py:448-636 and sevenn/nn/interaction_blocks.py:41-76, but
expressed as flat tables for GPU kernels operating on the engine's `ir_mul`
feature layout:

  * LinearSpec   -- o3.Linear / per-species FCTP as a list of per-irrep GEMMs
                    (sevenn/nn/linear.py:94-100, self_connection.py:11-114)
  * ConvSpec     -- 'uvu' tensor-product paths, weight-column offsets, merged
                    output blocks (sevenn/nn/convolution.py:61-82)
  * GateSpec     -- scalar / gate / gated segments of e3nn Gate
                    (sevenn/nn/equivariant_gate.py:26-49)

Parameter names are the reference checkpoint's state_dict keys so that
`checkpoint['model_state_dict']` can be fed to the engine unchanged.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .irreps import Irreps, infer_irreps_out

# activation ids of the kernels (csrc/snet_common.h::act_fwd): every entry of sevenn/_const.py:33-47
ACT_ID = {'silu': 0, 'tanh': 1, 'relu': 2, 'abs': 3, 'ssp': 4, 'sigmoid': 5, 'elu': 6}
# e3nn normalize2mom constants: silu / tanh as baked into the reference's deployed models; the others are what e3nn's
# `moment(f, 2) ** -0.5` evaluates to (1e6 float64 samples of torch's CPU generator seeded with 0 -- reproduced offline with
# this image's torch: it returns the baked silu value to the last digit, tanh to 3e-16)
ACT_CST = {'silu': 1.6791767923989418, 'tanh': 1.5937334472592695, 'relu': 1.4163393446331367, 'abs': 1.001110600838467,
           'ssp': 1.878204668541552, 'sigmoid': 1.8467055342154763, 'elu': 1.2467863885570512}

DEFAULT_CONFIG = dict(  # sevenn/_const.py:95-135
    cutoff=4.5, channel=32, irreps_manual=False, lmax=1, lmax_edge=-1, lmax_node=-1,
    is_parity=True, num_convolution_layer=3,
    radial_basis={'radial_basis_name': 'bessel'},
    cutoff_function={'cutoff_function_name': 'poly_cut'},
    act_radial='silu', act_scalar={'e': 'silu', 'o': 'tanh'}, act_gate={'e': 'silu', 'o': 'tanh'},
    weight_nn_hidden_neurons=[64, 64], conv_denominator=1.0, self_connection_type='nequip',
    _normalize_sph=True, shift=0.0, scale=1.0, version='0.12.0', use_bias_in_linear=False,
    readout_as_fcn=False, readout_fcn_hidden_neurons=[30, 30], readout_fcn_activation='relu',
)


# --------------------------------------------------------------------------- #
@dataclass
class GemmBlock:
    l: int            # rows per node = 2l+1
    in_off: int
    mul_in: int       # K
    out_off: int
    mul_out: int      # N
    w_off: int        # offset into the e3nn flat weight
    alpha: float
    accumulate: bool  # C += (another in-block already wrote this out-block)
    species: int = -1  # FCTP: which species' slice (w[u, species, w])


@dataclass
class LinearSpec:
    name: str                  # state_dict key of the flat weight
    irreps_in: Irreps
    irreps_out: Irreps
    blocks: List[GemmBlock]
    zero_out: List[Tuple[int, int]]  # (offset, length) of output blocks nobody writes
    n_species: int = 0         # >0: FullyConnectedTensorProduct with a one-hot operand
    numel: int = 0
    # multi-modal linear (sevenn/nn/linear.py:66-92): n_modal extra 0e inputs carrying a one-hot of the
    # fidelity channel.  For a fixed channel they are a constant bias on the 0e outputs:
    # (out_off, mul_out, w_off, alpha) per 0e output block, weights [n_modal, mul_out] row-major
    n_modal: int = 0
    modal_bias: List[Tuple[int, int, int, float]] = field(default_factory=list)
    # o3.Linear(biases=True) (`use_bias_in_linear`, sevenn/model_build.py:468,518): one bias per channel of every 0e output
    # block, concatenated in irreps_out order -> (out_off, mul, offset into the bias tensor)
    bias_blocks: List[Tuple[int, int, int]] = field(default_factory=list)

    @property
    def bias_name(self) -> str:
        return self.name[:-len('weight')] + 'bias'

    @property
    def n_bias(self) -> int:
        return sum(m for _, m, _ in self.bias_blocks)

    @property
    def dim_in(self):
        return self.irreps_in.dim

    @property
    def dim_out(self):
        return self.irreps_out.dim


def make_linear(name: str, irreps_in: Irreps, irreps_out: Irreps, n_species: int = 0, n_modal: int = 0,
                biases: bool = False) -> LinearSpec:
    """o3.Linear (n_species=0) or FCTP(x, n_species x 0e) -> per-irrep GEMM list.
    Normalisation: 1/sqrt(total fan-in of the output block).
    n_modal > 0: the reference appends `n_modal x 0e` to the input irreps (a separate last block), so
    every 0e output block gets n_modal more fan-in and one more weight block at the END of the flat
    weight (e3nn orders blocks in-major)."""
    assert not (n_species and n_modal)
    ns = max(n_species, 1)
    in_off, out_off = irreps_in.offsets(), irreps_out.offsets()
    pairs = [(i, j) for i, (_, li, pi) in enumerate(irreps_in)
             for j, (_, lj, pj) in enumerate(irreps_out) if (li, pi) == (lj, pj)]
    fan = [0] * len(irreps_out)
    for i, j in pairs:
        fan[j] += irreps_in[i][0] * ns
    modal_j = [j for j, (_, l, p) in enumerate(irreps_out) if (l, p) == (0, 1)] if n_modal else []
    for j in modal_j:
        fan[j] += n_modal
    blocks, seen, w_off = [], set(), 0
    for i, j in pairs:
        mi, l, _ = irreps_in[i]
        mo = irreps_out[j][0]
        for s in range(ns):
            blocks.append(GemmBlock(l, in_off[i], mi, out_off[j], mo, w_off, 1.0 / math.sqrt(fan[j]),
                                    accumulate=(j in seen), species=(s if n_species else -1)))
        seen.add(j)
        w_off += mi * ns * mo
    modal_bias = []
    for j in modal_j:
        mo = irreps_out[j][0]
        modal_bias.append((out_off[j], mo, w_off, 1.0 / math.sqrt(fan[j])))
        seen.add(j)
        w_off += n_modal * mo
    zero = [(out_off[j], irreps_out[j][0] * (2 * irreps_out[j][1] + 1))
            for j in range(len(irreps_out)) if j not in seen]
    bias_blocks, b_off = [], 0
    if biases:
        for j, (mo, l, p) in enumerate(irreps_out):
            if (l, p) == (0, 1):
                bias_blocks.append((out_off[j], mo, b_off))
                b_off += mo
    return LinearSpec(name, irreps_in, irreps_out, blocks, zero, n_species, w_off, n_modal, modal_bias, bias_blocks)


def linear_modal_bias(spec: LinearSpec, flat: np.ndarray, modal_idx: int, bias_flat=None, dtype=np.float32):
    """Constant row every output of this linear carries ([dim_out], or None when there is none): what the modal one-hot
    contributes for fidelity channel `modal_idx` (multi-modal linears) plus the o3.Linear bias (`use_bias_in_linear`,
    `bias_flat` = the layer's `.bias` tensor)."""
    if not spec.n_modal and not spec.bias_blocks:
        return None
    bias = np.zeros(spec.dim_out, np.float64)
    if spec.n_modal:
        flat = np.asarray(flat, dtype=np.float64).reshape(-1)
        for off, mo, w_off, alpha in spec.modal_bias:
            bias[off:off + mo] = flat[w_off:w_off + spec.n_modal * mo].reshape(spec.n_modal, mo)[modal_idx] * alpha
    if spec.bias_blocks:
        if bias_flat is None:
            raise KeyError(f'state_dict is missing {spec.bias_name}')
        b = np.asarray(bias_flat, dtype=np.float64).reshape(-1)
        assert b.size == spec.n_bias, (spec.bias_name, b.size, spec.n_bias)
        for off, mo, b_off in spec.bias_blocks:
            bias[off:off + mo] += b[b_off:b_off + mo]
    return bias.astype(dtype)


def linear_weight_matrices(spec: LinearSpec, flat: np.ndarray):
    """Split the e3nn flat weight into per-GEMM [K, N] matrices with alpha folded."""
    flat = np.asarray(flat, dtype=np.float64).reshape(-1)
    assert flat.size == spec.numel, (spec.name, flat.size, spec.numel)
    ns = max(spec.n_species, 1)
    out = []
    for b in spec.blocks:
        w = flat[b.w_off:b.w_off + b.mul_in * ns * b.mul_out]
        if spec.n_species:
            w = w.reshape(b.mul_in, ns, b.mul_out)[:, b.species, :]
        else:
            w = w.reshape(b.mul_in, b.mul_out)
        out.append(np.ascontiguousarray(w * b.alpha, dtype=np.float32))
    return out


def linear_rows_fp64(spec: LinearSpec, flat: np.ndarray, x: np.ndarray, species=None, modal_idx: int = -1, bias_flat=None) -> np.ndarray:
    """y = Linear(x) for ir_mul rows x[n, dim_in], evaluated in fp64 on the host (load time only).
    species[n] selects the weight slice of a per-species (FCTP) linear."""
    flat = np.asarray(flat, dtype=np.float64).reshape(-1)
    x = np.asarray(x, dtype=np.float64)
    ns = max(spec.n_species, 1)
    y = np.zeros((x.shape[0], spec.dim_out), np.float64)
    for b in spec.blocks:
        w = flat[b.w_off:b.w_off + b.mul_in * ns * b.mul_out]
        w = (w.reshape(b.mul_in, ns, b.mul_out)[:, b.species, :] if spec.n_species else w.reshape(b.mul_in, b.mul_out)) * b.alpha
        rows = slice(None) if b.species < 0 else np.nonzero(np.asarray(species) == b.species)[0]
        for m in range(2 * b.l + 1):
            y[rows, b.out_off + m * b.mul_out:b.out_off + (m + 1) * b.mul_out] += \
                x[rows, b.in_off + m * b.mul_in:b.in_off + (m + 1) * b.mul_in] @ w
    b = linear_modal_bias(spec, flat, modal_idx, bias_flat, dtype=np.float64)
    if b is not None:
        y += b[None, :]
    return y


def species_only_tables(sp: 'ModelSpec', sd, modal_idx: int):
    """The first interaction block's node-level inputs depend on the species alone (one-hot embedding,
    model_build.py:383-421), so `SI1(x)` and the self-connection `sc(x)` of layer 0 are one row per SPECIES.  They are
    evaluated once at load time in fp64 and rounded once to fp32 -- the per-atom GEMMs of that layer (and their rounding,
    which is the same vector on every atom of a species and therefore does not average out over atoms: it was the engine's
    whole energy error, profiles/r04_energy_error_attribution.txt) are replaced by a table lookup.
    -> (h0[n_species, dx0] float32, sc0[n_species, gin0] float32 or None)"""
    ls = sp.layers[0]
    e64 = linear_rows_fp64(sp.embed, sd[sp.embed.name], np.eye(sp.num_species), None, modal_idx, sd.get(sp.embed.bias_name))   # fp64 embedding rows
    species = np.arange(sp.num_species)
    h0 = linear_rows_fp64(ls.si1, sd[ls.si1.name], e64, species, modal_idx, sd.get(ls.si1.bias_name))
    sc0 = linear_rows_fp64(ls.sc, sd[ls.sc.name], e64, species, modal_idx) if ls.sc is not None else None
    return h0.astype(np.float32), (None if sc0 is None else sc0.astype(np.float32))


def folded_readout(sp: 'ModelSpec', sd, modal_idx: int):
    """The reference's readout is two o3.Linear maps with nothing between them (reduce_input_to_hidden then
    reduce_hidden_to_energy, sevenn/model_build.py; `readout_as_fcn` models are refused upstream of here), i.e. ONE vector:
    e_i = x_i . v + c.  The product is taken in fp64 at load time -> (v[dim_in] float64, c float)."""
    d = sp.readout1.dim_in

    def chain(x):
        return linear_rows_fp64(sp.readout2, sd[sp.readout2.name],
                                linear_rows_fp64(sp.readout1, sd[sp.readout1.name], x, None, modal_idx, sd.get(sp.readout1.bias_name)),
                                None, -1, sd.get(sp.readout2.bias_name))

    c = chain(np.zeros((1, d)))[0, 0]
    v = chain(np.eye(d))[:, 0] - c
    return np.ascontiguousarray(v, np.float64), float(c)


# --------------------------------------------------------------------------- #
@dataclass
class ConvPath:
    i_x: int
    i_sh: int
    l1: int
    l2: int
    l3: int
    mul: int
    w_off: int        # column offset in the per-edge weight row
    x_off: int        # offset of the x block (ir_mul)
    sh_off: int
    out_off: int      # offset of the *merged* output block
    out_mul: int      # multiplicity of the merged output block
    out_ch: int       # channel offset of this path inside the merged block


@dataclass
class ConvSpec:
    irreps_x: Irreps
    irreps_sh: Irreps
    irreps_mid: Irreps        # sorted, one block per path (reference layout)
    irreps_out: Irreps        # merged (simplified) -- the engine's output layout
    paths: List[ConvPath]
    weight_numel: int

    @property
    def key(self) -> str:
        ps = ';'.join(f'{p.i_x},{p.i_sh},{p.l3},{p.w_off},{p.out_off},{p.out_mul},{p.out_ch}' for p in self.paths)
        return f'x={self.irreps_x}|sh={self.irreps_sh}|{ps}'

    @property
    def tag(self) -> str:
        return hashlib.sha1(self.key.encode()).hexdigest()[:12]


def transposed_scalar_conv(conv: ConvSpec):
    """Source-row gradient of a convolution whose outputs are all scalars (the last interaction layer of every NequIP-style
    model: paths (l, l -> 0) only), as a FORWARD convolution over the edges grouped by SOURCE atom:

        g_x[j][a, u] = sum_{e: src(e) = j} w_e[u] * T_{a b 0} Y_b(e) * g_out[dst(e)][u]      (T = sqrt(2 l3 + 1) * w3j)

    i.e. a uvu tensor product with "x" = the destination's g_out row (scalars), the same spherical harmonics and the
    same radial weights up to one constant per path.  Gathering the 4 * dout bytes of g_out per edge instead of writing
    and re-reading the 4 * dx bytes of a per-edge g_xe row pays exactly when dout < dx.
    Returns (ConvSpec of the transposed product, kappa[path] = factor on the path's weight columns), or None when the
    convolution has a non-scalar output path."""
    from .codegen import _path_terms
    if not conv.paths or any(p.l3 != 0 for p in conv.paths):
        return None
    x_off = conv.irreps_x.offsets()
    order = sorted(range(len(conv.paths)), key=lambda k: (conv.paths[k].out_off, conv.paths[k].out_ch))
    if len({conv.paths[k].i_x for k in order}) != len(order):
        return None
    # "x" of the transposed product = g_out row: one scalar block per path, in memory order
    irreps_g = Irreps([(conv.paths[k].mul, 0, 1) for k in order])
    g_index = {k: i for i, k in enumerate(order)}
    g_off = irreps_g.offsets()
    # the dense concatenation of the per-path scalar blocks must BE the g_out row: every path at its real position and no
    # scalar column that no live path writes (pruned paths, an unread 0o block) -- otherwise the transposed kernel would
    # read g_out at the wrong offsets and the caller falls back to per-edge g_xe rows + a segment sum
    if irreps_g.dim != conv.irreps_out.dim or any(g_off[g_index[k]] != p.out_off + p.out_ch for k, p in enumerate(conv.paths)):
        return None
    paths, kappa = [], []
    for k, p in enumerate(conv.paths):      # same path order -> same weight-column offsets
        mul, l, par = conv.irreps_x[p.i_x]
        q = ConvPath(g_index[k], p.i_sh, 0, p.l2, l, p.mul, p.w_off, g_off[g_index[k]], p.sh_off, x_off[p.i_x], mul, 0)
        fwd = {(a, b): v for a, b, c, v in _path_terms(p)}            # out_0 = sum T[a, b, 0] x_a Y_b
        tr = {(c, b): v for a, b, c, v in _path_terms(q)}             # out_c = sum T'[0, b, c] g Y_b
        if set(fwd) != set(tr):
            return None
        r = [fwd[key] / tr[key] for key in fwd]
        if max(r) - min(r) > 1e-12 * max(abs(v) for v in r):
            return None
        kappa.append(float(r[0]))
        paths.append(q)
    spec = ConvSpec(irreps_g, conv.irreps_sh, Irreps(conv.irreps_x), Irreps(conv.irreps_x), paths, conv.weight_numel)
    return spec, kappa


def make_conv(irreps_x: Irreps, irreps_sh: Irreps, irreps_target: Irreps, sort_by_out: bool) -> ConvSpec:
    """Instruction generation of sevenn/nn/convolution.py:61-82.  Weight columns
    follow the (possibly re-sorted) instruction order, `mul_x` per instruction;
    the output is `irreps_mid.sort()` which in memory equals its simplified form."""
    ins, mid = [], []
    for i, (mul, l1, p1) in enumerate(irreps_x):
        for j, (_, l2, p2) in enumerate(irreps_sh):
            for l3 in range(abs(l1 - l2), l1 + l2 + 1):
                if irreps_target.has(l3, p1 * p2):
                    ins.append((i, j, len(mid)))
                    mid.append((mul, l3, p1 * p2))
    mid_sorted, perm = Irreps(mid).sorted()
    ins = [(i, j, perm[k]) for i, j, k in ins]
    if sort_by_out:
        ins = sorted(ins, key=lambda t: t[2])
    merged = mid_sorted.simplified()
    m_off = merged.offsets()
    # position of each sorted mid block inside the merged blocks
    where, cur, ch = [], -1, 0
    for (mul, l, p) in mid_sorted:
        if cur < 0 or (merged[cur][1], merged[cur][2]) != (l, p) or ch + mul > merged[cur][0]:
            cur += 1
            ch = 0
        where.append((cur, ch))
        ch += mul
    x_off, sh_off = irreps_x.offsets(), irreps_sh.offsets()
    paths, w_off = [], 0
    for i, j, k in ins:
        mul, l1, _ = irreps_x[i]
        blk, ch = where[k]
        paths.append(ConvPath(i, j, l1, irreps_sh[j][1], mid_sorted[k][1], mul, w_off, x_off[i], sh_off[j],
                              m_off[blk], merged[blk][0], ch))
        w_off += mul
    return ConvSpec(irreps_x, irreps_sh, mid_sorted, merged, paths, w_off)


# --------------------------------------------------------------------------- #
@dataclass
class GateSeg:
    kind: int        # 0 scalar, 1 gated
    in_off: int
    out_off: int
    mul: int
    l: int
    gate_off: int    # offset of this segment's gate scalars in the input (-1 for scalars)
    act: int


@dataclass
class GateSpec:
    irreps_in: Irreps
    irreps_out: Irreps
    segs: List[GateSeg]


def make_gate(irreps_x: Irreps, act_scalar: Dict[str, str], act_gate: Dict[str, str]) -> GateSpec:
    """e3nn Gate input layout = sort(scalars + gates + gated).simplify()
    (SURVEY.md §9): [l=0 odd scalars | l=0 even scalars | gates | gated...]."""
    pm = {1: 'e', -1: 'o'}
    scalars = [(m, l, p) for m, l, p in irreps_x if l == 0]
    gated = [(m, l, p) for m, l, p in irreps_x if l > 0]
    gp = 1 if any(p == 1 for _, _, p in scalars) else -1
    gates = [(m, 0, gp) for m, _, _ in gated]
    cat = Irreps(scalars + gates + gated)
    srt, perm = cat.sorted()
    starts = srt.offsets()
    irreps_in = srt.simplified()
    out_irreps = Irreps(scalars + gated)
    out_off = out_irreps.offsets()
    segs = []
    ns, ng = len(scalars), len(gates)
    for b, (m, l, p) in enumerate(scalars):
        segs.append(GateSeg(0, starts[perm[b]], out_off[b], m, 0, -1, ACT_ID[act_scalar[pm[p]]]))
    for b, (m, l, p) in enumerate(gated):
        segs.append(GateSeg(1, starts[perm[ns + ng + b]], out_off[ns + b], m, l,
                            starts[perm[ns + b]], ACT_ID[act_gate[pm[gp]]]))
    return GateSpec(irreps_in, out_irreps, segs)


# --------------------------------------------------------------------------- #
@dataclass
class LayerSpec:
    t: int
    irreps_x: Irreps
    irreps_out: Irreps
    sc: Optional[LinearSpec]
    si1: LinearSpec
    conv: ConvSpec            # what the engine evaluates: the reference's paths minus those whose output nothing reads
    mlp_dims: List[int]       # radial MLP widths of `conv` (last = conv.weight_numel)
    si2: LinearSpec
    gate: GateSpec
    denominator: float
    conv_full: Optional[ConvSpec] = None     # the reference's instruction list (checkpoint layout of the last MLP layer)
    mlp_dims_full: Optional[List[int]] = None
    w_cols: Optional[np.ndarray] = None      # columns of the checkpoint's last radial layer that `conv` keeps (None: all)

    def __post_init__(self):
        if self.conv_full is None:
            self.conv_full = self.conv
        if self.mlp_dims_full is None:
            self.mlp_dims_full = list(self.mlp_dims)

    def radial_weights(self, sd) -> List[np.ndarray]:
        """[W0, W1, ...] of this layer's radial MLP as stored (no 1/sqrt(fan_in)), the last one restricted to the live paths"""
        n = len(self.mlp_dims) - 1
        ws = [np.asarray(sd[f'{self.t}_convolution.weight_nn.layer{i}.weight']) for i in range(n)]
        if self.w_cols is not None:
            ws[-1] = ws[-1][:, self.w_cols]
        return ws


def prune_unread_paths(conv: ConvSpec, si2: LinearSpec):
    """Drop the tensor-product paths whose output block no block of the following linear reads (`o3.Linear` ignores input
    irreps it has no output for: SURVEY.md section 8 table note -- the third interaction layer of SevenNet-MF-ompa computes
    0o, 1e, 2o, 3e blocks that its SI2 discards: 34 of its 68 paths, half of its radial weights).  The output layout is
    kept (the unread columns are simply never written), so everything downstream is unchanged and results are identical.
    Returns (pruned ConvSpec, kept weight columns) or (conv, None)."""
    offs = si2.irreps_in.offsets()
    read = {b.in_off for b in si2.blocks}
    ends = [off + mul * (2 * l + 1) for off, (mul, l, _) in zip(offs, si2.irreps_in)]

    def block_of(off):
        for o, e in zip(offs, ends):
            if o <= off < e:
                return o
        raise AssertionError(off)
    keep = [k for k, p in enumerate(conv.paths) if block_of(p.out_off) in read]
    if len(keep) == len(conv.paths) or not keep:
        return conv, None
    paths, cols, w = [], [], 0
    for k in keep:
        p = conv.paths[k]
        paths.append(ConvPath(p.i_x, p.i_sh, p.l1, p.l2, p.l3, p.mul, w, p.x_off, p.sh_off, p.out_off, p.out_mul, p.out_ch))
        cols.extend(range(p.w_off, p.w_off + p.mul))
        w += p.mul
    return ConvSpec(conv.irreps_x, conv.irreps_sh, conv.irreps_mid, conv.irreps_out, paths, w), np.asarray(cols, np.int64)


@dataclass
class ModelSpec:
    config: dict
    cutoff: float
    num_species: int
    lmax_edge: int
    normalize_sph: bool
    irreps_sh: Irreps
    n_basis: int
    cutoff_kind: int       # 0 poly_cut, 1 XPLOR
    cutoff_p: int
    cutoff_on: float
    act_radial: str
    embed: LinearSpec
    layers: List[LayerSpec]
    readout1: LinearSpec
    readout2: LinearSpec
    type_map: Dict[int, int] = field(default_factory=dict)
    n_modal: int = 0
    # `readout_as_fcn` (sevenn/model_build.py:124-138, nn/linear.py:145-180): the readout is an e3nn FullyConnectedNet
    # [dim_in] + hidden + [1] with activation `readout_fcn_act` instead of the two linears; None otherwise
    readout_fcn_dims: Optional[List[int]] = None
    readout_fcn_act: str = 'relu'

    def linears(self) -> List[LinearSpec]:
        out = [self.embed]
        for ls in self.layers:
            out += [l_ for l_ in (ls.sc, ls.si1, ls.si2) if l_ is not None]
        return out + ([] if self.readout_fcn_dims else [self.readout1, self.readout2])

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        s: Dict[str, Tuple[int, ...]] = {}
        s['edge_embedding.basis_function.coeffs'] = (self.n_basis,)
        s[self.embed.name] = (self.embed.numel,)
        for ls in self.layers:
            if ls.sc is not None:
                s[ls.sc.name] = (ls.sc.numel,)
            s[ls.si1.name] = (ls.si1.numel,)
            s[f'{ls.t}_convolution.denominator'] = (1,)
            for i in range(len(ls.mlp_dims_full) - 1):
                s[f'{ls.t}_convolution.weight_nn.layer{i}.weight'] = (ls.mlp_dims_full[i], ls.mlp_dims_full[i + 1])
            s[ls.si2.name] = (ls.si2.numel,)
        if self.readout_fcn_dims:
            for i in range(len(self.readout_fcn_dims) - 1):
                s[f'readout_FCN.fcn.layer{i}.weight'] = (self.readout_fcn_dims[i], self.readout_fcn_dims[i + 1])
        else:
            s[self.readout1.name] = (self.readout1.numel,)
            s[self.readout2.name] = (self.readout2.numel,)
        for lin in self.linears():
            if lin.bias_blocks:
                s[lin.bias_name] = (lin.n_bias,)
        if self.n_modal:  # ModalWiseRescale (scale.py:196-363): per species, optionally per modal
            ns = self.num_species
            s['rescale_atomic_energy.shift'] = (self.n_modal, ns) if self.config.get('use_modal_wise_shift') else (ns,)
            s['rescale_atomic_energy.scale'] = (self.n_modal, ns) if self.config.get('use_modal_wise_scale') else (ns,)
            return s
        n = max(np.asarray(self.config['shift']).size, np.asarray(self.config['scale']).size)
        s['rescale_atomic_energy.shift'] = (n,)
        s['rescale_atomic_energy.scale'] = (n,)
        return s

    def modal_index(self, modal) -> int:
        """fidelity channel index from its name (config['_modal_map']) or an int; -1 for single-modal models"""
        if not self.n_modal:
            return -1
        if modal is None:
            raise ValueError('multi-modal model: a modal must be given '
                             f"(known: {list((self.config.get('_modal_map') or {}).keys())})")
        idx = int((self.config.get('_modal_map') or {})[modal]) if isinstance(modal, str) else int(modal)
        if not 0 <= idx < self.n_modal:
            raise ValueError(f'modal index {idx} out of range (model has {self.n_modal})')
        return idx

    def rescale_vectors(self, sd, modal_idx: int):
        """(scale, shift) as flat float arrays for the chosen fidelity channel: [1] global or [n_species].
        Shapes come from the checkpoint TENSORS (a species-wise checkpoint may still carry scalars in its
        config): [1] (Rescale), [n_species] (SpeciesWiseRescale), or -- multi-modal models -- [n_modal, n_species]."""
        ns = self.num_species
        out = []
        for name in ('scale', 'shift'):
            v = np.asarray(sd[f'rescale_atomic_energy.{name}'], np.float64)
            if self.n_modal and v.size == self.n_modal * ns and v.size != ns:
                v = v.reshape(self.n_modal, ns)[modal_idx]
            v = v.reshape(-1)
            if v.size not in (1, ns):
                raise ValueError(f'rescale_atomic_energy.{name} has {v.size} entries: expected 1, n_species = {ns}'
                                 + (f' or n_modal x n_species = {self.n_modal} x {ns}' if self.n_modal else ''))
            out.append(v)
        n = max(out[0].size, out[1].size)
        return np.broadcast_to(out[0], (n,)).copy(), np.broadcast_to(out[1], (n,)).copy()

    def num_weights(self) -> int:
        return sum(int(np.prod(v)) for k, v in self.param_shapes().items()
                   if not (k.endswith('denominator') or k.startswith('rescale')))


def _vt(v: str):
    return tuple(int(t) for t in str(v).split('.')[:3] if t.isdigit())


def old_convolution_order(version) -> bool:
    """True for checkpoints whose convolution instructions (and radial-weight columns) are in the
    pre-0.11 generation order: version < 0.11.0, or exactly `0.11.0.dev0`
    (patch_state_dict_if_old, sevenn/scripts/backward_compatibility.py:165-184)."""
    toks = str(version).split('.')
    vs = _vt(version)
    suffix = toks[3] if len(toks) == 4 else ''
    return vs < (0, 11, 0) or (vs == (0, 11, 0) and suffix == 'dev0')


def build_model_spec(config: dict) -> ModelSpec:
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config)
    ub = bool(cfg.get('use_bias_in_linear'))
    n_modal = int(cfg.get('_number_of_modalities', 0)) if cfg.get('use_modality') else 0
    if cfg.get('use_modality') and n_modal < 2:
        raise ValueError('use_modality needs _number_of_modalities >= 2')
    m_embed = n_modal if cfg.get('use_modal_node_embedding') else 0   # model_build.py:213-230
    m_si1 = n_modal if cfg.get('use_modal_self_inter_intro') else 0
    m_si2 = n_modal if cfg.get('use_modal_self_inter_outro') else 0
    m_out = n_modal if cfg.get('use_modal_output_block') else 0
    ns = int(cfg.get('_number_of_species') or cfg.get('num_species') or len(cfg['chemical_species']))
    ch = int(cfg['channel'])
    L = int(cfg['num_convolution_layer'])
    lmax_edge = cfg['lmax_edge'] if cfg['lmax_edge'] > 0 else cfg['lmax']
    lmax_node = cfg['lmax_node'] if cfg['lmax_node'] > 0 else cfg['lmax']
    irreps_sh = Irreps.spherical_harmonics(lmax_edge, -1 if cfg['is_parity'] else 1)
    legacy = bool(cfg.get('_legacy_v08', False))
    sort_by_out = not old_convolution_order(cfg['version'])
    manual = cfg['irreps_manual']
    if manual is not False:
        manual = [Irreps(s) for s in manual]
        if len(manual) != L + 1:
            raise RuntimeError('invalid irreps_manual input given')
    sc_types = cfg['self_connection_type']
    if isinstance(sc_types, str):
        sc_types = [sc_types] * L
    denom = cfg['conv_denominator']
    if not isinstance(denom, (list, tuple)):
        denom = [denom] * L
    cf = cfg['cutoff_function']
    if cf.get('cutoff_function_name') not in ('poly_cut', 'XPLOR'):
        raise NotImplementedError(f"cutoff function {cf.get('cutoff_function_name')!r}: the HIP engine implements "
                                  "'poly_cut' and 'XPLOR' (sevenn/nn/edge_embedding.py:106-160)")
    if cfg['radial_basis'].get('radial_basis_name', 'bessel') != 'bessel':
        raise NotImplementedError(f"radial basis {cfg['radial_basis'].get('radial_basis_name')!r}: the HIP engine "
                                  "implements 'bessel' (sevenn/nn/edge_embedding.py:81-103)")
    for key in ('act_radial',) + (('readout_fcn_activation',) if cfg.get('readout_as_fcn') else ()):
        if cfg[key] not in ACT_ID:
            raise ValueError(f"{key}={cfg[key]!r}: supported activations are {sorted(ACT_ID)}")
    for key in ('act_scalar', 'act_gate'):
        bad = [v for v in cfg[key].values() if v not in ACT_ID]
        if bad:
            raise ValueError(f"{key} uses {bad[0]!r}: supported activations are {sorted(ACT_ID)}")
    ckind = {'poly_cut': 0, 'XPLOR': 1}[cf['cutoff_function_name']]
    n_basis = int(cfg['radial_basis'].get('bessel_basis_num', 8))
    hidden = list(cfg['weight_nn_hidden_neurons'])

    irreps_x = Irreps(f'{ch}x0e') if manual is False else manual[0]
    embed = make_linear('onehot_to_feature_x.linear.weight', Irreps(f'{ns}x0e'), irreps_x, n_modal=m_embed, biases=ub)
    layers = []
    for t in range(L):
        parity_mode = 'full'
        if t == L - 1 and not legacy:
            lmax_node, parity_mode = 0, 'even'
        irreps_out = (infer_irreps_out(irreps_x, irreps_sh, lmax_node, parity_mode, ch)
                      if manual is False else manual[t + 1])
        irreps_out_tp = infer_irreps_out(irreps_x, irreps_sh, irreps_out.lmax, parity_mode, False)
        gate = make_gate(irreps_out, cfg['act_scalar'], cfg['act_gate'])
        conv = make_conv(irreps_x, irreps_sh, irreps_out_tp, sort_by_out)
        assert conv.irreps_out == irreps_out_tp, (conv.irreps_out, irreps_out_tp)
        if sc_types[t] == 'nequip':
            sc = make_linear(f'{t}_self_connection_intro.fc_tensor_product.weight', irreps_x, gate.irreps_in, ns)
        elif sc_types[t] == 'linear':
            sc = make_linear(f'{t}_self_connection_intro.linear.weight', irreps_x, gate.irreps_in)
        elif sc_types[t] == 'none':
            sc = None
        else:
            raise ValueError(f'Unknown self_connection_type found: {sc_types[t]}')
        si2 = make_linear(f'{t}_self_interaction_2.linear.weight', irreps_out_tp, gate.irreps_in, n_modal=m_si2, biases=ub)
        live, w_cols = prune_unread_paths(conv, si2) if cfg.get('_prune_unread_paths', True) else (conv, None)
        layers.append(LayerSpec(
            t, irreps_x, irreps_out, sc,
            make_linear(f'{t}_self_interaction_1.linear.weight', irreps_x, irreps_x, n_modal=m_si1, biases=ub),
            live, [n_basis] + hidden + [live.weight_numel], si2, gate, float(denom[t]),
            conv_full=conv, mlp_dims_full=[n_basis] + hidden + [conv.weight_numel], w_cols=w_cols))
        irreps_x = irreps_out
    hid = Irreps([((ch if legacy else irreps_x.dim) // 2, 0, 1)])
    tm = cfg.get('_type_map') or {}
    return ModelSpec(
        cfg, float(cfg['cutoff']), ns, lmax_edge, bool(cfg['_normalize_sph']), irreps_sh, n_basis,
        ckind, int(cf.get('poly_cut_p_value', 6)), float(cf.get('cutoff_on', 0.0)), cfg['act_radial'],
        embed, layers,
        make_linear('reduce_input_to_hidden.linear.weight', irreps_x, hid, n_modal=m_out, biases=ub),
        make_linear('reduce_hidden_to_energy.linear.weight', hid, Irreps('1x0e'), biases=ub),
        {int(k): int(v) for k, v in tm.items()}, n_modal,
        ([irreps_x.dim] + [int(v) for v in cfg.get('readout_fcn_hidden_neurons', [30, 30])] + [1]) if cfg.get('readout_as_fcn') else None,
        str(cfg.get('readout_fcn_activation', 'relu')))


# --------------------------------------------------------------------------- #
# named model shapes (hyper-parameters from the reference presets)
# --------------------------------------------------------------------------- #
def sevennet_0_config(num_species: int = 1, conv_denominator: float = 28.0) -> dict:
    """sevenn/presets/sevennet-0.yaml:4-31 (5 layers, SO(3)-only, XPLOR 4.5/5.0,
    `linear` self-connection).  The released checkpoints predate v0.10, so the
    spherical harmonics are evaluated on the raw edge vector
    (backward_compatibility.py:38-39)."""
    return dict(
        cutoff=5.0, channel=128, is_parity=False, lmax=2, num_convolution_layer=5,
        irreps_manual=['128x0e', '128x0e+64x1e+32x2e', '128x0e+64x1e+32x2e', '128x0e+64x1e+32x2e',
                       '128x0e+64x1e+32x2e', '128x0e'],
        weight_nn_hidden_neurons=[64, 64],
        radial_basis={'radial_basis_name': 'bessel', 'bessel_basis_num': 8},
        cutoff_function={'cutoff_function_name': 'XPLOR', 'cutoff_on': 4.5},
        act_gate={'e': 'silu', 'o': 'tanh'}, act_scalar={'e': 'silu', 'o': 'tanh'},
        conv_denominator=conv_denominator, self_connection_type='linear',
        _normalize_sph=False, _number_of_species=num_species, shift=0.0, scale=1.0, version='0.9.5')


def sevennet_mf_ompa_config(conv_denominator: float = 46.0) -> dict:
    """sevenn/presets/mf_ompa_fine_tune.yaml:4-45,83-99: O(3) parity, lmax 3, XPLOR 5.5/6.0, `nequip`
    self-connection over the universal 119-species table, two fidelity channels (mpa, omat24) feeding
    self-interaction 1/2 and the output block, modal-wise shift."""
    return dict(
        cutoff=6.0, channel=128, is_parity=True, lmax=3, num_convolution_layer=5,
        irreps_manual=['128x0e', '128x0e+64x1o+32x2e+32x3o', '128x0e+64x1o+64x1e+32x2o+32x2e+32x3o+32x3e',
                       '128x0o+128x0e+64x1o+64x1e+32x2o+32x2e+32x3o+32x3e', '128x0e+64x1o+32x2e+32x3o', '128x0e'],
        weight_nn_hidden_neurons=[64, 64],
        radial_basis={'radial_basis_name': 'bessel', 'bessel_basis_num': 8},
        cutoff_function={'cutoff_function_name': 'XPLOR', 'cutoff_on': 5.5},
        conv_denominator=conv_denominator, self_connection_type='nequip',
        _normalize_sph=True, _number_of_species=119, shift=0.0, scale=1.0, version='0.11.0',
        use_modality=True, _number_of_modalities=2, _modal_map={'mpa': 0, 'omat24': 1},
        use_modal_node_embedding=False, use_modal_self_inter_intro=True, use_modal_self_inter_outro=True,
        use_modal_output_block=True, use_modal_wise_shift=True, use_modal_wise_scale=False)


def sevennet_l3i5_config(num_species: int = 1, conv_denominator: float = 28.0) -> dict:
    """sevenn/presets/sevennet-l3i5.yaml:4-40."""
    return dict(
        cutoff=5.0, channel=128, is_parity=False, lmax=3, num_convolution_layer=5,
        irreps_manual=['128x0e'] + ['128x0e+64x1e+32x2e+32x3e'] * 4 + ['128x0e'],
        weight_nn_hidden_neurons=[64, 64],
        radial_basis={'radial_basis_name': 'bessel', 'bessel_basis_num': 8},
        cutoff_function={'cutoff_function_name': 'poly_cut', 'poly_cut_p_value': 6},
        conv_denominator=conv_denominator, self_connection_type='linear',
        _normalize_sph=True, _number_of_species=num_species, shift=0.0, scale=1.0, version='0.10.0')
