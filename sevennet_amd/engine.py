"""SevenNet energy/force evaluation on MI355X: host-side sequencer over libsnet_hip.so.

One `HipForceEngine.compute()` = one MD-step evaluation:
forward through edge embedding, L interaction blocks and the readout, then a
hand-scheduled analytic reverse pass that yields dE/d(edge_vec), forces and
virial -- no torch autograd anywhere in the step.  PyTorch is used only for
device memory, streams and (multi-GPU) torch.distributed.

Mirrors, op for op, what the reference executes in
  AtomGraphSequential.forward            sevenn/nn/sequential.py:179-183
  NequIP_interaction_block (op order)    sevenn/nn/interaction_blocks.py:41-76
  ForceStressOutputFromEdge.forward      sevenn/nn/force_output.py:171-230
and, with a `halo` object, the per-layer ghost exchange of
  PairE3GNNParallel::compute             sevenn/pair_e3gnn/pair_e3gnn_parallel.cpp:358-441
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .model_spec import (ACT_CST, ACT_ID, LinearSpec, ModelSpec, build_model_spec, folded_readout, linear_modal_bias,
                         linear_weight_matrices, species_only_tables, transposed_scalar_conv)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------- #
@dataclass
class Graph:
    """Device-resident edge list in the layout the kernels consume.

    Edges are sorted by center atom (CSR `row_ptr` over all n_total rows; ghost
    rows have empty segments).  `eperm`/`col_ptr` group edge ids by neighbor
    (source) atom for the deterministic reverse gathers."""
    n_total: int
    n_local: int
    n_edges: int
    types: torch.Tensor      # int32 [n_total]
    center: torch.Tensor     # int32 [E]  (edge_index[0], sorted)
    src: torch.Tensor        # int32 [E]  (edge_index[1])
    row_ptr: torch.Tensor    # int32 [n_total+1]
    col_ptr: torch.Tensor    # int32 [n_total+1]
    eperm: torch.Tensor      # int32 [E]
    edge_vec: torch.Tensor   # float32 [E,3]
    order: Optional[torch.Tensor] = None  # permutation applied to the caller's edge order (None = already sorted)
    species_rows: Optional[List[torch.Tensor]] = None  # per species: int32 local row ids (FCTP self-connection)
    # undirected pairs (snet_edge_pairs): w_row[e] = radial-weight row of edge e, pair_edge[p] = one edge of pair p
    w_row: Optional[torch.Tensor] = None
    pair_edge: Optional[torch.Tensor] = None
    n_pairs: int = 0
    # 16-edge tiles of the CSR segments (snet_edge_tiles): work list of the fused reverse kernels
    tile_ptr: Optional[torch.Tensor] = None
    tile_node: Optional[torch.Tensor] = None
    n_tiles: int = 0
    # edges grouped by SOURCE atom (order eperm): destination row and radial row of each (transposed scalar convolution)
    src_T: Optional[torch.Tensor] = None
    w_row_T: Optional[torch.Tensor] = None
    # bricks of a decomposition number their local atoms interior first (parallel.BrickGraph.n_interior): rows [0, n_interior)
    # have no ghost source.  0 = unknown / no split.
    n_interior: int = 0
    _tile_split: Optional[tuple] = None
    _tiles_packed: Optional[tuple] = None

    def by_source(self, lib, stream):
        """(center[eperm], w_row[eperm]) as int32 arrays, built on first use (w_row None: the edge's own row, eperm)"""
        if self.src_T is None:
            self.src_T, self.w_row_T = torch.empty_like(self.eperm), torch.empty_like(self.eperm)
            _lib.check(lib.snet_edges_by_source(_ptr(self.row_ptr), self.n_local, _ptr(self.eperm), _ptr(self.w_row),
                                                self.n_edges, _ptr(self.src_T), _ptr(self.w_row_T), stream),
                       'snet_edges_by_source')
        return self.src_T, self.w_row_T

    def _packed_tiles(self):
        """packed tiles (snet_edge_tiles_packed): ONE list, built per row range when the graph has an interior / boundary cut
        (no tile straddles it, and the tile of the cut's sentinel is the first boundary tile) -> (tile_e0, tile_nodes, n, k)"""
        lib = _lib.load()
        dev = self.edge_vec.device
        cut = self.n_interior if 0 < self.n_interior < self.n_local else 0
        with torch.cuda.device(dev):
            cap = self.n_local + self.n_edges // 16 + 2
            tp = torch.empty(cap + 1, dtype=torch.int32, device=dev)
            tn = torch.empty(2 * cap, dtype=torch.int32, device=dev)
            k = 0
            n = C.c_int64()
            if cut:
                _lib.check(lib.snet_edge_tiles_packed(_ptr(self.row_ptr), 0, cut, _ptr(tp), _ptr(tn), cap, C.byref(n), _stream()),
                           'snet_edge_tiles_packed')
                k = int(n.value)
            _lib.check(lib.snet_edge_tiles_packed(_ptr(self.row_ptr), cut, self.n_local, C.c_void_p(tp.data_ptr() + 4 * k),
                                                  C.c_void_p(tn.data_ptr() + 8 * k), cap - k, C.byref(n), _stream()),
                       'snet_edge_tiles_packed')
        return tp, tn, k + int(n.value), k

    def tiles(self, mode: int = 0):
        """(tile_ptr, tile_node, n_tiles) of one reverse kernel's work-list format (snet_fused_plan_tile_mode), built on first use"""
        if mode == 1:
            if self._tiles_packed is None:
                self._tiles_packed = self._packed_tiles()
            return self._tiles_packed[:3]
        if self.tile_ptr is None:
            lib = _lib.load()
            dev = self.edge_vec.device
            with torch.cuda.device(dev):
                tp = torch.empty(self.n_local + 1, dtype=torch.int32, device=dev)
                cap = self.n_local + self.n_edges // 16 + 1
                tn = torch.empty(cap, dtype=torch.int32, device=dev)
                n = C.c_int64()
                _lib.check(lib.snet_edge_tiles(_ptr(self.row_ptr), self.n_local, _ptr(tp), _ptr(tn), cap, C.byref(n),
                                               _stream()), 'snet_edge_tiles')
            self.tile_ptr, self.tile_node, self.n_tiles = tp, tn, int(n.value)
        return self.tile_ptr, self.tile_node, self.n_tiles

    def tiles_split(self, mode: int = 0):
        """the tile list cut at n_interior: ((tile_ptr, tile_node, n) of the interior rows, (tile_ptr', tile_node', n') of the
        boundary rows) -- views / one shifted copy of the full list, built on first use"""
        if mode == 1:
            tp, tn, nt = self.tiles(1)
            k = self._tiles_packed[3]
            return (tp, tn, k), (tp[k:], tn[2 * k:], nt - k)
        if self._tile_split is None:
            tp, tn, nt = self.tiles()
            k = int(tp[self.n_interior].item())       # first tile of the first boundary row (one readback per graph)
            self._tile_split = ((tp, tn, k), ((tp - k).contiguous(), tn[k:], nt - k))
        return self._tile_split

    def share_pairs(self):
        """Number the undirected pairs so the radial MLP runs once per pair (in place; returns self)."""
        lib = _lib.load()
        E = self.n_edges
        if E == 0:
            return self
        dev = self.edge_vec.device
        with torch.cuda.device(dev):
            w_row = torch.empty(E, dtype=torch.int32, device=dev)
            pair_edge = torch.empty(E, dtype=torch.int32, device=dev)
            n = C.c_int64()
            _lib.check(lib.snet_edge_pairs(_ptr(self.row_ptr), _ptr(self.src), _ptr(self.edge_vec), self.n_local, E,
                                           _ptr(w_row), _ptr(pair_edge), C.byref(n), _stream()), 'snet_edge_pairs')
        self.w_row, self.pair_edge, self.n_pairs = w_row, pair_edge[:n.value].contiguous(), int(n.value)
        return self


def species_row_lists(types_local: torch.Tensor, num_species: int) -> List[torch.Tensor]:
    """per species: int32 ids of the local rows of that species, ascending -- views of ONE sorted tensor
    (one stable sort + one bincount, one device sync; 119-species models would otherwise pay one
    `nonzero` sync per species)"""
    t = types_local.long()
    order = torch.sort(t, stable=True).indices.to(torch.int32)
    counts = torch.bincount(t, minlength=num_species).cpu().tolist()
    rows, o = [], 0
    for s in range(num_species):
        rows.append(order[o:o + counts[s]])
        o += counts[s]
    return rows


def build_graph(types, edge_index, edge_vec, n_local: Optional[int] = None, device='cuda',
                num_species: int = 0, share_pairs: bool = True, n_interior: int = 0) -> Graph:
    """types[n_total] (species index), edge_index[2,E] (row 0 = center / destination,
    row 1 = neighbor / source; pair_e3gnn.cpp:192-197 convention), edge_vec[E,3]."""
    dev = torch.device(device)
    types = torch.as_tensor(types).to(dev, torch.int32)
    ei = torch.as_tensor(edge_index).to(dev, torch.int64)
    ev = torch.as_tensor(edge_vec).to(dev, torch.float32)
    n_total = int(types.shape[0])
    n_local = n_total if n_local is None else int(n_local)
    E = int(ei.shape[1])
    order = None
    if E > 1 and bool((ei[0, 1:] < ei[0, :-1]).any()):
        order = torch.sort(ei[0], stable=True).indices
        ei = ei[:, order]
        ev = ev[order]
    center, src = ei[0], ei[1]
    if E and (int(center.max()) >= n_local):
        raise ValueError('edge centers must be owned (local) atoms')
    row_ptr = torch.zeros(n_total + 1, dtype=torch.int64, device=dev)
    col_ptr = torch.zeros(n_total + 1, dtype=torch.int64, device=dev)
    if E:
        row_ptr[1:] = torch.cumsum(torch.bincount(center, minlength=n_total), 0)
        col_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=n_total), 0)
        eperm = torch.sort(src, stable=True).indices
    else:
        eperm = torch.zeros(0, dtype=torch.int64, device=dev)
    rows = None
    if num_species:
        rows = species_row_lists(types[:n_local], num_species)
    g = Graph(n_total, n_local, E, types, center.to(torch.int32).contiguous(), src.to(torch.int32).contiguous(),
              row_ptr.to(torch.int32), col_ptr.to(torch.int32), eperm.to(torch.int32).contiguous(),
              ev.contiguous(), order, rows)
    g.n_interior = int(n_interior) if 0 < int(n_interior) < n_local else 0
    return g.share_pairs() if share_pairs and dev.type == 'cuda' else g


# --------------------------------------------------------------------------- #
def _pack_split(m: np.ndarray, dev) -> torch.Tensor:
    """[K,N] fp32 weights -> device buffer of split-precision B fragments (snet_gemm_split_pack)."""
    lib = _lib.load()
    m = np.ascontiguousarray(m, dtype=np.float32)
    K, N = m.shape
    buf = np.empty(int(lib.snet_gemm_split_size(K, N)), np.uint8)
    _lib.check(lib.snet_gemm_split_pack(C.c_void_p(m.ctypes.data), K, N, C.c_void_p(buf.ctypes.data)),
               'snet_gemm_split_pack')
    return torch.from_numpy(buf).to(dev)


class _Linear:
    """Device weights of one LinearSpec (per-GEMM [K,N] and [N,K] copies) and its launch plan:
    per-irrep GEMMs that write distinct output blocks are grouped into one launch."""

    def __init__(self, spec: LinearSpec, flat: np.ndarray, dev, split: bool = True, modal_idx: int = -1, bias_flat=None):
        self.spec = spec
        self.split = split
        b = linear_modal_bias(spec, flat, modal_idx, bias_flat)   # constant row: multi-modal one-hot for this channel + o3.Linear bias
        self.bias = None if b is None else torch.from_numpy(b).to(dev)
        mats = linear_weight_matrices(spec, flat)
        if split:  # bf16 x 6 split-precision MFMA: weights live on the device as packed B fragments
            self.w = [_pack_split(m, dev) for m in mats]
            self.wt = [_pack_split(np.ascontiguousarray(m.T), dev) for m in mats]
        else:
            self.w = [torch.from_numpy(m).to(dev) for m in mats]
            self.wt = [torch.from_numpy(np.ascontiguousarray(m.T)).to(dev) for m in mats]
        fed = {b.in_off for b in spec.blocks}   # input column ranges no block reads (model_file._write_linear's zero_in)
        self.zero_in = [(off, m * (2 * l + 1)) for off, (m, l, _) in zip(spec.irreps_in.offsets(), spec.irreps_in) if off not in fed]
        self.groups_fwd = self._plan(transpose=False)
        self.groups_T = self._plan(transpose=True)
        # largest row norm of the TRANSPOSED map: |(Linear^T g)[k]| <= t_norm * ||g||_2 for every input entry k
        # (Cauchy-Schwarz over all blocks that feed the same input block; per species for the FCTP self-connection)
        acc = {}
        for b, m in zip(spec.blocks, mats):
            acc[(b.in_off, b.species)] = acc.get((b.in_off, b.species), 0.0) + (np.asarray(m, np.float32).astype(np.float64) ** 2).sum(1)   # fp32 weights, as the native host reads them
        self.t_norm = float(max(np.sqrt(v).max() for v in acc.values())) if acc else 0.0

    def _plan(self, transpose: bool):
        """-> list of (species, GemmDesc array, n, first_use_accumulates[list of target offsets])"""
        sp = self.spec
        groups = []          # [species, [descs], set(targets)]
        written = set()      # targets already written by an earlier launch (-> accumulate)
        pair_seen = {}       # transposed FCTP: species slices of one (in,out) pair hit disjoint rows
        for b, w, wt in zip(sp.blocks, self.w, self.wt):
            if transpose:
                tgt, a_off, c_off, K, N, B = b.in_off, b.out_off, b.in_off, b.mul_out, b.mul_in, wt
            else:
                tgt, a_off, c_off, K, N, B = b.out_off, b.in_off, b.out_off, b.mul_in, b.mul_out, w
            key = (tgt, b.species)
            acc = key in written
            desc = _lib.GemmDesc(None if self.split else B.data_ptr(), B.data_ptr() if self.split else None, a_off, c_off,
                                 2 * b.l + 1, K, N, int(acc))
            placed = False
            if groups and groups[-1][0] == b.species and len(groups[-1][1]) < 8 and tgt not in groups[-1][2]:
                groups[-1][1].append(desc)
                groups[-1][2].add(tgt)
                placed = True
            if not placed:
                groups.append([b.species, [desc], {tgt}])
            written.add(key)
        out = []
        for species, descs, _ in groups:
            arr = (_lib.GemmDesc * len(descs))(*descs)
            out.append((species, arr, len(descs)))
        return out


class _Span:
    """HIP-event bracket around one kernel class (bench.py's per-kernel timing)."""

    def __init__(self, eng, name):
        self.eng, self.name = eng, name

    def __enter__(self):
        self.on = self.eng.events is not None and (self.eng.event_filter is None or self.name in self.eng.event_filter)
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            self.eng.events.append((self.name, self.a, self.b))


class HipForceEngine:
    OVERLAP_MAX_EDGES = 1_000_000

    def __init__(self, config: dict, state_dict: Dict[str, np.ndarray], device='cuda:0', mlp_mode: str = 'bf16x6',
                 linear_mode: str = 'bf16x6', fused='auto', fused_terms='f16x3', modal=None, overlap: bool = True,
                 mlp_tail: bool = True, transposed_conv: bool = True, species_tables: bool = True,
                 fold_readout: bool = True):
        """mlp_mode / linear_mode: 'bf16x6' (split-precision MFMA, fp32-class accuracy, default) or
        'fp32' (exact fp32 MFMA) for the fused radial MLP / the node-level equivariant linears.
        fused: 'auto' (default) / True / False / 'fwd' / 'bwd' -- run the radial MLP's last layer INSIDE the
        tensor-product kernels (snet_conv_fwd_fused / snet_conv_bwd_fused): neither w[E,wn] nor g_w[E,wn] is
        materialised; what is kept per layer is h2[pairs,64].  Needs mlp_mode 'bf16x6' and a shape whose channel
        multiplicities are multiples of 16 ('auto': used where available; True: required).
        fused_terms: precision mode of the in-kernel products w = h2 @ W2 and g_h2 = g_w @ W2^T (name or C-ABI code):
        'f16x3' / 4 (default) = two fp16 terms per operand, three products hi*hi + hi*lo + lo*hi on
        v_mfma_f32_16x16x32_f16 with power-of-two operand scaling: fp32-rounding class (22 significand bits per
        operand) at the matrix-core cost of bf16x3; 'bf16x6' / 3 = three bf16 terms, six products (fp32-rounding
        class, twice the matrix-core work); 'bf16x3' / 2 = two bf16 terms (~2^-17 per product: at MD-scale forces,
        max|F| = 8 eV/A, 7e-5 .. 3.5e-4 eV/A vs the fp64 oracle where the fp32-class modes give 1e-5 .. 4e-5:
        profiles/r03_terms_accuracy.txt -- NOT inside the 1e-4 eV/A bar); 'bf16' / 1 = plain bf16 (1e-2 relative).
        overlap: run the radial MLPs on a second HIP stream -- forward: all layers' weights are produced
        from the edge embedding while the node-level work of earlier layers runs; reverse: the MLP reverse of
        layer t (which only feeds the final radial gradient) runs beside the rest of the reverse pass.
        Applied to graphs of at most OVERLAP_MAX_EDGES edges (the per-GPU share of a 100k-atom cell on
        4-8 GPUs, and mid-size single-GPU systems).  Above that every kernel fills the GPU by itself: the
        measured gain shrinks to 0.5-2 % (2.7e6 edges: 58.9 -> 56.9 ms) while per-kernel timings stop being
        exclusive, so large graphs stay on one stream.  Every buffer the side stream touches is allocated on the main stream and
        reused explicitly (double-buffered g_w): `record_stream`-deferred frees of 10-GB blocks made the
        caching allocator fall back to hipMalloc/hipFree, a 3x slowdown at 100k atoms.
        transposed_conv: scalar-output layers (the last one) take their source-row gradient from a forward convolution of the
        transposed tensor product over the edges grouped by source atom instead of per-edge g_xe rows + a segment sum.
        species_tables: layer 0's SI1(x) and self-connection rows, which depend on the species alone, come from per-species
        tables evaluated in fp64 at load time (model_spec.species_only_tables) instead of per-atom GEMMs (False: the
        GEMMs, kept for A/B measurements).
        fold_readout: the two readout linears as one fp64-folded vector, per-atom dot product and rescale in fp64
        (snet_readout_energy) instead of two GEMMs + snet_rescale_reduce (False: the GEMMs, for A/B measurements).
        modal: fidelity channel (name from config['_modal_map'] or index) of a multi-modal model; the
        one-hot inputs of its linears become constant biases, shift/scale rows are selected at load.
        """
        if mlp_mode not in ('bf16x6', 'fp32') or linear_mode not in ('bf16x6', 'fp32'):
            raise ValueError("mlp_mode / linear_mode must be 'bf16x6' or 'fp32'")
        if fused not in ('auto', True, False, 'fwd', 'bwd'):
            raise ValueError("fused must be 'auto', True, False, 'fwd' or 'bwd'")
        self.mlp_mode = mlp_mode
        self.linear_mode = linear_mode
        codes = {'bf16': 1, 'bf16x3': 2, 'bf16x6': 3, 'f16x3': 4}
        if fused_terms not in codes and fused_terms not in codes.values():
            raise ValueError(f"fused_terms must be one of {sorted(codes)} (or the C-ABI code 1..4)")
        self.fused_terms = int(codes.get(fused_terms, fused_terms))
        self.fused_mode = {v: k for k, v in codes.items()}[self.fused_terms]
        self.overlap = bool(overlap)
        self.halo_split = True   # False: exchange, then the whole convolution (A/B measurements of the overlap)
        self._side = None  # second stream, created on first use
        self._acc_descs = {}  # id(descriptor array) -> its all-accumulating copy (the arrays live as long as the engine)
        self.events = None  # set to [] to collect (name, start, end) HIP events per kernel class
        self.event_filter = None  # optional set of class names: only those spans are recorded
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('HipForceEngine needs a ROCm GPU (no CPU fallback exists)')
        self.dev = torch.device(device)
        self.spec: ModelSpec = build_model_spec(config)
        sp = self.spec
        self.modal_idx = mi = sp.modal_index(modal)
        shapes = sp.param_shapes()
        sd = {}
        for k, shp in shapes.items():
            if k not in state_dict:
                raise KeyError(f'state_dict is missing {k}')
            v = state_dict[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if k.startswith('rescale_atomic_energy.'):  # shape taken from the tensor itself (ModelSpec.rescale_vectors)
                sd[k] = v.astype(np.float64)
                continue
            if v.size != int(np.prod(shp)):
                raise ValueError(f'{k}: checkpoint tensor has {v.size} entries, the model config implies {shp}')
            sd[k] = v.astype(np.float64).reshape(shp)

        self.edge_params = _lib.EdgeParams(sp.cutoff, sp.n_basis, sp.cutoff_kind, sp.cutoff_p, sp.cutoff_on,
                                           sp.lmax_edge, int(sp.normalize_sph))
        self.coeffs = (C.c_float * sp.n_basis)(*[float(v) for v in sd['edge_embedding.basis_function.coeffs']])
        self.nsh = sp.irreps_sh.dim
        with torch.cuda.device(self.dev):
            emb = linear_weight_matrices(sp.embed, sd[sp.embed.name])
            assert len(emb) == 1
            eb = linear_modal_bias(sp.embed, sd[sp.embed.name], mi, sd.get(sp.embed.bias_name))
            table = emb[0] if eb is None else emb[0] + eb[None, :]
            self.embed_table = torch.from_numpy(np.ascontiguousarray(table, np.float32)).to(self.dev)  # [n_species, dim0]
            self.layers = []
            split = linear_mode == 'bf16x6'
            for ls in sp.layers:
                L = type('L', (), {})()
                L.spec = ls
                L.sc = _Linear(ls.sc, sd[ls.sc.name], self.dev, split, mi) if ls.sc is not None else None
                L.si1 = _Linear(ls.si1, sd[ls.si1.name], self.dev, split, mi, sd.get(ls.si1.bias_name))
                L.si2 = _Linear(ls.si2, sd[ls.si2.name], self.dev, split, mi, sd.get(ls.si2.bias_name))
                L.mlp_w, L.mlp_wt = [], []
                rw = ls.radial_weights(sd)   # last layer restricted to the live paths' columns
                for i in range(len(ls.mlp_dims) - 1):
                    w = rw[i] / np.sqrt(ls.mlp_dims[i])
                    L.mlp_w.append(torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(self.dev))
                    L.mlp_wt.append(torch.from_numpy(np.ascontiguousarray(w.T, dtype=np.float32)).to(self.dev))
                L.fused_mlp = (len(ls.mlp_dims) == 4 and ls.mlp_dims[1] == 64 and ls.mlp_dims[2] == 64
                               and ls.mlp_dims[0] <= 32)
                L.mlp_plan = None
                if L.fused_mlp:
                    d = ls.mlp_dims
                    hw = [np.ascontiguousarray(rw[i] / np.sqrt(d[i]), dtype=np.float32) for i in range(3)]
                    fp = [w.ctypes.data_as(C.POINTER(C.c_float)) for w in hw]
                    mp = C.c_void_p()
                    _lib.check(self.lib.snet_radial_mlp_plan_create(d[0], d[1], d[2], d[3], fp[0], fp[1], fp[2],
                                                                    ACT_ID[sp.act_radial], ACT_CST[sp.act_radial],
                                                                    1 if mlp_mode == 'bf16x6' else 0, C.byref(mp)),
                               'snet_radial_mlp_plan_create')
                    L.mlp_plan = mp
                L.scale = 1.0 / float(sd[f'{ls.t}_convolution.denominator'][0])
                if ls.conv.tag not in _lib.compiled_conv_tags():   # a model outside sevennet_amd/shapes.py: compile its shape now
                    from .jit import ensure_conv_shape
                    ensure_conv_shape(ls.conv)
                plan = C.c_void_p()
                _lib.check(self.lib.snet_conv_plan_create(ls.conv.tag.encode(), C.byref(plan)), 'snet_conv_plan_create')
                L.plan = plan
                L.fplan = None
                L.mlp_tail = False
                can = bool(L.fused_mlp and mlp_mode == 'bf16x6' and self.lib.snet_conv_fused_available(plan))
                if fused is True and not can:
                    raise RuntimeError(f'layer {ls.t}: no fused tensor-product kernels for this shape / mlp_mode')
                if fused is not False and can:
                    fpl = C.c_void_p()
                    _lib.check(self.lib.snet_fused_plan_create(plan, L.mlp_plan, self.fused_terms, C.byref(fpl)),
                               'snet_fused_plan_create')
                    L.fplan = fpl
                    # the reverse kernel also reverses the MLP's hidden layers (g_h2 never reaches memory)
                    L.mlp_tail = bool(self.lib.snet_fused_plan_has_mlp_tail(fpl)) and mlp_tail
                    nch = ls.si1.dim_out // 16   # chunk order of the fused reverse kernel's g_xe rows
                    cp = (C.c_int32 * nch)()
                    _lib.check(self.lib.snet_fused_plan_gxe_chunks(fpl, cp, nch), 'snet_fused_plan_gxe_chunks')
                    L.gxe_chunks = torch.tensor(list(cp), dtype=torch.int32, device=self.dev)
                L.fused_fwd = L.fplan is not None and fused in ('auto', True, 'fwd')
                L.fused_bwd = L.fplan is not None and fused in ('auto', True, 'bwd')
                L.tile_mode = int(self.lib.snet_fused_plan_tile_mode(L.fplan)) if L.fplan is not None else 0
                # scalar-output layer (the last one): its source-row gradient as a forward convolution of the transposed
                # product over the edges grouped by source -- gathers dout floats per edge instead of writing and
                # re-reading a dx-float g_xe row (model_spec.transposed_scalar_conv)
                L.tplan = None
                tr = transposed_scalar_conv(ls.conv) if (L.fused_bwd and transposed_conv and ls.t > 0) else None
                if tr is not None and ls.conv.irreps_out.dim < ls.conv.irreps_x.dim:
                    spec_t, kappa = tr
                    if spec_t.tag not in _lib.compiled_conv_tags():
                        from .jit import ensure_conv_shape
                        ensure_conv_shape(spec_t)
                    # the column factors and dead ranges come from the shape's own tables, like in the C++ sequencer
                    ttag, col = C.create_string_buffer(13), np.empty(ls.conv.weight_numel, np.float32)
                    dead, nd = (C.c_int32 * 32)(), C.c_int32()
                    _lib.check(self.lib.snet_conv_plan_transposed(plan, ttag, col.ctypes.data_as(C.c_void_p), C.cast(dead, C.c_void_p),
                                                                  32, C.byref(nd)), 'snet_conv_plan_transposed')
                    assert ttag.value.decode() == spec_t.tag, (ttag.value, spec_t.tag)
                    w2t = np.ascontiguousarray(np.asarray(hw[2], np.float32) * col[None, :], dtype=np.float32)
                    tpl, tmlp, tfp = C.c_void_p(), C.c_void_p(), C.c_void_p()
                    _lib.check(self.lib.snet_conv_plan_create(spec_t.tag.encode(), C.byref(tpl)), 'snet_conv_plan_create')
                    _lib.check(self.lib.snet_radial_mlp_plan_create(d[0], d[1], d[2], d[3], fp[0], fp[1],
                                                                    w2t.ctypes.data_as(C.POINTER(C.c_float)),
                                                                    ACT_ID[sp.act_radial], ACT_CST[sp.act_radial], 1, C.byref(tmlp)),
                               'snet_radial_mlp_plan_create')
                    _lib.check(self.lib.snet_fused_plan_create(tpl, tmlp, self.fused_terms, C.byref(tfp)), 'snet_fused_plan_create')
                    L.tplan, L.tmlp, L.tconv = tfp, tmlp, tpl
                    L.t_dead = [(dead[2 * i], dead[2 * i + 1]) for i in range(nd.value)]
                segs = (_lib.GateSeg * len(ls.gate.segs))()
                inv_act = {v: k for k, v in ACT_ID.items()}
                for i, s in enumerate(ls.gate.segs):
                    segs[i] = _lib.GateSeg(s.kind, s.in_off, s.out_off, s.mul, s.l, s.gate_off, s.act,
                                           ACT_CST[inv_act[s.act]])
                L.gate_segs = segs
                self.layers.append(L)
            self.h0_table = self.sc0_table = None
            if species_tables:
                h0, sc0 = species_only_tables(sp, sd, mi)
                self.h0_table = torch.from_numpy(h0).to(self.dev)
                self.sc0_table = None if sc0 is None else torch.from_numpy(sc0).to(self.dev)
            self.ro_v = self.ro_fcn = None
            if sp.readout_fcn_dims:   # `readout_as_fcn`: e3nn FullyConnectedNet on the final scalars (exact fp32 GEMMs + activations)
                d = sp.readout_fcn_dims
                ws = [np.ascontiguousarray(sd[f'readout_FCN.fcn.layer{i}.weight'] / np.sqrt(d[i]), dtype=np.float32) for i in range(len(d) - 1)]
                self.ro_fcn = type('F', (), {})()
                self.ro_fcn.dims = d
                self.ro_fcn.w = [torch.from_numpy(w).to(self.dev) for w in ws]
                self.ro_fcn.wt = [torch.from_numpy(np.ascontiguousarray(w.T)).to(self.dev) for w in ws]
                self.ro_fcn.act, self.ro_fcn.cst = ACT_ID[sp.readout_fcn_act], ACT_CST[sp.readout_fcn_act]
            else:
                if fold_readout:
                    v, self.ro_c = folded_readout(sp, sd, mi)
                    self.ro_v = torch.from_numpy(v).to(self.dev)
                self.ro1 = _Linear(sp.readout1, sd[sp.readout1.name], self.dev, split, mi, sd.get(sp.readout1.bias_name))
                self.ro2 = _Linear(sp.readout2, sd[sp.readout2.name], self.dev, split, mi, sd.get(sp.readout2.bias_name))
            sc_v, sh_v = sp.rescale_vectors(sd, mi)
            self.n_scale = len(sc_v)
            self.scale = torch.tensor(sc_v, dtype=torch.float32, device=self.dev)
            self.shift = torch.tensor(sh_v, dtype=torch.float32, device=self.dev)
            self.scale0 = float(sc_v[0])
        self.act_radial = ACT_ID[sp.act_radial]
        self.act_cst = ACT_CST[sp.act_radial]
        self.needs_species_rows = any(ls.sc is not None and ls.sc.n_species for ls in sp.layers)

    def __del__(self):
        try:
            for L in getattr(self, 'layers', []):
                if getattr(L, 'fplan', None) is not None:
                    self.lib.snet_fused_plan_destroy(L.fplan)
                if getattr(L, 'tplan', None) is not None:
                    self.lib.snet_fused_plan_destroy(L.tplan)
                    self.lib.snet_conv_plan_destroy(L.tconv)
                    self.lib.snet_radial_mlp_plan_destroy(L.tmlp)
                self.lib.snet_conv_plan_destroy(L.plan)
                if getattr(L, 'mlp_plan', None) is not None:
                    self.lib.snet_radial_mlp_plan_destroy(L.mlp_plan)
        except Exception:
            pass

    def kernel_times_ms(self) -> Dict[str, List[float]]:
        """Elapsed ms of every recorded span, grouped by kernel class (call after a device sync).  Classes
        suffixed '@side' ran on the second stream: their brackets overlap main-stream kernels, so their
        times are not exclusive and do not add up with the others to the step time."""
        out: Dict[str, List[float]] = {}
        for name, a, b in self.events or []:
            out.setdefault(name, []).append(a.elapsed_time(b))
        return out

    # ------------------------------------------------------------------ ops
    def _new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def _gemm(self, A, B, Cm, n_nodes, d, K, N, a_stride, a_off, c_stride, c_off, rows=None, acc=False):
        _lib.check(self.lib.snet_gemm(_ptr(A), _ptr(B), _ptr(Cm), n_nodes, d, K, N, a_stride, a_off, c_stride,
                                      c_off, _ptr(rows), int(acc), _stream()), 'snet_gemm')

    def _run_groups(self, groups, A, Cm, n, a_stride, c_stride, g: Graph, force_acc=False):
        for species, arr, cnt in groups:
            rows, m = None, n
            if species >= 0:
                rows = g.species_rows[species]
                m = rows.numel()
                if m == 0:
                    continue
            if force_acc:   # the same launches with every descriptor accumulating (built once per descriptor array)
                arr2 = self._acc_descs.get(id(arr))
                if arr2 is None:
                    arr2 = (_lib.GemmDesc * cnt)(*[_lib.GemmDesc(d.B, d.B_split, d.a_off, d.c_off, d.d, d.K, d.N, 1) for d in arr[:cnt]])
                    self._acc_descs[id(arr)] = arr2
                arr = arr2
            _lib.check(self.lib.snet_gemm_grouped(arr, cnt, _ptr(A), _ptr(Cm), m, a_stride, c_stride, _ptr(rows),
                                                  _stream()), 'snet_gemm_grouped')

    def _linear(self, lin: _Linear, x, n, g: Graph, out=None, accumulate=False):
        """y[:n] (+)= Linear(x[:n]) on ir_mul rows."""
        sp = lin.spec
        y = self._new(max(n, 0), sp.dim_out) if out is None else out
        if not accumulate:
            for off, ln in sp.zero_out:
                y[:, off:off + ln].zero_()
        self._run_groups(lin.groups_fwd, x, y, n, sp.dim_in, sp.dim_out, g, force_acc=accumulate)
        if lin.bias is not None and n > 0:
            _lib.check(self.lib.snet_add_row_bias(_ptr(y), _ptr(lin.bias), n, sp.dim_out, _stream()), 'snet_add_row_bias')
        return y

    def _linear_T(self, lin: _Linear, gy, n, g: Graph, out=None, accumulate=False):
        """g_x[:n] (+)= Linear^T(g_y[:n]).  The first (in,out) pair that reaches an input block
        overwrites it, later pairs accumulate; the species slices of one pair hit disjoint rows."""
        sp = lin.spec
        gx = out if out is not None else self._new(max(n, 0), sp.dim_in)
        if not accumulate:
            fed = {b.in_off for b in sp.blocks}
            for off, (m, l, _) in zip(sp.irreps_in.offsets(), sp.irreps_in):
                if off not in fed:
                    gx[:, off:off + m * (2 * l + 1)].zero_()
        self._run_groups(lin.groups_T, gy, gx, n, sp.dim_out, sp.dim_in, g, force_acc=accumulate)
        return gx

    def _mlp_fwd(self, L, emb, E, out=None):
        """radial weights w[E,wn]; returns (w, saved) where saved feeds _mlp_bwd."""
        dims = L.spec.mlp_dims
        if L.fused_mlp:
            w = self._new(E, dims[3]) if out is None else out
            _lib.check(self.lib.snet_radial_mlp_fwd(L.mlp_plan, _ptr(emb), E, _ptr(w), _stream()), 'snet_radial_mlp_fwd')
            return w, None
        zs, a = [], emb
        for i, wm in enumerate(L.mlp_w):
            z = self._new(E, dims[i + 1])
            self._gemm(a, wm, z, E, 1, dims[i], dims[i + 1], dims[i], 0, dims[i + 1], 0)
            if i + 1 < len(L.mlp_w):
                zs.append(z)
                a = self._new(E, dims[i + 1])
                _lib.check(self.lib.snet_act_fwd(_ptr(z), _ptr(a), z.numel(), self.act_radial, self.act_cst,
                                                 _stream()), 'snet_act_fwd')
                zs.append(a)
            else:
                return z, zs  # zs = [z1, a1, z2, a2, ...]

    def _mlp_bwd(self, L, emb, zs, g_w, g_emb_total, E):
        dims = L.spec.mlp_dims
        if L.fused_mlp:
            _lib.check(self.lib.snet_radial_mlp_bwd(L.mlp_plan, _ptr(emb), _ptr(g_w), E, _ptr(g_emb_total), _stream()),
                       'snet_radial_mlp_bwd')
            return
        g = g_w
        nl = len(L.mlp_w)
        for i in range(nl - 1, -1, -1):
            if i == 0:
                # accumulate straight into the all-layer edge-embedding gradient
                self._gemm(g, L.mlp_wt[0], g_emb_total, E, 1, dims[1], dims[0], dims[1], 0, dims[0], 0, None, True)
            else:
                ga = self._new(E, dims[i])
                self._gemm(g, L.mlp_wt[i], ga, E, 1, dims[i + 1], dims[i], dims[i + 1], 0, dims[i], 0)
                z = zs[2 * (i - 1)]
                _lib.check(self.lib.snet_act_bwd(_ptr(z), _ptr(ga), _ptr(ga), z.numel(), self.act_radial,
                                                 self.act_cst, _stream()), 'snet_act_bwd')
                g = ga

    # -------------------------------------------------------------- compute
    def compute(self, g: Graph, halo=None, want_atomic_virial: bool = False, keep: bool = False):
        """Energy, per-atom energies, dE/d(edge_vec), forces and virial for one graph.

        halo: object with forward(x[n_total,dim], n_local) filling ghost rows from their owners
        and reverse(gx[n_total,dim], n_local) accumulating ghost-row gradients into the owners
        (sevennet_amd.parallel.HaloExchange); None for a single-process graph."""
        lib, sp = self.lib, self.spec
        if self.needs_species_rows and g.species_rows is None:
            raise ValueError('graph was built without num_species but the model has a per-species self-connection')
        with torch.cuda.device(self.dev):
            c = self._begin(g, halo, keep)            # edge embedding, radial-weight streams, hidden radial layers
            x = self._forward_layers(c)                # interaction layers; what the reverse pass needs goes to c.saved
            e_atom, energy, g_x = self._readout(c, x)  # atomic energies, their sum, dE/dx of the last layer's output
            g_vec = self._reverse_layers(c, g_x)       # dE/d(edge vector), spherical and radial parts
            return self._forces(c, g_vec, e_atom, energy, want_atomic_virial)

    def _begin(self, g: Graph, halo, keep: bool):
        """first phase of compute: edge embedding, one radial row per undirected pair, the side stream's radial weights (unfused
        layers), the hidden radial layers of all fused layers; returns the evaluation's shared state"""
        from types import SimpleNamespace
        lib, sp = self.lib, self.spec
        st = _stream()
        N, NT, E = g.n_local, g.n_total, g.n_edges
        nb, nsh = sp.n_basis, self.nsh
        inter = {}
        emb, sh, dsh = self._new(E, nb), self._new(E, nsh), self._new(E, nsh * 3)
        with _Span(self, 'edge_embed_fwd'):
            _lib.check(lib.snet_edge_embed_fwd(C.byref(self.edge_params), self.coeffs, _ptr(g.edge_vec), E,
                                               _ptr(emb), _ptr(sh), _ptr(dsh), st), 'snet_edge_embed_fwd')
        # radial weights once per undirected pair (fused-MLP layers only: the unfused fallback keeps
        # its per-edge hidden activations for the reverse pass)
        pairs = g.w_row is not None and all(L.fused_mlp for L in self.layers)
        w_row = g.w_row if pairs else None
        if pairs:
            emb_p = self._new(g.n_pairs, nb)
            _lib.check(lib.snet_gather_rows(_ptr(emb), _ptr(g.pair_edge), _ptr(emb_p), g.n_pairs, nb, st),
                       'snet_gather_rows')
        # second stream: every layer's radial weights depend on the edge embedding only, so they are all
        # enqueued now and the main stream waits for layer t's event right before its tensor product
        side = None
        w_ready = {}
        any_fused = any(L.fused_fwd or L.fused_bwd for L in self.layers)
        if E > 0:   # work lists of the fused reverse kernels (the topology's only device sync: built once per Graph)
            for L in self.layers:
                if L.fused_bwd:
                    g.tiles(L.tile_mode)
        if self.overlap and E <= self.OVERLAP_MAX_EDGES and not any_fused and all(L.fused_mlp for L in self.layers):
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            side = self._side
            main = torch.cuda.current_stream()
            # every buffer the side stream touches is allocated on the MAIN stream and stays referenced until
            # the main stream has waited for the side stream again: no record_stream, hence no deferred frees
            # (with 10-GB blocks those made the caching allocator fall back to hipMalloc / hipFree)
            rows_w = g.n_pairs if pairs else E
            w_bufs = [self._new(rows_w, L_.spec.conv.weight_numel) for L_ in self.layers]
            wn_max = max(L_.spec.conv.weight_numel for L_ in self.layers)
            gw_bufs = [self._new(E * wn_max), self._new(E * wn_max)]
            gw_done = [None, None]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for t_, L_ in enumerate(self.layers):
                    with _Span(self, f'radial_mlp_fwd[wn={L_.spec.conv.weight_numel}]@side'):
                        self._mlp_fwd(L_, emb_p if pairs else emb, rows_w, out=w_bufs[t_])
                    ev = torch.cuda.Event()
                    ev.record(side)
                    w_ready[t_] = (w_bufs[t_], ev)
        d0 = sp.embed.dim_out
        x = None
        if keep or self.h0_table is None:
            x = self._new(NT, d0)  # ghost layer-0 features depend only on species (model_build.py:383-421)
            _lib.check(lib.snet_embed_rows(_ptr(self.embed_table), _ptr(g.types), _ptr(x), NT, d0, st),
                       'snet_embed_rows')
        if keep:
            inter['edge_embedding'], inter['edge_attr'], inter['x_embed'] = emb, sh, x[:N]
        saved = []
        # interior / boundary split of the convolutions around the ghost exchange: needs the brick's interior-first row
        # numbering, a halo with the split protocol, and edges (the reverse side walks a tile list)
        split = bool(halo is not None and g.n_interior and hasattr(halo, 'forward_start') and hasattr(halo, 'reverse_start')
                     and E > 0 and self.halo_split)
        # hidden activations of every layer's radial MLP, one row per pair, in ONE launch: the layers share the edge embedding
        # (five launches of a latency-bound kernel at brick sizes; the rows are kept for the reverse pass anyway)
        h2_rows = g.n_pairs if pairs else E
        h2_of = {t_: self._new(h2_rows, 64) for t_, L_ in enumerate(self.layers) if L_.fused_fwd or L_.fused_bwd}
        with _Span(self, 'radial_mlp_hidden_fwd'):
            ts = sorted(h2_of)
            for i in range(0, len(ts), 8):
                grp = ts[i:i + 8]
                plans = (C.c_void_p * len(grp))(*[self.layers[t_].mlp_plan for t_ in grp])
                outs = (C.c_void_p * len(grp))(*[h2_of[t_].data_ptr() for t_ in grp])
                _lib.check(lib.snet_radial_mlp_hidden_fwd_layers(plans, len(grp), _ptr(emb_p if pairs else emb), h2_rows, outs, st),
                           'snet_radial_mlp_hidden_fwd_layers')
        return SimpleNamespace(g=g, halo=halo, keep=keep, st=st, N=N, NT=NT, E=E, nb=nb, nsh=nsh, inter=inter, emb=emb, sh=sh, dsh=dsh,
                               pairs=pairs, w_row=w_row, emb_p=emb_p if pairs else None, side=side, w_ready=w_ready,
                               gw_bufs=gw_bufs if side is not None else None, gw_done=gw_done if side is not None else None,
                               x=x, saved=saved, split=split, h2_of=h2_of)

    def _forward_layers(self, c):
        """the interaction layers (interaction_blocks.py:41-76): SI1, ghost exchange, self-connection, convolution, SI2, gate"""
        lib, g, halo, keep, st, inter = self.lib, c.g, c.halo, c.keep, c.st, c.inter
        N, NT, E = c.N, c.NT, c.E
        emb, sh, pairs, emb_p, w_row, side, w_ready = c.emb, c.sh, c.pairs, c.emb_p, c.w_row, c.side, c.w_ready
        x, saved, split, h2_of = c.x, c.saved, c.split, c.h2_of
        for t, L in enumerate(self.layers):
            ls = L.spec
            n_in = NT if t == 0 else N  # rows of x that are valid
            with _Span(self, 'node_linear_fwd'):
                h = self._new(NT, ls.si1.dim_out)
                if t == 0 and self.h0_table is not None:   # species-only rows: fp64-evaluated table lookup
                    _lib.check(lib.snet_embed_rows(_ptr(self.h0_table), _ptr(g.types), _ptr(h), NT, ls.si1.dim_out, st),
                               'snet_embed_rows')
                else:
                    self._linear(L.si1, x, n_in, g, out=h)
            # ghost rows of h travel while the self-connection and the radial MLP (which do not read
            # them) run: hosts with a split exchange overlap the transfer with that work
            pending = None
            if t > 0 and halo is not None:
                with _Span(self, 'halo_fwd'):
                    if hasattr(halo, 'forward_start'):
                        pending = halo.forward_start(h, N)
                    else:
                        halo.forward(h, N)
            with _Span(self, 'node_linear_fwd'):
                if t == 0 and self.h0_table is not None:
                    sc = None
                    if self.sc0_table is not None:
                        sc = self._new(N, ls.gate.irreps_in.dim)
                        _lib.check(lib.snet_embed_rows(_ptr(self.sc0_table), _ptr(g.types), _ptr(sc), N, ls.gate.irreps_in.dim,
                                                       st), 'snet_embed_rows')
                else:
                    sc = self._linear(L.sc, x, N, g) if L.sc is not None else None
            dmid = ls.conv.irreps_out.dim
            m = self._new(N, dmid)
            if E == 0:
                m.zero_()
            else:   # columns of pruned (unread) paths are never written by the tensor-product kernel: defined zeros
                for off, ln in L.si2.zero_in:
                    m[:, off:off + ln].zero_()
            h2 = w = zs = None
            rows_w = g.n_pairs if pairs else E
            h2 = h2_of.pop(t, None)   # (computed before the layer loop)
            if not (L.fused_fwd and L.fused_bwd):  # someone still reads w[rows, wn]
                if side is not None:
                    (w, ev), zs = w_ready.pop(t), None
                    torch.cuda.current_stream().wait_event(ev)
                else:
                    with _Span(self, f'radial_mlp_fwd[wn={ls.conv.weight_numel}]'):
                        # one weight row per undirected pair when the graph carries the pair map
                        w, zs = self._mlp_fwd(L, emb_p, g.n_pairs) if pairs else self._mlp_fwd(L, emb, E)
            def conv_rows(a, b):   # the forward convolution of destination rows [a, b): pointer offsets, same kernels
                if b <= a:
                    return
                rp, mo = C.c_void_p(g.row_ptr.data_ptr() + 4 * a), C.c_void_p(m.data_ptr() + 4 * a * dmid)
                if L.fused_fwd:
                    with _Span(self, f'conv_fwd_fused[{ls.conv.tag}]'):
                        _lib.check(lib.snet_conv_fwd_fused(L.fplan, _ptr(h), _ptr(sh), _ptr(h2), _ptr(w_row), rp,
                                                           _ptr(g.src), b - a, L.scale, mo, st), 'snet_conv_fwd_fused')
                else:
                    with _Span(self, f'conv_fwd[{ls.conv.tag}]'):
                        _lib.check(lib.snet_conv_fwd(L.plan, _ptr(h), _ptr(sh), _ptr(w), _ptr(w_row), rp,
                                                     _ptr(g.src), b - a, L.scale, mo, st), 'snet_conv_fwd')
            # rows without a ghost source (bricks number them first) do not wait for the exchange
            n_int = g.n_interior if (pending is not None and split) else 0
            conv_rows(0, n_int)
            if pending is not None:
                with _Span(self, 'halo_fwd'):
                    halo.forward_finish(pending)
            conv_rows(n_int, N)
            if L.fused_bwd:
                w = None  # the reverse pass rebuilds its weight tiles from h2
            with _Span(self, 'node_linear_fwd'):
                # y = SI2(m) + self-connection: the linear map ACCUMULATES into the self-connection's rows (the gate kernel used to
                # add them: one read of sc and one write of y per node and layer more); y is kept for the reverse pass
                y = self._linear(L.si2, m, N, g) if sc is None else self._linear(L.si2, m, N, g, out=sc, accumulate=True)
            xo = self._new(N, ls.gate.irreps_out.dim)
            with _Span(self, 'gate_fwd'):
                _lib.check(lib.snet_gate_fwd(_ptr(y), None, _ptr(xo), N, ls.gate.irreps_in.dim,
                                             ls.gate.irreps_out.dim, L.gate_segs, len(ls.gate.segs), st), 'snet_gate_fwd')
            saved.append((h, w, zs, y, h2))
            if keep:
                inter[f'{t}_si1'], inter[f'{t}_conv'], inter[f'{t}_gate_in'], inter[f'{t}_x'] = h[:N], m, y, xo
            x = xo
        return x

    def _readout(self, c, x):
        """readout (plain, folded to one fp64 vector, or `readout_as_fcn`), rescale, energy sum; returns dE/dx beside them"""
        lib, sp, g, st, N = self.lib, self.spec, c.g, c.st, c.N
        e_atom = self._new(N)
        energy = torch.empty(1, dtype=torch.float64, device=self.dev)
        d_ro = sp.readout1.dim_in
        if self.ro_fcn is not None:   # readout_as_fcn: x -> act(x W0) cst -> ... -> e, then its reverse (nn/linear.py:145-180)
            F = self.ro_fcn
            d, nl = F.dims, len(F.w)
            zs, a = [], x
            for i in range(nl):
                z = self._new(N, d[i + 1])
                self._gemm(a, F.w[i], z, N, 1, d[i], d[i + 1], d[i], 0, d[i + 1], 0)
                if i + 1 < nl:
                    a = self._new(N, d[i + 1])
                    _lib.check(lib.snet_act_fwd(_ptr(z), _ptr(a), z.numel(), F.act, F.cst, st), 'snet_act_fwd')
                    zs.append(z)
            _lib.check(lib.snet_rescale_reduce(_ptr(z), _ptr(g.types), _ptr(self.scale), _ptr(self.shift),
                                               self.n_scale, N, _ptr(e_atom), _ptr(energy), st), 'snet_rescale_reduce')
            gz = self._new(N, 1)
            if self.n_scale > 1:
                _lib.check(lib.snet_embed_rows(_ptr(self.scale), _ptr(g.types), _ptr(gz), N, 1, st), 'snet_embed_rows')
            else:
                gz.fill_(self.scale0)
            for i in range(nl - 1, -1, -1):
                ga = self._new(N, d[i])
                self._gemm(gz, F.wt[i], ga, N, 1, d[i + 1], d[i], d[i + 1], 0, d[i], 0)
                if i > 0:
                    _lib.check(lib.snet_act_bwd(_ptr(zs[i - 1]), _ptr(ga), _ptr(ga), ga.numel(), F.act, F.cst, st), 'snet_act_bwd')
                gz = ga
            g_x = gz
        elif self.ro_v is not None:   # folded readout: fp64 dot product + rescale + energy sum in one pass
            _lib.check(lib.snet_readout_energy(_ptr(x), N, d_ro, _ptr(self.ro_v), self.ro_c, _ptr(g.types), _ptr(self.scale),
                                               _ptr(self.shift), self.n_scale, _ptr(e_atom), _ptr(energy), st), 'snet_readout_energy')
            g_x = self._new(N, d_ro)
            _lib.check(lib.snet_readout_grad(_ptr(self.ro_v), d_ro, _ptr(g.types), _ptr(self.scale), self.n_scale, N, _ptr(g_x),
                                             st), 'snet_readout_grad')
        else:
            h1 = self._linear(self.ro1, x, N, g)
            e_sc = self._linear(self.ro2, h1, N, g)
            _lib.check(lib.snet_rescale_reduce(_ptr(e_sc), _ptr(g.types), _ptr(self.scale), _ptr(self.shift),
                                               self.n_scale, N, _ptr(e_atom), _ptr(energy), st), 'snet_rescale_reduce')
            # ---------------- reverse pass: dE/d(e_scaled) = scale[type]
            g_e = self._new(N, 1)
            if self.n_scale > 1:
                _lib.check(lib.snet_embed_rows(_ptr(self.scale), _ptr(g.types), _ptr(g_e), N, 1, st), 'snet_embed_rows')
            else:
                g_e.fill_(self.scale0)
            g_h1 = self._linear_T(self.ro2, g_e, N, g)
            g_x = self._linear_T(self.ro1, g_h1, N, g)
        return e_atom, energy, g_x

    def _reverse_layers(self, c, g_x):
        """reverse pass through the interaction layers and the edge embedding; returns g_vec[E, 3] = dE/d(edge vector)"""
        lib, g, halo, st = self.lib, c.g, c.halo, c.st
        N, NT, E, nb, nsh = c.N, c.NT, c.E, c.nb, c.nsh
        emb, sh, dsh, w_row, side, gw_bufs, gw_done = c.emb, c.sh, c.dsh, c.w_row, c.side, c.gw_bufs, c.gw_done
        saved, split = c.saved, c.split
        sh_T = None   # spherical harmonics in source-grouped edge order (transposed scalar convolution)
        g_vec = torch.zeros(E, 3, dtype=torch.float32, device=self.dev)  # spherical part, all layers
        g_emb = torch.zeros(E, nb, dtype=torch.float32, device=self.dev)
        # fp16 operands of the fused reverse kernels: row maxima of the source rows and of the incoming gradient bound every
        # edge's g_w, from which the kernel derives that edge's power-of-two scale (no overflow possible).  The source-row
        # bounds of ALL layers come from one launch here (the rows have been complete since the forward pass).
        x_max_of = {}
        if self.fused_terms == 4 and E > 0:
            ts = [t_ for t_, L_ in enumerate(self.layers) if L_.fused_bwd]
            for t_ in ts:
                x_max_of[t_] = self._new(NT)
            with _Span(self, 'row_bounds'):
                for i in range(0, len(ts), 8):
                    grp = ts[i:i + 8]
                    k = len(grp)
                    _lib.check(lib.snet_row_absmax_multi((C.c_void_p * k)(*[saved[t_][0].data_ptr() for t_ in grp]), (C.c_int64 * k)(*[NT] * k),
                                                         (C.c_int32 * k)(*[self.layers[t_].spec.si1.dim_out for t_ in grp]),
                                                         (C.c_void_p * k)(*[x_max_of[t_].data_ptr() for t_ in grp]), k, st),
                               'snet_row_absmax_multi')
        for t in range(len(self.layers) - 1, -1, -1):
            L = self.layers[t]
            ls = L.spec
            h, w, zs, y, h2 = saved[t]
            g_y = self._new(N, ls.gate.irreps_in.dim)
            # fp16 operands of the fused reverse kernel: row bound of g_m = SI2^T g_y through the narrower g_y and SI2's
            # largest row norm (Cauchy-Schwarz), taken while the gate's reverse pass has the row in registers
            g_max = self._new(N) if (L.fused_bwd and self.fused_terms == 4 and E > 0) else None
            with _Span(self, 'gate_bwd'):
                _lib.check(lib.snet_gate_bwd_norm(_ptr(y), _ptr(g_x), _ptr(g_y), N, ls.gate.irreps_in.dim, ls.gate.irreps_out.dim,
                                                  L.gate_segs, len(ls.gate.segs), L.si2.t_norm if g_max is not None else 0.0,
                                                  _ptr(g_max), st), 'snet_gate_bwd_norm')
            with _Span(self, 'node_linear_bwd'):
                g_m = self._linear_T(L.si2, g_y, N, g)
            # layer 0: inputs depend on species only -> no source-row gradient needed
            use_t = t > 0 and getattr(L, 'tplan', None) is not None and E > 0
            g_xe = self._new(E, ls.si1.dim_out) if (t > 0 and not use_t) else None
            g_w = g_h2 = None
            # reverse split (t > 0, fused kernels): boundary tiles first -- they hold every edge with a ghost source --
            # then the ghost rows of g_h, whose exchange starts at once; interior tiles, the local rows and sc^T run under it
            rsplit = split and t > 0 and L.fused_bwd
            pending = None
            g_h = self._new(NT, ls.si1.dim_out) if t > 0 else None
            if use_t:
                src_t, w_row_t = g.by_source(lib, st)
                if L.t_dead:
                    g_h.zero_()
                if sh_T is None:
                    sh_T = self._new(E, nsh)
                    _lib.check(lib.snet_gather_rows(_ptr(sh), _ptr(g.eperm), _ptr(sh_T), E, nsh, st), 'snet_gather_rows')

            def gh_rows(a, b):
                """source rows [a, b) of g_h: transposed scalar convolution over the edges grouped by source, or the segment
                sum of the per-edge rows the reverse kernel wrote"""
                if b <= a:
                    return
                cp, go = C.c_void_p(g.col_ptr.data_ptr() + 4 * a), C.c_void_p(g_h.data_ptr() + 4 * a * ls.si1.dim_out)
                if use_t:
                    with _Span(self, f'conv_bwd_node[transposed {ls.conv.tag}]'):
                        _lib.check(lib.snet_conv_fwd_fused(L.tplan, _ptr(g_m), _ptr(sh_T), _ptr(h2), _ptr(w_row_t), cp, _ptr(src_t),
                                                           b - a, L.scale, go, st), 'snet_conv_fwd_fused')
                    return
                with _Span(self, 'conv_bwd_node[segment_sum]'):
                    if L.fused_bwd:   # the fused kernel's g_xe rows: chunk order undone while summing
                        _lib.check(lib.snet_segment_sum_rows_chunked(_ptr(g_xe), cp, _ptr(g.eperm), b - a, ls.si1.dim_out,
                                                                     _ptr(L.gxe_chunks), go, st), 'snet_segment_sum_rows_chunked')
                    else:
                        _lib.check(lib.snet_segment_sum_rows(_ptr(g_xe), cp, _ptr(g.eperm), b - a, ls.si1.dim_out, go, st),
                                   'snet_segment_sum_rows')

            if L.fused_bwd:
                g_h2 = None if L.mlp_tail else self._new(E, 64)
                x_max = x_max_of.pop(t, None)   # (computed before the layer loop)

                def bwd_tiles(tp_, tn_, nt_):
                    if nt_ <= 0:
                        return
                    with _Span(self, f'conv_bwd_fused[{ls.conv.tag}]'):
                        _lib.check(lib.snet_conv_bwd_fused(L.fplan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(h2), _ptr(w_row),
                                                           _ptr(g.row_ptr), _ptr(g.src), _ptr(tp_), _ptr(tn_), nt_, L.scale,
                                                           _ptr(g_m), _ptr(g_xe), _ptr(g_h2),
                                                           _ptr(emb) if L.mlp_tail else None, _ptr(g_emb) if L.mlp_tail else None,
                                                           _ptr(g_vec), _ptr(x_max), _ptr(g_max), st),
                                   'snet_conv_bwd_fused')
                if rsplit:
                    (tpi, tni, nti), (tpb, tnb, ntb) = g.tiles_split(L.tile_mode)
                    if use_t:                  # g_h does not depend on the reverse kernel: ghost rows straight away
                        gh_rows(N, NT)
                    else:
                        bwd_tiles(tpb, tnb, ntb)
                        gh_rows(N, NT)
                    with _Span(self, 'halo_rev'):
                        pending = halo.reverse_start(g_h, N)
                    if use_t:
                        bwd_tiles(*g.tiles(L.tile_mode))
                    else:
                        bwd_tiles(tpi, tni, nti)
                    gh_rows(0, N)
                elif E > 0:
                    bwd_tiles(*g.tiles(L.tile_mode))
            else:
                if side is not None:  # double-buffered: the MLP reverse of layer t+2 may still be reading this one
                    if gw_done[t & 1] is not None:
                        torch.cuda.current_stream().wait_event(gw_done[t & 1])
                    g_w = gw_bufs[t & 1][:E * ls.conv.weight_numel].view(E, ls.conv.weight_numel)
                else:
                    g_w = self._new(E, ls.conv.weight_numel)
                with _Span(self, f'conv_bwd_edge[{ls.conv.tag}]'):
                    _lib.check(lib.snet_conv_bwd_edge_vec(L.plan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(w), _ptr(w_row),
                                                          _ptr(g.row_ptr), _ptr(g.src), N, L.scale, _ptr(g_m), _ptr(g_w),
                                                          _ptr(g_xe), _ptr(g_vec), st), 'snet_conv_bwd_edge_vec')
            # the source-row gradient goes first so that its ghost rows can travel to their owners while
            # the radial MLP's reverse pass (independent of them) runs
            if t > 0 and not rsplit:
                gh_rows(0, NT)
                if halo is not None:
                    with _Span(self, 'halo_rev'):
                        if hasattr(halo, 'reverse_start'):
                            pending = halo.reverse_start(g_h, N)
                        else:
                            halo.reverse(g_h, N)
            del g_xe
            if L.fused_bwd and L.mlp_tail:
                pass
            elif L.fused_bwd:
                with _Span(self, 'radial_mlp_hidden_bwd'):
                    _lib.check(lib.snet_radial_mlp_hidden_bwd(L.mlp_plan, _ptr(emb), _ptr(g_h2), E, _ptr(g_emb), st),
                               'snet_radial_mlp_hidden_bwd')
            elif side is not None:  # g_w is complete on the main stream; its consumer runs beside what follows
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    with _Span(self, f'radial_mlp_bwd[wn={ls.conv.weight_numel}]@side'):
                        self._mlp_bwd(L, emb, zs, g_w, g_emb, E)
                    gw_done[t & 1] = torch.cuda.Event()
                    gw_done[t & 1].record(side)
            else:
                with _Span(self, f'radial_mlp_bwd[wn={ls.conv.weight_numel}]'):
                    self._mlp_bwd(L, emb, zs, g_w, g_emb, E)
            del g_w, g_h2
            if t == 0:
                break
            # g_x = sc^T g_y + SI1^T g_h: the self-connection's share does not need the ghost gradients, so it runs
            # while they travel (reverse_start above put the exchange on the halo's stream)
            with _Span(self, 'node_linear_bwd'):
                g_x = self._linear_T(L.sc, g_y, N, g) if L.sc is not None else None
            if pending is not None:
                with _Span(self, 'halo_rev'):
                    halo.reverse_finish(pending, g_h)
            with _Span(self, 'node_linear_bwd'):
                g_x = self._linear_T(L.si1, g_h, N, g, out=g_x, accumulate=g_x is not None)
            saved[t] = None
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        _lib.check(lib.snet_edge_embed_bwd(C.byref(self.edge_params), self.coeffs, _ptr(g.edge_vec), E, _ptr(g_emb),
                                           None, _ptr(g_vec), 1, st), 'snet_edge_embed_bwd')
        return g_vec

    def _forces(self, c, g_vec, e_atom, energy, want_atomic_virial: bool):
        """forces / virial from the edge gradients (force_output.py:171-230), ghost contributions folded into their owners"""
        lib, g, halo, st, N, NT, E = self.lib, c.g, c.halo, c.st, c.N, c.NT, c.E
        keep, inter = c.keep, c.inter
        forces = self._new(NT, 3)
        vir_atom = self._new(NT, 6) if want_atomic_virial else None
        virial = torch.empty(6, dtype=torch.float64, device=self.dev)
        _lib.check(lib.snet_edge_force(_ptr(g_vec), _ptr(g.edge_vec), _ptr(g.row_ptr), _ptr(g.col_ptr),
                                       _ptr(g.eperm), NT, E, _ptr(forces), _ptr(vir_atom), _ptr(virial), st),
                   'snet_edge_force')
        if halo is not None:  # fold ghost-atom force (and atomic virial) contributions into their owners: ONE exchange
            with _Span(self, 'halo_rev'):
                if vir_atom is None:
                    halo.reverse(forces, N)
                else:
                    fv = torch.cat([forces, vir_atom], 1).contiguous()
                    halo.reverse(fv, N)
                    forces, vir_atom = fv[:, :3].contiguous(), fv[:, 3:].contiguous()
        if g.order is not None:  # report dE/dr in the caller's edge order
            tmp = torch.empty_like(g_vec)
            tmp[g.order] = g_vec
            g_vec = tmp
        out = dict(energy=energy, atomic_energy=e_atom, dE_dr=g_vec, forces=forces[:N], virial=virial)
        if vir_atom is not None:
            out['atomic_virial'] = vir_atom[:N]
        if keep:
            out['inter'] = inter
        return out
