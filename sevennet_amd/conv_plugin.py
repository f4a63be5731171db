"""Drop-in for the reference's tensor-product accelerator plug-in point.

The reference swaps `IrrepsScatterGatterFusedConvolution.convolution_cls`
(sevenn/nn/convolution.py:145-284) for a third-party fused gather-TP-scatter op
(flash_helper.py:33-48, cue_helper.py:201-248, oeq_helper.py:74-83).  This module
is the MI355X equivalent:

    conv = HipUvuConvolution(irreps_in1, irreps_in2, irreps_out, instructions)
    out  = conv(x[N,dx], edge_filter[E,nsh], weight[E,wn], edge_src[E] i32, edge_dst[E] i32)

with the reference's semantics (convolution.py:270-276): e3nn `mul_ir` layout,
out[i] = sum_{e: dst=i} TP_uvu(x[src_e], filter_e; weight_e), differentiable wrt
x, edge_filter and weight (first order: forces are taken by autograd through it).
`patch_convolution(irreps_convolution)` mirrors the reference's patch helpers.

All arithmetic runs in libsnet_hip.so through the C ABI; there is no fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np
import torch

from . import _lib
from .irreps import Irreps, irmul_to_mulir_index, mulir_to_irmul_index
from .model_spec import ConvPath, ConvSpec


def _irreps(obj) -> Irreps:
    return obj if isinstance(obj, Irreps) else Irreps(str(obj))


def conv_spec_from_instructions(irreps_in1, irreps_in2, irreps_out, instructions: Sequence) -> ConvSpec:
    """ConvSpec for an explicit e3nn-style instruction list [(i_in1, i_in2, i_out, 'uvu', ...)];
    weight columns follow the list order (convolution.py:69,94), output blocks are `irreps_out`
    (= sorted irreps_mid, one block per instruction)."""
    x, sh, mid = _irreps(irreps_in1), _irreps(irreps_in2), _irreps(irreps_out)
    merged = mid.simplified()
    m_off = merged.offsets()
    where, cur, ch = [], -1, 0
    for (mul, l, p) in mid:
        if cur < 0 or (merged[cur][1], merged[cur][2]) != (l, p) or ch + mul > merged[cur][0]:
            cur += 1
            ch = 0
        where.append((cur, ch))
        ch += mul
    x_off, sh_off = x.offsets(), sh.offsets()
    paths: List[ConvPath] = []
    w_off = 0
    for ins in instructions:
        i, j, k = int(ins[0]), int(ins[1]), int(ins[2])
        if len(ins) > 3 and ins[3] != 'uvu':
            raise NotImplementedError(f"only 'uvu' instructions are supported, got {ins[3]!r}")
        mul, l1, p1 = x[i]
        _, l2, p2 = sh[j]
        mo, l3, p3 = mid[k]
        if mo != mul or p3 != p1 * p2 or not abs(l1 - l2) <= l3 <= l1 + l2:
            raise ValueError(f'inconsistent instruction {ins}')
        blk, c = where[k]
        paths.append(ConvPath(i, j, l1, l2, l3, mul, w_off, x_off[i], sh_off[j], m_off[blk], merged[blk][0], c))
        w_off += mul
    return ConvSpec(x, sh, mid, merged, paths, w_off)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _EdgePlan:
    """CSR-by-destination view of one (edge_src, edge_dst) pair plus the grouping by source that the
    reverse pass needs.  The reference hands the SAME edge tensors to every interaction layer of a step
    (and they arrive sorted by center from its neighbor-list builders), so the plan is built once and
    shared by all HipUvuConvolution modules; sorted input needs no row permutation at all."""

    def __init__(self, edge_src, edge_dst, n_nodes: int):
        dev = edge_dst.device
        self.keep = (edge_src, edge_dst)  # strong refs: their storage cannot be recycled while this is cached
        dst = edge_dst.to(torch.int64)
        E = dst.numel()
        self.order = None
        if E > 1 and bool((dst[1:] < dst[:-1]).any()):
            self.order = torch.sort(dst, stable=True).indices
            dst = dst[self.order]
        src = edge_src.to(torch.int64)
        if self.order is not None:
            src = src[self.order]
        self.src = src.to(torch.int32).contiguous()
        self.dst = dst.to(torch.int32).contiguous()
        row_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
        col_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
        if E:
            row_ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_nodes), 0)
            col_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=n_nodes), 0)
            self.eperm = torch.sort(src, stable=True).indices.to(torch.int32)
        else:
            self.eperm = torch.zeros(0, dtype=torch.int32, device=dev)
        self.row_ptr, self.col_ptr = row_ptr.to(torch.int32), col_ptr.to(torch.int32)


_PLAN_CACHE: dict = {}
_PLAN_CACHE_SIZE = 2


def _edge_plan(edge_src, edge_dst, n_nodes: int) -> _EdgePlan:
    key = (edge_src.data_ptr(), edge_dst.data_ptr(), edge_src.numel(), n_nodes, edge_src._version, edge_dst._version,
           str(edge_dst.device), edge_src.dtype, edge_dst.dtype)
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        plan = _EdgePlan(edge_src, edge_dst, n_nodes)
        while len(_PLAN_CACHE) >= _PLAN_CACHE_SIZE:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
        _PLAN_CACHE[key] = plan
    return plan


class _UvuConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sh, w, edge_src, edge_dst, mod):
        lib = mod.lib
        if not (x.is_cuda and sh.is_cuda and w.is_cuda):
            raise RuntimeError('HipUvuConvolution needs ROCm tensors (no CPU path exists)')
        N = x.shape[0]
        ep = _edge_plan(edge_src, edge_dst, N)
        sh_s = sh.detach().float()
        w_s = w.detach().float()
        if ep.order is not None:
            sh_s, w_s = sh_s[ep.order], w_s[ep.order]
        sh_s, w_s = sh_s.contiguous(), w_s.contiguous()
        x_im = torch.empty(N, mod.dx, dtype=torch.float32, device=x.device)
        xc = x.detach().float().contiguous()
        _lib.check(lib.snet_permute_cols(_p(xc), _p(mod.idx_in), _p(x_im), N, mod.dx, _st()), 'snet_permute_cols')
        out_im = torch.empty(N, mod.dout, dtype=torch.float32, device=x.device)
        _lib.check(lib.snet_conv_fwd(mod.plan, _p(x_im), _p(sh_s), _p(w_s), None, _p(ep.row_ptr), _p(ep.src), N, 1.0,
                                     _p(out_im), _st()), 'snet_conv_fwd')
        out = torch.empty_like(out_im)
        _lib.check(lib.snet_permute_cols(_p(out_im), _p(mod.idx_out_inv), _p(out), N, mod.dout, _st()), 'snet_permute_cols')
        ctx.mod, ctx.ep = mod, ep
        ctx.save_for_backward(x_im, sh_s, w_s)
        return out

    @staticmethod
    def backward(ctx, g_out):
        mod, lib, ep = ctx.mod, ctx.mod.lib, ctx.ep
        x_im, sh_s, w_s = ctx.saved_tensors
        N, E = x_im.shape[0], sh_s.shape[0]
        dev = x_im.device
        g_im = torch.empty(N, mod.dout, dtype=torch.float32, device=dev)
        g_c = g_out.float().contiguous()
        _lib.check(lib.snet_permute_cols(_p(g_c), _p(mod.idx_out), _p(g_im), N, mod.dout, _st()), 'snet_permute_cols')
        g_w = torch.empty(E, mod.wn, dtype=torch.float32, device=dev)
        g_sh = torch.zeros(E, mod.nsh, dtype=torch.float32, device=dev)
        g_xe = torch.empty(E, mod.dx, dtype=torch.float32, device=dev)
        # per-edge gradients (incl. each edge's share of d/dx[src]), then one segmented sum per source row
        _lib.check(lib.snet_conv_bwd_edge(mod.plan, _p(x_im), _p(sh_s), _p(w_s), None, _p(ep.row_ptr), _p(ep.src), N, 1.0,
                                          _p(g_im), _p(g_w), _p(g_xe), _p(g_sh), _st()), 'snet_conv_bwd_edge')
        g_x_im = torch.empty(N, mod.dx, dtype=torch.float32, device=dev)
        _lib.check(lib.snet_segment_sum_rows(_p(g_xe), _p(ep.col_ptr), _p(ep.eperm), N, mod.dx, _p(g_x_im), _st()),
                   'snet_segment_sum_rows')
        g_x = torch.empty_like(g_x_im)
        _lib.check(lib.snet_permute_cols(_p(g_x_im), _p(mod.idx_in_inv), _p(g_x), N, mod.dx, _st()), 'snet_permute_cols')
        if ep.order is not None:  # back to the caller's edge order
            g_w_o, g_sh_o = torch.empty_like(g_w), torch.empty_like(g_sh)
            g_w_o[ep.order] = g_w
            g_sh_o[ep.order] = g_sh
            g_w, g_sh = g_w_o, g_sh_o
        return g_x, g_sh, g_w, None, None, None


class HipUvuConvolution(torch.nn.Module):
    """`convolution_cls` for IrrepsScatterGatterFusedConvolution (convolution.py:237-247)."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, shared_weights: bool = False,
                 internal_weights: bool = False, **_ignored):
        super().__init__()
        if shared_weights or internal_weights:
            raise NotImplementedError('HipUvuConvolution: per-edge external weights only')
        self.lib = _lib.load()
        self.spec = conv_spec_from_instructions(irreps_in1, irreps_in2, irreps_out, instructions)
        # any irreps the reference's hook may construct (convolution.py:237-247): a shape that is not in the library
        # yet is generated, compiled with hipcc and registered now (cached on disk: sevennet_amd/jit.py)
        from .jit import ensure_conv_shape
        ensure_conv_shape(self.spec)
        plan = C.c_void_p()
        _lib.check(self.lib.snet_conv_plan_create(self.spec.tag.encode(), C.byref(plan)), 'snet_conv_plan_create')
        self.plan = plan
        self.dx, self.dout = self.spec.irreps_x.dim, self.spec.irreps_out.dim
        self.nsh, self.wn = self.spec.irreps_sh.dim, self.spec.weight_numel
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32)  # noqa: E731
        self.register_buffer('idx_in', i32(mulir_to_irmul_index(self.spec.irreps_x)), persistent=False)
        self.register_buffer('idx_in_inv', i32(irmul_to_mulir_index(self.spec.irreps_x)), persistent=False)
        self.register_buffer('idx_out', i32(mulir_to_irmul_index(self.spec.irreps_out)), persistent=False)
        self.register_buffer('idx_out_inv', i32(irmul_to_mulir_index(self.spec.irreps_out)), persistent=False)

    def forward(self, x, edge_filter, weight, edge_src, edge_dst):
        if self.idx_in.device != x.device:
            self.to(x.device)
        return _UvuConvFn.apply(x, edge_filter, weight, edge_src, edge_dst, self)


def is_hip_available() -> bool:
    try:
        _lib.load()
    except Exception:  # noqa: BLE001
        return False
    return torch.cuda.is_available()


def patch_convolution(irreps_convolution):
    """Analogue of sevenn.nn.flash_helper.patch_convolution (flash_helper.py:33-48): turn a not yet
    instantiated reference `IrrepsConvolution` into the fused variant backed by this library."""
    from sevenn.nn.convolution import IrrepsScatterGatterFusedConvolution  # reference package, if installed
    assert not irreps_convolution.layer_instantiated
    ret = IrrepsScatterGatterFusedConvolution.from_irreps_convolution(irreps_convolution)
    ret.convolution_cls = HipUvuConvolution
    return ret
