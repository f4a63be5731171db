"""HIP source generator for the FUSED radial-weight + tensor-product kernels (gfx950).

What the reference's accelerator plug-ins promise for this op (sevenn/nn/flash_helper.py:43,
sevenn/nn/convolution.py:118-141: no `weight[E,wn]` / `message[E,dmid]` tensor survives) is done here one
step further: the radial MLP's last layer  w_e = h2_e @ W2  (64 -> weight_numel, the largest FLOP
term of the model) is evaluated INSIDE the tensor-product kernels on
v_mfma_f32_16x16x32_bf16, so neither w[E,wn] nor its gradient g_w[E,wn] ever exists in HBM.

Two kernels per convolution shape (all channel multiplicities must be multiples of 16):

  conv_bwdf  reverse pass.  One wavefront = one 16-edge tile of a destination node
             (lane & 15 = edge, lane >> 4 = group of 4 channels).  Per 16-column tile of W2:
               w^T[col, edge]  = W2^T-tile (A, from LDS) x h2^T (B, registers)        -> 4 channels per lane
               tensor-product reverse per lane (sparse CG as immediates): g_w, d/dY, d/dx[src]
               g_h2^T[k, edge] += W2[k, cols] (A, from LDS) x g_w^T (B = the registers just produced)
             -- accumulator layout == next operand layout, no transposition, no LDS round trip.
             d/dY is accumulated per lane over all channels (no cross-lane reduction per edge).
             Epilogue (FusedTail): the MLP's two hidden layers reversed on the g_h2 accumulators -- same layout
             trick, fragments staged through the idle slab buffers -- so g_h2 is not written either.
             Outputs: g_xe[E,dx] (per-edge source-row gradient), g_vec[E,3] +=, g_emb[E,nb] += (or g_h2[E,64]).
  conv_fwdf  forward.  One wavefront = one destination node, up to two 16-edge tiles per pass
             (lane & 15 = channel, lane >> 4 = group of 4 edges): w[edge, col] = h2 (A) x W2-tile (B),
             tensor product accumulated over the lane's edges, two permlane adds per output value.
             The 16 source rows' slice of a (block, tile) stage is loaded one stage ahead and parked in
             wave-private LDS; output rows leave through LDS as 16-byte stores.

W2 reaches both kernels as one stream of pre-split bf16 MFMA fragments in sub-step order
(`sub_cols`, packed by snet_fused_plan_create); a workgroup's wavefronts walk the stream in
lockstep through a double-buffered LDS slab, so each fragment is fetched once per workgroup.
"""
from __future__ import annotations

import struct
from typing import Dict, List

from .codegen import OPTS, _Cat, _f, _path_terms, _sum_expr
from .model_spec import ConvSpec


def fusable(spec: ConvSpec) -> bool:
    return all(mul % 16 == 0 for mul, _, _ in spec.irreps_x) and spec.irreps_sh.dim <= 16 and len(spec.paths) > 0


def schedule(spec: ConvSpec):
    """-> (cats, subs): subs = list of (ci, ct-relative pair list); the flat sub-step order is
    for cat: for ct: for pair.  Returns per-cat pair lists and the flat (col_a, col_b) table."""
    cats = []
    for i in range(len(spec.irreps_x)):
        cat = _Cat(spec, i, 0)
        if cat.paths:
            cats.append(cat)
    pairs_of, cols = [], []
    for cat in cats:
        ps = [pi for pi, _ in cat.paths]
        pairs = [(ps[k], ps[k + 1] if k + 1 < len(ps) else None) for k in range(0, len(ps), 2)]
        pairs_of.append(pairs)
        for ct in range(cat.mul // 16):
            for pa, pb in pairs:
                cols.append((spec.paths[pa].w_off + 16 * ct, spec.paths[pb].w_off + 16 * ct if pb is not None else -1))
    return cats, pairs_of, cols


def schedule_bwd(spec: ConvSpec):
    """Blocks of the reverse kernel -> (per x block: dict(cat, U, ncb, steps), flat (col_a, col_b) table of its weight stream).
    A block covers U channel tiles of one x block; `steps` lists its sub-steps as pairs of (path, member tile u) -- or None
    for an empty second tile.  U = 2 where the x block has an odd number of paths and an even number of channel tiles: the
    partnerless path's tiles of the block's two channel tiles then share a sub-step (and the g_out entries of both tiles fit
    the wave's 32-row LDS buffer).  Stream order: for x block: for
    block: for sub-step.  SNET_CODEGEN_OPTS=nopairct=1 keeps U = 1 everywhere (the round-2 schedule)."""
    out, cols = [], []
    # first interaction layer of an lmax-2 model (one x block, three paths): it keeps one channel tile per block.  Two-tile
    # blocks push its 8-wave configuration past 128 registers (2.46 ms against 2.06 stand-alone); in 4-wave workgroups they
    # win stand-alone (1.83 ms) but not inside the step with the hidden-layer tail (1.77 against 1.72 ms)
    n_cats = sum(1 for i in range(len(spec.irreps_x)) if _Cat(spec, i, 0).paths)
    for i in range(len(spec.irreps_x)):
        cat = _Cat(spec, i, 0)
        if not cat.paths:
            continue
        ps = [pi for pi, _ in cat.paths]
        nct = cat.mul // 16
        # (a two-tile block parks the g_out entries of both tiles: only where they fit 32 rows of the wave's LDS buffer --
        # the lmax-3 middle layers would otherwise lose a workgroup per CU to LDS)
        entries = sum(2 * p.l3 + 1 for _, p in cat.paths)
        if len(ps) % 2 == 1 and nct % 2 == 0 and 2 * entries <= 32 and n_cats > 1 and not OPTS.get('nopairct'):
            full = [(ps[k], ps[k + 1]) for k in range(0, len(ps) - 1, 2)]
            steps = [((a, 0), (b, 0)) for a, b in full] + [((a, 1), (b, 1)) for a, b in full] + [((ps[-1], 0), (ps[-1], 1))]
            U = 2
        else:
            steps = [((ps[k], 0), (ps[k + 1], 0) if k + 1 < len(ps) else None) for k in range(0, len(ps), 2)]
            U = 1
        ncb = nct // U
        out.append(dict(cat=cat, U=U, ncb=ncb, steps=steps))
        for cb in range(ncb):
            for ta, tb in steps:
                ca = spec.paths[ta[0]].w_off + 16 * (U * cb + ta[1])
                cb_ = spec.paths[tb[0]].w_off + 16 * (U * cb + tb[1]) if tb is not None else -1
                cols.append((ca, cb_))
    return out, cols


def _emit_staging(A, lines: str):
    """stage_load(s, b) / stage_store(b): the next sub-step's `lines` fragment lines (1 KB each) travel
    global -> LDS either through registers (load early, ds_write late) or (GLDS) by gfx950's direct
    global_load_lds (no staging VGPRs; lane i of a wave lands at base + 16 i)"""
    A('  u32x4 st[GLDS ? 1 : NST];')
    A('  // (raw buffer resource over the fragment stream: stride 0, byte offsets, gfx9 data-format word)')
    A('  const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(slabs), 0, 0x7fffffff, 0x00020000);')
    A('  auto stage_load = [&](int s, int b) {')
    A('    if constexpr (GLDS) {')
    A(f'      for (int l = wave; l < {lines}; l += NWV)')
    A('        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(slabs + (size_t)s * (LPS * 64) + l * 64 + lane),')
    A('                                         (__attribute__((address_space(3))) void *)(&slab[b][l * 64]), 16, 0, 0);')
    A('    } else {')
    # (buffer loads: the stream's base sits in a scalar resource, the sub-step and line offsets in the scalar offset, the lane offset
    # in ONE vector register for the whole kernel -- the flat form spent two 64-bit vector additions per sub-step on addresses)
    A('#pragma unroll')
    A(f'      for (int i = 0; i < NST; ++i) if (({lines} * 64) % NTH == 0 || tid + NTH * i < {lines} * 64)')
    A('        st[i] = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, tid * 16, (s * (LPS * 64) + NTH * i) * 16, 0);')
    A('    }')
    A('  };')
    A('  auto stage_store = [&](int b) {')
    A('    if constexpr (!GLDS) {')
    A('#pragma unroll')
    A(f'      for (int i = 0; i < NST; ++i) if (({lines} * 64) % NTH == 0 || tid + NTH * i < {lines} * 64) slab[b][tid + NTH * i] = st[i];')
    A('    }')
    A('  };')


def reverse_plan(p):
    """Nonzero pattern the reverse tensor-product body walks: per x component a, per output component c,
    the spherical-harmonic components b it couples with and C[a,b,c] (incl. the sqrt(2 l3 + 1) path norm)."""
    plan: Dict[int, Dict[int, List[tuple]]] = {}
    for a, b, c, v in _path_terms(p):
        plan.setdefault(a, {}).setdefault(c, []).append((b, v))
    return [(a, sorted(cd.items())) for a, cd in sorted(plan.items())]


def reverse_body_reference(p, x, y, w, G):
    """NumPy statement of exactly the arithmetic `_emit_reverse_body` generates for one (edge, channel):
    x[2 l1 + 1], y[nsh] (all spherical harmonics of the edge), w scalar, G[2 l3 + 1] -> (g_w, g_y[nsh], g_x[2 l1 + 1]).
    Checked against the dense contraction in tests/test_codegen_fused_cpu.py."""
    import numpy as np
    gy = np.zeros(len(y))
    gx = np.zeros(2 * p.l1 + 1)
    gw = 0.0
    for a, cl in reverse_plan(p):
        wx = w * x[a]
        P = 0.0
        for c, bl in cl:
            V = sum(v * y[p.sh_off + b] for b, v in bl)      # contraction of C with the edge's harmonics: per edge, not per channel
            P += V * G[c]
            s = wx * G[c]                                    # (summed over the lane's channels in the kernel)
            for b, v in bl:
                gy[p.sh_off + b] += v * s
        gw += x[a] * P
        gx[a] += w * P
    return gw, gy, gx


def _emit_reverse_body(A, pi, p):
    """Reverse tensor product of one path for a lane's 4 channels (lane = edge).  The Clebsch-Gordan tensor is
    contracted with the edge's spherical harmonics FIRST (V_ac = sum_b C_abc Y_b: per edge, shared by the 4 channels),
    so the per-channel work is two multiply-adds per nonzero (a, c) pair instead of one per nonzero C_abc plus four
    per (a, b) pair:   P_a = sum_c V_ac G_c;  g_w = sum_a x_a P_a;  g_x_a += w P_a;
                       g_Y_b += sum_ac C_abc s_ac  with  s_ac = sum_channels (w x_a) G_c   (one dot product per (a, c)).
    SevenNet-0 middle layer: 6 340 instead of 9 584 vector instructions per 16-edge tile; lmax-3 shapes gain more."""
    d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
    A('template <bool GX>   // GX = false: the launch has no g_xe output (first / last layer): the source-row gradient is not formed at all')
    A(f'__device__ __forceinline__ void bwdf_p{pi}(const f32x4 (&xr)[{d1}], const float (&ys)[NSH], const f32x4 w,')
    A(f'    const f32x4 (&G)[{d3}], f32x4 &gw, float (&gy)[NSH], f32x4 (&gx)[{d1}]) {{')
    A('  float gw0 = 0.f, gw1 = 0.f, gw2 = 0.f, gw3 = 0.f;')
    for a, cl in reverse_plan(p):
        A(f'  {{  // x component {a}')
        for r in range(4):
            A(f'    const float wx{r} = w[{r}] * xr[{a}][{r}];')
        A('    float P0, P1, P2, P3;')
        for k, (c, bl) in enumerate(cl):
            tl = [f'{_f(v)} * ys[{p.sh_off + b}]' for b, v in bl]
            A(f'    {{ const float V = {_sum_expr(tl)};')
            for r in range(4):
                A(f'      P{r} = ' + (f'V * G[{c}][{r}];' if k == 0 else f'fmaf(V, G[{c}][{r}], P{r});'))
            A(f'      const float s = fmaf(wx3, G[{c}][3], fmaf(wx2, G[{c}][2], fmaf(wx1, G[{c}][1], wx0 * G[{c}][0])));')
            for b, v in bl:
                A(f'      gy[{p.sh_off + b}] = fmaf({_f(v)}, s, gy[{p.sh_off + b}]);')
            A('    }')
        for r in range(4):
            A(f'    gw{r} = fmaf(xr[{a}][{r}], P{r}, gw{r});')
            A(f'    if constexpr (GX) gx[{a}][{r}] = fmaf(w[{r}], P{r}, gx[{a}][{r}]);')
        A('  }')
    A('  gw = f32x4{gw0, gw1, gw2, gw3};')
    A('}')


def _emit_reverse_body_pk(A, pi, p, pkgy=True):
    """The reverse body of `_emit_reverse_body` on PACKED fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32): the lane's 4 channels are two
    register pairs, the per-edge scalars (V, the Clebsch-Gordan constants) enter as splat operands.  Same products, same
    accumulation order per channel; s_ac keeps its channels as two partial sums ((0 + 2), (1 + 3)) that dE/dY accumulates SEPARATELY
    (f32x2 gy[NSH], round 6) and the kernel's epilogue adds.  Why (round 5,
    profiles/r05_issue_rate_probe.txt): a wave issues one vector instruction per ~7.5 cycles whatever its kind, so at the two waves per
    SIMD these kernels run at, the vector pipe is capped by the issue cadence at about half its rate and the kernel's time is its
    INSTRUCTION COUNT times that cadence -- a packed instruction does two lanes' worth of multiply-adds in the same issue slot
    (measured: v_pk_fma_f32 7.0 cycles per wave-instruction against 7.5 for v_fma_f32 at one and two waves per SIMD)."""
    d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
    A('template <bool GX>')
    A(f'__device__ __forceinline__ void bwdf_p{pi}(const f32x4 (&xr)[{d1}], const float (&ys)[NSH], const f32x4 w,')
    A(f'    const f32x4 (&G)[{d3}], f32x4 &gw, {"f32x2" if pkgy else "float"} (&gy)[NSH], f32x4 (&gx)[{d1}]) {{')
    A('  const f32x2 wl = lo2(w), wh = hi2(w);')
    A('  f32x2 gwl = f32x2{0.f, 0.f}, gwh = gwl;')
    for a, cl in reverse_plan(p):
        A(f'  {{  // x component {a}')
        A(f'    const f32x2 xl = lo2(xr[{a}]), xh = hi2(xr[{a}]);')
        A('    const f32x2 wxl = wl * xl, wxh = wh * xh;')
        A('    f32x2 Pl, Ph;')
        for k, (c, bl) in enumerate(cl):
            tl = [f'{_f(v)} * ys[{p.sh_off + b}]' for b, v in bl]
            A(f'    {{ const float V = {_sum_expr(tl)};')
            A('      const f32x2 V2 = f32x2{V, V};')
            A(f'      const f32x2 Gl = lo2(G[{c}]), Gh = hi2(G[{c}]);')
            if k == 0:
                A('      Pl = V2 * Gl; Ph = V2 * Gh;')
            else:
                A('      Pl = __builtin_elementwise_fma(V2, Gl, Pl); Ph = __builtin_elementwise_fma(V2, Gh, Ph);')
            A('      const f32x2 s2 = __builtin_elementwise_fma(wxh, Gh, wxl * Gl);')
            for b, v in bl:   # (the two halves of s_ac are summed once per tile, in the epilogue: one add per (a, c) pair less; the
                #               Clebsch-Gordan constant reaches the packed FMA as an SGPR splat -- s_mov, no vector instruction)
                if pkgy:
                    A(f'      gy[{p.sh_off + b}] = __builtin_elementwise_fma(f32x2{{{_f(v)}, {_f(v)}}}, s2, gy[{p.sh_off + b}]);')
                else:
                    A(f'      gy[{p.sh_off + b}] = fmaf({_f(v)}, s2[0] + s2[1], gy[{p.sh_off + b}]);')
            A('    }')
        A('    gwl = __builtin_elementwise_fma(xl, Pl, gwl); gwh = __builtin_elementwise_fma(xh, Ph, gwh);')
        A('    if constexpr (GX) {')
        A(f'      const f32x2 gl_ = __builtin_elementwise_fma(wl, Pl, lo2(gx[{a}])), gh_ = __builtin_elementwise_fma(wh, Ph, hi2(gx[{a}]));')
        A(f'      gx[{a}] = f32x4{{gl_[0], gl_[1], gh_[0], gh_[1]}};')
        A('    }')
        A('  }')
    A('  gw = f32x4{gwl[0], gwl[1], gwh[0], gwh[1]};')
    A('}')


class _Gen:
    """One shape's generator state: the spec, its two sub-step schedules, every per-shape decision (made ONCE, here) and the
    output lines.  The emitters below are plain functions over it, in source order: header, per-path bodies, reverse kernel
    (prologue / block top / sub-step / block end / hidden-layer tail / epilogue), forward kernel (prologue / pass top / block),
    launchers."""

    def __init__(self, spec: ConvSpec):
        self.spec = spec
        tag = self.tag = spec.tag
        # default configuration of the two kernels: (waves per workgroup, direct global->LDS staging, waves per SIMD the
        # register allocation targets); SNET_CODEGEN_OPTS="fexp=<tag>" adds every combination for one shape, selected at
        # run time by SNET_FV_BWD / SNET_FV_FWD="nwv,glds,occ" (kernel tuning only)
        self.def_b = (int(OPTS.get('fnwv', 8)), int(OPTS.get('fglds', 0)), int(OPTS.get('focc', 2)))
        self.def_f = (int(OPTS.get('fnwvf', 8)), int(OPTS.get('fgldsf', 0)), int(OPTS.get('foccf', 2)))
        self.exp = OPTS.get('fexp') == tag
        self.DX, self.DOUT, self.NSH, self.WN = spec.irreps_x.dim, spec.irreps_out.dim, spec.irreps_sh.dim, spec.weight_numel
        NSH = self.NSH
        cats, self.pairs_of, self.cols = schedule(spec)
        self.cats = cats
        self.NS = len(self.cols)
        self.offs_x = spec.irreps_x.offsets()
        self.dead_x = [i for i in range(len(spec.irreps_x)) if not any(p.i_x == i for p in spec.paths)]
        self.L: List[str] = []
        self.A = self.L.append
        # row stride of the forward kernel's spherical-harmonics staging: the four edge groups of a wave read rows 4 apart,
        # which for nsh = 16 (lmax 3) all fall on one LDS bank (measured: 36 % of the kernel's LDS cycles were conflicts)
        self.NSHP = NSH + 1 if (4 * NSH) % 32 == 0 else NSH
        self.gxe_std = bool(OPTS.get('gxestd'))   # kernel-tuning builds: standard g_xe row order (round-2 layout) for A/B runs
        self.stamped = tag in (OPTS.get('stamp'), OPTS.get('stampl'), OPTS.get('stampf'), OPTS.get('stampfl'))
        # forward kernel: a tile with m <= 12 edges keeps ceil(m / 4) accumulator rows per lane group instead of filling groups in turn
        # (SNET_CODEGEN_OPTS=frow=0: the previous row order; middle layers 2.79 -> 2.73 ms same box, profiles/r05_ab_forward_variants.txt)
        self.FROW = bool(int(OPTS.get('frow', 1)))
        self.FRSB = int(OPTS.get('frsb', 0))   # scheduler fence between the rows of a forward body with >= frsb Clebsch-Gordan entries (0: none)
        self._reverse_decisions()
        self._forward_decisions()

    # ---------------------------------------------------------------- decisions of the reverse kernel
    def _reverse_decisions(self):
        spec, cats, NSH = self.spec, self.cats, self.NSH
        # The reverse kernel walks the weight columns in BLOCKS: one x block (cat) and U = 1 or 2 of its 16-channel tiles.  An x
        # block with an odd number of paths leaves one path without a partner for the second 16-column tile of a sub-step;
        # with U = 2 that path's tiles of two consecutive channel tiles share a sub-step instead (schedule_bwd): the
        # SevenNet-0 middle layer needs 30 sub-steps instead of 34, its last layer (three one-path x blocks) 7 instead of 14.
        bsched, cols_b = schedule_bwd(spec)
        self.bsched, self.cols_b = bsched, cols_b
        # the node's g_out entries a block needs, in the order its lanes fetch them: (path, output component, member tile)
        glists = [[(pi, m3, u) for u in range(bs['U']) for pi, p in bs['cat'].paths for m3 in range(2 * p.l3 + 1)] for bs in bsched]
        self.glists = glists
        NGP = max((len(gl) + 15) // 16 * 16 for gl in glists)   # rows of the LDS buffer (padded to 16 entries per load)
        NK = NGP // 16
        self.NGP, self.NK = NGP, NK
        maxd1 = max(2 * c_.l1 + 1 for c_ in cats)
        maxd3 = max(2 * p_.l3 + 1 for p_ in spec.paths)

        def live_regs(gs):
            """lower bound of the kernel's live vector registers: source rows, their prefetch and their gradient (3 x 4 d1), h2^T
            split and g_h2 accumulators (32), Y gradient, g_out prefetch (gs = 1, or 2 with packed tiles), slab staging, one path's
            g_out entries, addresses / scales"""
            return 12 * maxd1 + 32 + NSH + 4 * gs * NK + 16 + 4 * maxd3 + 35
        # Packed tiles (SNET_CODEGEN_OPTS=xtile=1): a tile is a window of <= 16 consecutive CSR edges that may run from one destination
        # node (A) into the next one that has edges (B); snet_edge_tiles_packed writes tile_ptr[t] = first edge, tile_node[2t .. 2t+1] = A, B.
        # The wave keeps BOTH nodes' g_out entries in its LDS buffer and every edge lane reads its own node's set.
        # Chosen per shape: the second node's prefetched entries cost 4 NK more registers, which the lmax-3 shapes (at 256 already) do
        # not have (round 4: 20 .. 2400 spilled registers with it; estimate 227 .. 243 against 195 for the largest shape that fits).  Nor
        # where the g_out entries are most of a block's vector-memory instructions or the shape would leave three waves per SIMD for two
        # (same box, in the step: first layer ecc5d202727d 1.69 -> 2.03 ms, last layer 005c575f8ec2 1.50 -> 1.63 ms with packed tiles; the
        # middle layers 5.64 -> 5.24 ms).  SNET_CODEGEN_OPTS=xtile=0 / 1 forces it.
        packed_class = live_regs(2) <= 200 and live_regs(1) > 168 and len(cats) > 1   # two waves per SIMD, with register headroom
        XT = self.XT = bool(int(OPTS['xtile'])) if 'xtile' in OPTS else packed_class
        GS = self.GS = 2 if XT else 1
        # Packed fp32 reverse bodies (round 5, `_emit_reverse_body_pk`): on by default for the same class -- the shapes that run two waves
        # per SIMD with register headroom (SevenNet-0 middle layers 5.04 -> 4.92 ms, same box) -- and off elsewhere: the first layer's
        # 8-wave kernel crosses 128 registers with them (127 -> 132: 1.89 -> 2.29 ms), the lmax-3 shapes at 256 registers start to
        # spill (0 -> 6, 7 -> 37).  SNET_CODEGEN_OPTS=pk=0 / 1 forces it (profiles/r05_ab_packed_fp32_bodies.txt).
        self.PK = bool(int(OPTS['pk'])) if 'pk' in OPTS else XT
        self.PKGY = self.PK and bool(int(OPTS.get('pkgy', 1)))   # dE/dY accumulated as two packed partial sums (round 6)
        # Order of the vector-memory operations (round 4).  vmcnt retires IN ORDER, and the slab fragments of the next sub-step
        # are waited for at the end of every sub-step: whatever was issued before those slab loads -- the gathers of the next
        # block's source rows and g_out entries, the g_xe stores of the block just finished -- is waited for with them.  So the
        # first sub-step of a block does not request its slab itself: the request is HOISTED in front of the previous block's
        # stores (and, for the first block, into the prologue); the long-latency operations then have two sub-steps to complete
        # instead of (at best) one.  SNET_CODEGEN_OPTS=novmord=1 restores the round-3 order.
        VMORD = self.VMORD = not OPTS.get('novmord')
        self.GRAW = VMORD and not OPTS.get('nograw')    # g_out entries loaded raw, 1/denominator applied when they are parked
        GUNC = VMORD and not OPTS.get('nogunc')         # padding entries of the g_out fetch: unconditional loads of offset 0
        live = self.live = live_regs(GS)
        self.three_waves = 2 * 8 * 2 * 1024 + 4 * (2 * NGP * 64 + NSH * 64) <= 53 * 1024 and live <= (160 if XT else 168)
        if self.three_waves:
            GUNC = GUNC and bool(OPTS.get('gunc3'))   # three-waves-per-SIMD shapes (168 registers): the unpredicated form spilled 9 there
        self.GUNC = GUNC
        # rows of the NEXT x block requested one block ahead (xp): only where the extra U d1 vector registers fit the budget of the
        # occupancy this shape runs at (the lmax-3 shapes sit at 256 already and spilled 200+ registers with it)
        budget = (168 if self.three_waves else 256) - 24   # margin: the estimate is a lower bound of what hipcc's allocator ends up with
        xp_regs = max([4 * b_['U'] * (2 * b_['cat'].l1 + 1) for b_ in bsched[1:]] or [0])
        self.XPF = VMORD and len(bsched) > 1 and live + xp_regs + 8 <= budget and not OPTS.get('noxpf')
        # the hoisted slab request keeps the staging registers live across the block boundary: same budget rule
        self.HOIST = VMORD and live + 16 + 8 <= budget and not OPTS.get('nohoist')
        # x blocks with 2 l + 1 >= noxn get no register prefetch of the next block's source rows.  Default: the l = 3 blocks of the
        # shapes that sit at 256 registers (round 4: with the prefetch those kernels spilled 15 .. 33 registers, and a spill reload
        # waits for every older gather; without it 0 .. 7, l3i5 middle layer 11.22 -> 11.02 ms).  SNET_CODEGEN_OPTS=noxn=<d1> overrides
        self.noxn = int(OPTS.get('noxn', 7 if live > 200 else 0))
        # Phase stamps (SNET_CODEGEN_OPTS=stamp=<tag>, kernel-tuning builds only): s_memtime at the phase boundaries of the reverse kernel,
        # per-phase cycle sums of every wave added to the device array snet_stamps (read back by snet_debug_stamps; tools/microbench.py
        # --stamps).  Every stamp drains the wave's LDS counter and fences the scheduler, so the instrumented kernel runs ~10 % slower
        # than the shipped one: the split between phases is what it is for.
        # stampl=<tag>: the LIGHT form -- stamps at the block-level boundaries only (prologue, block top, the block's sub-steps as one
        # phase [id 9], block end, tail, epilogue: ~25 stamps per tile instead of ~270), so the split between "inside the sub-steps" and
        # "around them" is measured on a kernel that runs close to the shipped one
        self.STL = OPTS.get('stampl') == self.tag
        self.ST = OPTS.get('stamp') == self.tag or self.STL

    def block_info(self, ci):
        """(x block, member tiles per block, blocks, 2 l1 + 1, no register prefetch of the next block's rows) of reverse block ci"""
        bs = self.bsched[ci]
        d1 = 2 * bs['cat'].l1 + 1
        return bs['cat'], bs['U'], bs['ncb'], d1, (self.noxn > 0 and d1 >= self.noxn)

    def bwd_lds(self, nt, nwv):
        return 2 * 8 * nt * 1024 + nwv * (2 * self.NGP * 64 + (128 if self.XT else 0) + self.NSH * 64)

    def bwd_cfg(self, nt):
        """(waves per workgroup, direct global->LDS staging, waves per SIMD) of the reverse kernel at nt operand terms"""
        # measured on MI355X (SevenNet-0 middle layer): three 4-wave workgroups per CU (LDS <= 53 KB each, <= 168
        # VGPRs) beat two; when the slab does not leave room for three, two 4-wave workgroups at 256 VGPRs
        if 'fnwv' in OPTS:
            return self.def_b
        # first interaction layer (scalar inputs only, one x block): one 8-wave workgroup sharing each slab beats three
        # 4-wave ones (in the step: 1.76 vs 2.01 ms).  NOT the last layer's shape (three one-path x blocks): 8 waves
        # win its stand-alone timing (3.21 vs 3.46 ms) but lose inside the step with the hidden-layer tail (4.04 vs 3.53)
        if len(self.cats) == 1 and self.bwd_lds(nt, 8) <= 80 * 1024:
            return (8, 0, 2)
        # three waves per SIMD only where the kernel's live state fits 168 VGPRs (live_regs above).  The SevenNet-0 middle layer
        # needs ~200: at 168 the compiler spilled 38 registers and the spill traffic queues with the prefetch loads (round 3, same
        # box: 6.80 ms at three waves with spills, 6.43 at two waves without)
        if self.bwd_lds(nt, 4) <= 53 * 1024 and self.live <= (160 if self.XT else 168):   # (packed tiles: 164 estimated spilled 20 at 168)
            return (4, 0, 3)
        if self.bwd_lds(nt, 4) <= 80 * 1024:
            return (4, 0, 2)
        return (2, 0, 2)

    # ---------------------------------------------------------------- decisions of the forward kernel
    def _forward_decisions(self):
        spec, cats = self.spec, self.cats
        # One wavefront = one destination node, up to two 16-edge tiles per pass.  The weight stream is consumed in
        # BLOCKS of up to FG sub-steps (all paths of one (x block, 16-channel tile) when there are <= 2 FG of them):
        # one workgroup barrier per block; inside a block the wave walks its tiles, and for each tile stages the 16
        # source rows' slice ONCE in wave-private LDS with d1 16-byte loads per lane (instead of 4 d1 4-byte gathers
        # per path pair), then runs every path of the block on it.  Output rows leave through LDS too: 16-byte
        # stores of 4 channels per lane instead of one 64-byte row segment per instruction.
        FG = int(OPTS.get('ffg', 3))
        fgroups = []   # per cat: list of groups, each a list of (sub-step index within the (cat, ct) block, (pa, pb))
        for ci, cat in enumerate(cats):
            prs = list(enumerate(self.pairs_of[ci]))
            fgroups.append([prs[i:i + FG] for i in range(0, len(prs), FG)])
        self.fgroups = fgroups
        self.LPB = max(len(grp) for gl in fgroups for grp in gl)          # sub-steps per slab block
        NOE = max(sum(2 * spec.paths[pi].l3 + 1 for _, pr in grp for pi in pr if pi is not None) for gl in fgroups for grp in gl)
        self.NOEP = (NOE + 15) // 16 * 16
        self.MAXD1 = max(2 * cat.l1 + 1 for cat in cats)
        # Output-row offsets of every (x block, path group, 16-entry chunk): one register per chunk, live for the whole kernel.  The lmax-3
        # shapes have 13 .. 25 of them (SevenNet-0: 5) beside 228 .. 256 other live registers -- MF-ompa's middle layer spilled 24 --, so
        # where there are more than 8 the table lives in LDS and a lane reads its entry at the store (round 5).
        self.N_OOFF = sum((sum(2 * spec.paths[pi].l3 + 1 for _, pr in grp for pi in pr if pi is not None) + 15) // 16 for gl in fgroups for grp in gl)
        self.OOLDS = self.N_OOFF > 8 if 'oolds' not in OPTS else bool(int(OPTS['oolds']))
        # Phase stamps of the FORWARD kernel (SNET_CODEGEN_OPTS=stampf=<tag>, stampfl=<tag> the light form: one stamp per tile instead of
        # two per path tile), same device array and read-back as the reverse kernel's; a row per destination node
        self.STFL = OPTS.get('stampfl') == self.tag
        self.STF = OPTS.get('stampf') == self.tag or self.STFL
        self.olists, self.oo_index = {}, {}

    def fwd_lds(self, nt, nwv):
        return 2 * self.LPB * 4 * nt * 1024 + nwv * (32 * self.NSHP * 4 + self.MAXD1 * 1024 + self.NOEP * 64) + 4 * nwv + (64 * self.N_OOFF if self.OOLDS else 0)

    def fwd_cfg(self, nt):
        """(waves per workgroup, direct global->LDS staging, waves per SIMD) of the forward kernel at nt operand terms"""
        # measured: occupancy decides -- one 12-wave workgroup per CU at <= 168 VGPRs (3 waves per SIMD, direct
        # global->LDS staging) where the LDS and the register count of this nt allow it, 8 or 4 waves otherwise
        if 'fnwvf' in OPTS:
            return self.def_f
        if len(self.cats) == 1 and self.fwd_lds(nt, 8) <= 80 * 1024:   # first layer (scalar inputs only): 0.87 vs 1.03 ms at 12 waves;
            return (8, 0, 2)                                             # slabs staged through registers: 0.85 vs 1.00 ms direct
        if nt <= 2 and self.fwd_lds(nt, 12) <= 160 * 1024:
            return (12, 0 if OPTS.get('f12reg') else 1, 3)
        for w in (8, 4, 2, 1):
            if self.fwd_lds(nt, w) <= 160 * 1024:
                return (w, 1, 2)
        raise NotImplementedError(f'conv shape {self.spec.key}: the fused forward kernel does not fit the LDS')

    # ---------------------------------------------------------------- small emit helpers shared by several emitters
    @staticmethod
    def out_index(p, m3):
        return p.out_off + m3 * p.out_mul + p.out_ch

    def S(self, i, ind='      '):
        """reverse-kernel phase stamp i (only in stamp builds; the light form keeps the block-level ones)"""
        if self.ST and (not self.STL or i in (0, 1, 10, 11, 12, 13)):
            self.A(f'{ind}stamp({i});')

    def SF(self, i, ind, light=True):
        if self.STF and (light or not self.STFL):
            self.A(f'{ind}stamp({i});')

    def g_row(self, ci, pi, m3, u):
        return self.glists[ci].index((pi, m3, u))

    def emit_g_loads(self, ind, ci, cb_expr):
        A, glists, bsched, exp, GRAW, GUNC, XT = self.A, self.glists, self.bsched, self.exp, self.GRAW, self.GUNC, self.XT
        gl = glists[ci]
        U_ = bsched[ci]['U']
        for k in range((len(gl) + 15) // 16):
            # (the 1/denominator factor is applied when the entries are PARKED: multiplied here, the value is needed at once and
            # the compiler answers with s_waitcnt vmcnt(0) right behind the load -- a full gather latency, and a drain of the
            # g_xe stores and the source-row prefetch in front of it, at the top of every block: round 4, `vmord`)
            dg = '(diag & 8) ? f32x4{1.f, 1.f, 1.f, 1.f} : ' if exp else ''
            sc_ = ' * scale' if not GRAW else ''
            pred = '' if GUNC else f'(goff{ci}_{k} < 0) ? f32x4{{0.f, 0.f, 0.f, 0.f}} : '
            A(f'{ind}gpre[{k}] = {dg}{pred}*reinterpret_cast<const f32x4 *>(gnode + goff{ci}_{k} + {16 * U_} * ({cb_expr})){sc_};')
        if XT:   # node B's entries: a wave-uniform branch, most tiles of a long segment have one node
            A(f'{ind}if (two) {{')
            for k in range((len(gl) + 15) // 16):
                A(f'{ind}  gpre_b[{k}] = {dg}{pred}*reinterpret_cast<const f32x4 *>(gnode_b + goff{ci}_{k} + {16 * U_} * ({cb_expr})){sc_};')
            A(f'{ind}}}')

    def emit_g_park(self, ind, ci, buf_expr):
        A, glists, GRAW, XT = self.A, self.glists, self.GRAW, self.XT
        gl = glists[ci]
        for k in range((len(gl) + 15) // 16):
            sc_ = ' * g_park'
            A(f'{ind}*reinterpret_cast<f32x4 *>(&s_g[wave][{"0" if XT else buf_expr}][(16 * {k} + (lane >> 2)) * 16 + 4 * (lane & 3)]) = gpre[{k}]{sc_};')
        if XT:
            A(f'{ind}if (two) {{')
            for k in range((len(gl) + 15) // 16):
                A(f'{ind}  *reinterpret_cast<f32x4 *>(&s_g[wave][1][(16 * {k} + (lane >> 2)) * 16 + 4 * (lane & 3)]) = gpre_b[{k}]{sc_};')
            A(f'{ind}}}')

    def emit_x_loads(self, ind, ci_, ct_expr, tl_):
        A, cats = self.A, self.cats
        cat_ = cats[ci_]
        A(f'{ind}{{ const float *xs_ = x + {cat_.x_off} + 16 * ({ct_expr}) + 4 * (lane & 3) + (size_t)srs[{tl_}] * DX;')
        for m in range(2 * cat_.l1 + 1):
            A(f'{ind}  xq[{m}] = *reinterpret_cast<const f32x4 *>(xs_ + {m * cat_.mul});')
        A(f'{ind}}}')


def _emit_header(cx: _Gen):
    A, spec, tag, cats = cx.A, cx.spec, cx.tag, cx.cats
    DX, DOUT, NSH, NSHP, WN, NS, cols, cols_rev = cx.DX, cx.DOUT, cx.NSH, cx.NSHP, cx.WN, cx.NS, cx.cols, cx.cols_b
    A('// GENERATED by sevennet_amd/codegen_fused.py -- do not edit.')
    A(f'// fused conv shape {tag}: x = {spec.irreps_x}, sh = {spec.irreps_sh}, out = {spec.irreps_out}')
    A(f'// {len(spec.paths)} paths, weight_numel = {WN}, {NS} sub-steps of two 16-column tiles')
    A('#include <cstdio>')
    A('#include <cstdlib>')
    A('#include "snet_common.h"')
    A('#include "snet_split.h"')
    A('namespace {')
    A('using namespace snet;')
    A('__device__ __forceinline__ f32x2 lo2(const f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }')
    A('__device__ __forceinline__ f32x2 hi2(const f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }')
    A(f'constexpr int DX = {DX}, DOUT = {DOUT}, NSH = {NSH}, NSHP = {NSHP}, WN = {WN}, NS = {NS};')
    A('const int32_t SUB_COLS[NS * 2] = {' + ', '.join(f'{a}, {b}' for a, b in cols) + '};')
    A(f'const int32_t SUB_COLS_B[{2 * len(cols_rev)}] = {{' + ', '.join(f'{a}, {b}' for a, b in cols_rev) + '};')
    # g_xe[E, DX] is a private intermediate (reverse kernel -> segment sum): inside a row its 16-channel chunks are kept in
    # the order the kernel produces them, [x block][channel tile][component], so that the 2 l + 1 stores of one
    # (block, tile) write one contiguous run per edge (whole 128-byte lines) instead of 64-byte halves of lines whose other
    # half arrives one channel tile later.  GXE_CHUNK[standard chunk] = chunk position inside the g_xe row.
    gxe_chunk = list(range(DX // 16))
    for cat in ([] if cx.gxe_std else cats):
        d1c = 2 * cat.l1 + 1
        for ct in range(cat.mul // 16):
            for m in range(d1c):
                gxe_chunk[(cat.x_off + m * cat.mul) // 16 + ct] = cat.x_off // 16 + ct * d1c + m
    assert sorted(gxe_chunk) == list(range(DX // 16))
    A(f'const int32_t GXE_CHUNK[{DX // 16}] = {{' + ', '.join(str(v) for v in gxe_chunk) + '};')
    A('')


def _emit_path_functions(cx: _Gen):
    """per path: the reverse body (lane = edge, 4 channels per lane) and the forward body (lane = channel, 4 edges per lane)"""
    A, spec, PK, FROW, FRSB = cx.A, cx.spec, cx.PK, cx.FROW, cx.FRSB
    for pi, p in enumerate(spec.paths):
        d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
        terms = _path_terms(p)
        # ---- reverse: lane = edge, 4 channels in the vector components
        A(f'// path {pi}: ({p.l1} x {p.l2} -> {p.l3})')
        d2 = 2 * p.l2 + 1
        byab: Dict[tuple, List[tuple]] = {}
        for a_, b_, cc, v in terms:
            byab.setdefault((a_, b_), []).append((cc, v))
        if PK:
            _emit_reverse_body_pk(A, pi, p, cx.PKGY)
        else:
            _emit_reverse_body(A, pi, p)
        # ---- forward: lane = channel, the 4 edges of the lane's group in the vector components
        A(f'__device__ __forceinline__ void fwdf_p{pi}(const float (&xr)[4][{d1}], const float *ysl, const f32x4 w,')
        A(f'    float (&acc)[{d3}], const int rows) {{')
        # rows (wave-uniform): accumulator rows per lane group that hold an edge in this tile -- a short tile is spread over the four
        # lane groups (FROW below), so that its empty row is the SAME r = 3 in every lane and its arithmetic can be skipped.  Only
        # r = 3 is conditional and its harmonics are read up front: one branch per body, the other three rows stay one basic block
        ybs = sorted({b_ for (_, b_) in byab})
        if FROW:
            for b_ in ybs:
                A(f'  const float y3_{b_} = ysl[3 * NSHP + {p.sh_off + b_}];')
        # One-to-one paths -- (0, l -> l) and (l, 0 -> l): every output component has exactly one term, all with one coefficient --
        # need no intermediate sums: t = w v x_0 (resp. w v Y_0) once per edge, then ONE multiply-add per output component
        # (round 6: 6 instead of 10 vector instructions per edge for l = 2, 4 instead of 6 for l = 1)
        one_to_one = (p.l1 == 0 or p.l2 == 0) and len(terms) == d3 and len({c_ for _, _, c_, _ in terms}) == d3 and \
            len({struct.unpack('f', struct.pack('f', v_))[0] for _, _, _, v_ in terms}) == 1   # (equal as fp32 constants)
        for r in range(4 if one_to_one else 0):
            A(f'  if ({"true" if (r < 3 or not FROW) else "rows > 3"}) {{  // edge {r} of the lane\'s group')
            v_ = struct.unpack('f', struct.pack('f', terms[0][3]))[0]
            if p.l1 == 0:
                A(f'    const float t = w[{r}] * xr[{r}][0]' + (f' * {_f(v_)};' if v_ != 1.0 else ';'))
                for a_, b_, c_, _ in terms:
                    A(f'    acc[{c_}] = fmaf(t, ' + (f'y3_{b_}' if (FROW and r == 3) else f'ysl[{r} * NSHP + {p.sh_off + b_}]') + f', acc[{c_}]);')
            else:
                b0 = terms[0][1]
                A(f'    const float t = w[{r}] * ' + (f'y3_{b0}' if (FROW and r == 3) else f'ysl[{r} * NSHP + {p.sh_off + b0}]') + (f' * {_f(v_)};' if v_ != 1.0 else ';'))
                for a_, b_, c_, _ in terms:
                    A(f'    acc[{c_}] = fmaf(t, xr[{r}][{a_}], acc[{c_}]);')
            A('  }')
        # Scalar-output paths (l, l -> 0) with one coefficient: acc += (w v) sum_a x_a Y_a -- a dot product and one multiply-add
        dot = not one_to_one and d3 == 1 and len({struct.unpack('f', struct.pack('f', v_))[0] for _, _, _, v_ in terms}) == 1
        for r in range(4 if dot else 0):
            A(f'  if ({"true" if (r < 3 or not FROW) else "rows > 3"}) {{  // edge {r} of the lane\'s group')
            yv = lambda b_: (f'y3_{b_}' if (FROW and r == 3) else f'ysl[{r} * NSHP + {p.sh_off + b_}]')  # noqa: E731
            (a0, b0, _, v_), rest = terms[0], terms[1:]
            A(f'    float u = xr[{r}][{a0}] * {yv(b0)};')
            for a_, b_, _, _ in rest:
                A(f'    u = fmaf(xr[{r}][{a_}], {yv(b_)}, u);')
            A(f'    acc[0] = fmaf(w[{r}] * {_f(v_)}, u, acc[0]);')
            A('  }')
        for r in range(0 if (one_to_one or dot) else 4):
            A(f'  if ({"true" if (r < 3 or not FROW) else "rows > 3"}) {{  // edge {r} of the lane\'s group (its spherical harmonics: wave-private LDS rows)')
            # the weight goes on whichever side is narrower: on the 2 l1 + 1 source components up front (then every Clebsch-Gordan
            # entry accumulates straight into acc) or on the 2 l3 + 1 sums at the end
            w_first = d1 < d3
            for i in range(0 if w_first else d3):
                A(f'    float s{i} = 0.f;')
            for b_ in ybs:
                if FROW and r == 3:
                    A(f'    const float y{b_} = y3_{b_};')
                else:
                    A(f'    const float y{b_} = ysl[{r} * NSHP + {p.sh_off + b_}];')
            if w_first:
                for a_ in sorted({a_ for (a_, _) in byab}):
                    A(f'    const float xw{a_} = w[{r}] * xr[{r}][{a_}];')
            for (a_, b_), cl in sorted(byab.items()):
                A(f'    {{ const float xy = ' + (f'xw{a_}' if w_first else f'xr[{r}][{a_}]') + f' * y{b_};')
                for cc, v in cl:
                    A(f'      acc[{cc}] = fmaf({_f(v)}, xy, acc[{cc}]);' if w_first else f'      s{cc} = fmaf({_f(v)}, xy, s{cc});')
                A('    }')
            for i in range(0 if w_first else d3):
                A(f'    acc[{i}] = fmaf(w[{r}], s{i}, acc[{i}]);')
            A('  }')
            if r < 3 and FRSB and len(terms) >= FRSB:
                A('  __builtin_amdgcn_sched_barrier(0);   // one row at a time: the rows\' temporaries are not interleaved (register pressure of the large paths)')
        A('}')
    A('')


def _rev_prologue(cx: _Gen):
    """kernel head, the wave's tile, the first loads (slab, g_out entries, source rows, h2, harmonics), operand scales"""
    A, spec, tag, cats, exp = cx.A, cx.spec, cx.tag, cx.cats, cx.exp
    NSH, XT, ST, GUNC, HOIST = cx.NSH, cx.XT, cx.ST, cx.GUNC, cx.HOIST
    bsched, glists, NGP, NK, NSB = cx.bsched, cx.glists, cx.NGP, cx.NK, len(cx.cols_b)
    S, emit_g_loads, emit_g_park, out_index = cx.S, cx.emit_g_loads, cx.emit_g_park, cx.out_index
    NPH = 16
    if cx.stamped:
        A('constexpr int SNET_STAMP_TILES = 1 << 18;')
        A('__device__ unsigned snet_stamps[16 * SNET_STAMP_TILES];   // one row per tile: no atomics (2.7 M atomics on 16 hot words stalled the whole chip)')
    A('template <int NT, bool F16, int NWV, bool GLDS, int OCC, bool GX>')
    A(f'__global__ __launch_bounds__(64 * NWV, OCC) void conv_bwdf_{tag}(const float *__restrict__ x, const float *__restrict__ sh,')
    A('    const float *__restrict__ dsh, const float *__restrict__ h2, const int32_t *__restrict__ w_row,')
    A('    const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ src, const int32_t *__restrict__ tile_ptr,')
    A('    const int32_t *__restrict__ tile_node, int n_tiles, const u32x4 *__restrict__ slabs, float scale,')
    A('    const float *__restrict__ g_out, float *__restrict__ g_xe, float *__restrict__ g_h2, float *__restrict__ g_vec,')
    A('    const snet::FusedTail tail, int diag) {')
    A('  // diag: always 0 in production (bit 0 also serves as the opaque branch condition around the tensor-product bodies);')
    A('  // kernel-tuning builds: 1 skip the tensor product, 2 skip the g_h2 products, 4 skip the w products, 8 skip the')
    A('  // g_out loads, 16 skip the g_xe stores -- timing decomposition, results are then garbage')
    A('  constexpr int LPS = 8 * NT, NTH = 64 * NWV, NST = (LPS * 64 + NTH - 1) / NTH, GLN = 1;  // 1-KB fragment lines per sub-step; sub-steps per slab')
    A('  __shared__ u32x4 slab[2][GLN * LPS * 64];')
    A('  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar address arithmetic)')
    A('  const int j = lane & 15, g = lane >> 4;')
    if ST:
        A(f'  unsigned ph[{NPH}];')
        A('#pragma unroll')
        A(f'  for (int i = 0; i < {NPH}; ++i) ph[i] = 0u;')
        A('  unsigned t_prev;')
        A('  { unsigned long long t0_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); t_prev = (unsigned)t0_; }')
        A('  auto stamp = [&](int i) {')
        A('    __builtin_amdgcn_sched_barrier(0);')
        A('    unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");')
        A('    const unsigned tn = (unsigned)t_;')
        A('    ph[i] += tn - t_prev;')
        A('    t_prev = tn;')
        A('    __builtin_amdgcn_sched_barrier(0);')
        A('  };')
    # Prologue order: every load is requested as soon as its address is known -- the first weight slab at once, the
    # node's g_out entries with the row pointers, the first source rows with h2 -- so the tile pays four dependent
    # memory latencies (tile -> node -> edge -> rows) instead of seven before its first matrix product.
    _emit_staging(A, 'LPS')
    A('  stage_load(0, 0);')
    A('  const int t_raw = snet::xcd_node(blockIdx.x, gridDim.x) * NWV + wave;')
    A('  const bool live = t_raw < n_tiles;')
    A('  const int t = __builtin_amdgcn_readfirstlane(live ? t_raw : n_tiles - 1);  // idle waves shadow the last tile, stores masked')
    if XT:
        A('  const int node = __builtin_amdgcn_readfirstlane(tile_node[2 * t]);       // A: the node of the tile\'s first edge')
        A('  const int node_b = __builtin_amdgcn_readfirstlane(tile_node[2 * t + 1]);  // B: the node of its last edge (= A for most tiles of a long segment)')
        A('  const bool two = node_b != node;')
    else:
        A('  const int node = __builtin_amdgcn_readfirstlane(tile_node[t]);')
    # The node's g_out entries of one (x block, 16-channel tile) are shared by the 16 edge lanes of a group.  They
    # are fetched ONCE per wave, one block ahead, by one or two 16-byte loads per lane (lane L: channels 4 (L & 3)
    # .. +3 of entry 16 k + (L >> 2); vector-memory instructions, not bytes, are what this kernel runs out of),
    # parked in a wave-private LDS buffer and read back as group-broadcast 16-byte reads.
    A(f'  constexpr int NGP = {NGP}, NK = {NK}, NSUB = {NSB};   // NSUB: sub-steps of the reverse kernel\'s weight stream')
    if XT:
        # (one buffer per node, not two per block parity: the entries of the next block are parked after this block's last read, and a
        # wave's LDS operations execute in order)
        A(f'  __shared__ __attribute__((aligned(16))) float s_g[NWV][2][NGP * 16 + {int(OPTS.get("gpad", 16))}];   // (+ 16: the two sets of a (row, group) read on disjoint banks)')
        A('  f32x4 gpre[NK], gpre_b[NK];')
        A('  const float *gnode = g_out + (size_t)node * DOUT + 4 * (lane & 3);')
        A('  const float *gnode_b = g_out + (size_t)node_b * DOUT + 4 * (lane & 3);')
    else:
        A('  __shared__ __attribute__((aligned(16))) float s_g[NWV][2][NGP * 16];')
        A('  f32x4 gpre[NK];')
        A('  const float *gnode = g_out + (size_t)((diag & 256) ? (node & 63) : node) * DOUT + 4 * (lane & 3);')
    for ci, gl in enumerate(glists):   # per-lane offsets of the entries this lane fetches (-1: none)
        for k in range((len(gl) + 15) // 16):
            # (vmord: padding entries load offset 0 unconditionally -- their LDS rows are never read -- instead of a predicated
            # load behind a zero fill, whose write-after-write hazard made the compiler wait for every pending gather)
            offs = [out_index(spec.paths[gl[q][0]], gl[q][1]) + 16 * gl[q][2] if q < len(gl) else (0 if GUNC else -1) for q in range(16 * k, 16 * k + 16)]
            A(f'  static const int32_t GOFF{ci}_{k}[16] = {{' + ', '.join(str(o) for o in offs) + '};')
            A(f'  const int goff{ci}_{k} = GOFF{ci}_{k}[lane >> 2];')
    emit_g_loads('  ', 0, '0')
    if XT:
        A('  const int e0 = tile_ptr[t];')
        A('  const int cnt = tile_ptr[t + 1] - e0;')
        A('  const int e_b = two ? row_ptr[node + 1] : 0x7fffffff;   // first edge of node B')
    else:
        A('  const int e0 = row_ptr[node] + 16 * (t - tile_ptr[node]);')
        A('  const int cnt = min(16, row_ptr[node + 1] - e0);')
    A('  const bool valid = live && j < cnt;')
    A('  const int e = e0 + min(j, cnt - 1);')
    if XT:
        A('  const int in_b = e >= e_b ? 1 : 0;')
    A('  const int s_src = src[e];')
    A('  const int wr = w_row ? w_row[e] : e;')
    A('  float x_bound = 0.f;')
    A('  if constexpr (F16) x_bound = tail.x_max[s_src] * tail.g_max[' + ('in_b ? node_b : node' if XT else 'node') + '];')
    _c0, _U0 = cats[0], bsched[0]['U']
    _d0 = 2 * _c0.l1 + 1
    A(f'  const float *xs0 = x + (size_t)s_src * DX + {_c0.x_off} + 4 * g;')
    A(f'  f32x4 xr0[{_U0}][{_d0}], xn0[{_U0}][{_d0}];')
    for u in range(_U0):
        for m in range(_d0):
            A(f'  xr0[{u}][{m}] = *reinterpret_cast<const f32x4 *>(xs0 + {16 * u + m * _c0.mul});')

    A('  // the edge\'s spherical harmonics wait in wave-private LDS ([component][edge]: conflict-free, the four channel')
    A('  // groups of an edge read the same word) and are re-read per path: 9 .. 16 fewer live registers per lane')
    A('  __shared__ float s_y[NWV][NSH * 16];')
    A('  if (g == 0) {')
    A('#pragma unroll')
    A('    for (int k = 0; k < NSH; ++k) s_y[wave][k * 16 + j] = sh[(size_t)e * NSH + k];')
    A('  }')
    A('  const float *yl = &s_y[wave][j];')
    A('  SplitN<NT> hb[2];  // h2^T as B operand: column = edge, k slots (g, 0..7) <-> hidden unit 32 q + 8 g + slot')
    A('  float w_unscale = 1.f;  // fp16 terms: h2 is scaled per tile (wave-uniform power of two), W2 per matrix on the host')
    A('  {')
    A('    float hv[2][8];')
    A('#pragma unroll')
    A('    for (int q = 0; q < 2; ++q) {')
    A('      const f32x4 lo4 = *reinterpret_cast<const f32x4 *>(h2 + (size_t)wr * 64 + 32 * q + 8 * g);')
    A('      const f32x4 hi4 = *reinterpret_cast<const f32x4 *>(h2 + (size_t)wr * 64 + 32 * q + 8 * g + 4);')
    A('#pragma unroll')
    A('      for (int i = 0; i < 4; ++i) { hv[q][i] = lo4[i]; hv[q][4 + i] = hi4[i]; }')
    A('    }')
    A('    if constexpr (F16) {')
    A('      const int kh = snet::F16_TOP - snet::bound_exp(snet::wave_max(fmaxf(snet::max8(hv[0]), snet::max8(hv[1]))));')
    A('      const float sc = snet::pow2f(kh);')
    A('#pragma unroll')
    A('      for (int q = 0; q < 2; ++q)')
    A('#pragma unroll')
    A('        for (int i = 0; i < 8; ++i) hv[q][i] *= sc;')
    A('      w_unscale = snet::pow2f(-(kh + tail.w2_exp));')
    A('    }')
    A('    hb[0] = splitn8<NT, F16>(hv[0]);')
    A('    hb[1] = splitn8<NT, F16>(hv[1]);')
    A('  }')
    A('  f32x4 ga[4];  // g_h2^T[k = 16 m + 4 g + r][edge]')
    A('#pragma unroll')
    A('  for (int m = 0; m < 4; ++m) ga[m] = f32x4{0.f, 0.f, 0.f, 0.f};')
    if cx.PKGY:   # packed bodies keep two partial sums of dE/dY per component (added in the epilogue)
        A('  f32x2 gy[NSH];')
        A('#pragma unroll')
        A('  for (int k = 0; k < NSH; ++k) gy[k] = f32x2{0.f, 0.f};')
    else:
        A('  float gy[NSH];')
        A('#pragma unroll')
        A('  for (int k = 0; k < NSH; ++k) gy[k] = 0.f;')
    # fp16 terms, g_h2 += W2 g_w^T: the operand g_w[edge, channel] = sum_abc C_abc G_c x_a Y_b of a path is bounded by
    # (sum |C|) max|G| max|x| max|Y|.  ONE power of two per tile (round 6; it was one per edge) puts the largest bound of the tile's 16
    # edges below 2^F16_TOP -- no entry can overflow fp16, entries down to 2^-17 of the tile's bound keep all 22 bits -- and, being
    # wave-uniform, it rides on the multiply that parks the node's g_out entries (g_park = scale 2^kg): the tensor-product bodies then
    # produce g_w 2^kg directly, and the eight multiplies per sub-step in front of the operand split are gone.  Likewise the matrix
    # product's own factor w_unscale is no longer applied to w: the bodies run on the raw accumulators, which leaves g_x and dE/dY
    # times 2^kg / w_unscale -- undone once per block (gx_fix at the g_xe stores) and once per tile (epilogue).  kg <= 60 keeps the
    # parked entries finite whatever the bound is (an all-zero source row would otherwise ask for 2^114).
    KC = max(sum(abs(v) for _, _, _, v in _path_terms(p)) for p in spec.paths)
    A('  float g_sc = 1.f, g_unsc = 1.f, gx_fix = 1.f;')
    A('  if constexpr (F16) {')
    A('    float ym = 0.f;')
    A('#pragma unroll')
    A('    for (int k = 0; k < NSH; ++k) ym = fmaxf(ym, fabsf(yl[k * 16]));')
    A(f'    const int kg = min(60, snet::F16_TOP - snet::bound_exp({_f(KC)} * fabsf(scale) * snet::wave_max(x_bound * ym)));')
    A('    g_sc = snet::pow2f(kg);')
    A('    g_unsc = snet::pow2f(-(kg + tail.w2_exp));')
    A('    gx_fix = w_unscale * snet::pow2f(-kg);')
    A('  }')
    A('  const float g_park = ' + ('scale * g_sc' if cx.GRAW else 'g_sc') + ';   // factor of the g_out entries when they are parked in LDS')
    emit_g_park('  ', 0, '0')
    A('  stage_store(0);')
    A('  __syncthreads();')
    if HOIST:
        A('  if (1 < NSUB' + (' && !(diag & 32)' if exp else '') + ') stage_load(1, 1);   // the first block\'s first sub-step does not request its slab itself')
    A('  int sidx = 0, buf = 0, gbuf = 0;')
    S(0, '  ')


def _rev_block_top(cx: _Gen, ci: int):
    """x block ci: its first rows, then -- per block of U channel tiles -- the requests of the NEXT block's rows and g_out entries"""
    A, exp, XT, XPF, bsched = cx.A, cx.exp, cx.XT, cx.XPF, cx.bsched
    S, emit_g_loads = cx.S, cx.emit_g_loads
    bs = bsched[ci]
    cat, U, ncb, d1, NOXN = cx.block_info(ci)   # (NOXN: this block's next rows are requested at its end, straight into xr)
    A(f'  // ---- x block {cat.i_x}: {cat.mul}x l={cat.l1}, {len(cat.paths)} paths, {U} channel tile(s) per block, {len(bs["steps"])} sub-steps per block')
    # source rows: the next block's slices are requested one block ahead (gather latency ~1-2 us)
    if ci > 0:   # (the first x block's rows were requested in the prologue)
        A(f'  const float *xs{ci} = x + (size_t)s_src * DX + {cat.x_off} + 4 * g;')
        A(f'  f32x4 xr{ci}[{U}][{d1}], xn{ci}[{U}][{d1}];')
        for u in range(U):   # (XPF: requested at the top of the previous x block's last block, one block ahead like the others)
            for m in range(d1):
                A(f'  xr{ci}[{u}][{m}] = ' + (f'xp{ci}[{u}][{m}];' if XPF else f'*reinterpret_cast<const f32x4 *>(xs{ci} + {16 * u + m * cat.mul});'))
    if XPF and ci + 1 < len(bsched):   # the next x block's first rows are requested from inside this one: declared here
        cat_n, U_n = bsched[ci + 1]['cat'], bsched[ci + 1]['U']
        d1_n = 2 * cat_n.l1 + 1
        A(f'  const float *xs{ci + 1}p = x + (size_t)s_src * DX + {cat_n.x_off} + 4 * g;')
        A(f'  f32x4 xp{ci + 1}[{U_n}][{d1_n}];')
    A(f'  for (int cb = 0; cb < {ncb}; ++cb) {{')
    A(f'    f32x4 (&xr)[{U}][{d1}] = xr{ci};')
    A(f'    f32x4 gx[{U}][{d1}];')
    for u in range(U):
        for m in range(d1):
            A(f'    gx[{u}][{m}] = f32x4{{0.f, 0.f, 0.f, 0.f}};')
    def emit_next_xblock_rows(ind):
        for u in range(U_n):
            for m in range(d1_n):
                A(f'{ind}xp{ci + 1}[{u}][{m}] = *reinterpret_cast<const f32x4 *>(xs{ci + 1}p + {16 * u + m * cat_n.mul});')
    if ncb > 1:
        A(f'    if (cb + 1 < {ncb}' + (' && !(diag & 64)' if exp else '') + ') {')
        for u in range(U):
            for m in range(d1):
                if not NOXN:
                    A(f'      xn{ci}[{u}][{m}] = *reinterpret_cast<const f32x4 *>(xs{ci} + {16 * U} * (cb + 1) + {16 * u + m * cat.mul});')
        if XPF and ci + 1 < len(bsched):
            A('    } else {')
            emit_next_xblock_rows('      ')
        A('    }')
    elif XPF and ci + 1 < len(bsched):
        emit_next_xblock_rows('    ')
    A('    const float *gl_ = &s_g[wave][' + ('in_b' if XT else 'gbuf') + '][4 * g];')
    # next block's g_out entries: this x block's next block, or the first block of the next x block
    if ncb > 1:
        A(f'    if (cb + 1 < {ncb}) {{')
        emit_g_loads('      ', ci, 'cb + 1')
        A('    }' + (' else {' if ci + 1 < len(bsched) else ''))
        if ci + 1 < len(bsched):
            emit_g_loads('      ', ci + 1, '0')
            A('    }')
    elif ci + 1 < len(bsched):
        emit_g_loads('    ', ci + 1, '0')
    S(1, '    ')


def _rev_substep(cx: _Gen, ci: int, si_: int, ta, tb):
    """one sub-step = two 16-column tiles of W2: per tile  w = W2^T h2^T (matrix cores) -> reverse tensor-product body; then the
    operand split of the two g_w tiles and  g_h2^T += W2 g_w^T;  the next sub-step's slab is parked, one workgroup barrier"""
    A, spec, exp, ST, STL, HOIST = cx.A, cx.spec, cx.exp, cx.ST, cx.STL, cx.HOIST
    S, g_row = cx.S, cx.g_row
    U = cx.bsched[ci]['U']
    A('    {')
    if not (HOIST and si_ == 0):   # (first sub-step of a block: requested at the end of the previous block / the prologue)
        A('      if (sidx + 1 < NSUB' + (' && !(diag & 32)' if exp else '') + ') stage_load(sidx + 1, (buf ^ 1));')
    # hipcc's machine scheduler, left alone, sinks the prefetch loads next to their use (zero overlap) and
    # interleaves the phases until ~200 VGPRs spill: pin the prefetch at the top and fence the phases
    A('      __builtin_amdgcn_sched_barrier(0);')
    S(2)
    A('      const u32x4 *sl = slab[buf];')
    A('      f32x4 gw0 = f32x4{0.f, 0.f, 0.f, 0.f}, gw1 = gw0;')
    for tp, tl_ in enumerate((ta, tb)):
        if tl_ is None:
            continue
        pi, u = tl_
        p = spec.paths[pi]
        d3 = 2 * p.l3 + 1
        A(f'      {{  // tile {tp}: path {pi}, channel tile {U} cb + {u}')
        A('        f32x4 wv = f32x4{0.f, 0.f, 0.f, 0.f};')
        if exp:
            A('        if (diag & 4) wv = f32x4{yl[0], yl[16], yl[32], yl[48]}; else')
        A('#pragma unroll')
        A('        for (int q = 0; q < 2; ++q) {')
        A('          bf16x8 a[NT];')
        A('#pragma unroll')
        A(f'          for (int tm = 0; tm < NT; ++tm) a[tm] = as_bf16x8(sl[(({tp} * 2 + q) * NT + tm) * 64 + lane]);')
        A('          wv = mfma16_split<NT, F16>(a, hb[q], wv);')
        A('        }')
        if ST and not STL:
            A('        asm volatile("" :: "v"(wv[0]), "v"(wv[3]));')
        S(3 + 2 * tp, '        ')
        A(f'        f32x4 G[{d3}];')
        for m3 in range(d3):
            A(f'        G[{m3}] = *reinterpret_cast<const f32x4 *>(gl_ + {g_row(ci, pi, m3, u)} * 16);')
        A('        float ys[NSH];')
        for b_ in range(2 * p.l2 + 1):
            A(f'        ys[{p.sh_off + b_}] = yl[{(p.sh_off + b_) * 16}];')
        # The tensor-product body sits in its own (always taken) branch on an opaque kernel argument: as
        # straight-line code hipcc's scheduler interleaves it with the surrounding matrix products and
        # loads until ~200 VGPRs spill to scratch (measured: 692 spilled registers without the branch, 0 with).
        # (Round 6 tried two guards that end in a trap instead -- block boundaries without an else path, hence without the
        # never-executed zero fills and the register copies where the two paths meet: 288 spilled registers in the SevenNet-0
        # middle layer, 1 482 .. 4 295 in the lmax-3 shapes; IR-level code motion crosses a guard it knows to be cold.)
        A(f'        gw{tp} = f32x4{{0.f, 0.f, 0.f, 0.f}};')
        A(f'        if (!(diag & 1)) bwdf_p{pi}<GX>(xr[{u}], ys, wv, G, gw{tp}, gy, gx[{u}]);')
        if ST and not STL:
            A(f'        asm volatile("" :: "v"(gw{tp}[0]), "v"(gw{tp}[3]));')
        S(4 + 2 * tp, '        ')
        A('      }')
    A('      float v[8] = {gw0[0], gw0[1], gw0[2], gw0[3], gw1[0], gw1[1], gw1[2], gw1[3]};')
    A('      const SplitN<NT> b = splitn8<NT, F16>(v);')
    if exp:
        A('      if (diag & 2) ga[0] += f32x4{v[0], v[1], v[4], v[5]}; else')
    A('#pragma unroll')
    A('      for (int m = 0; m < 4; ++m) {')
    A('        bf16x8 a[NT];')
    A('#pragma unroll')
    A('        for (int tm = 0; tm < NT; ++tm) a[tm] = as_bf16x8(sl[(4 * NT + m * NT + tm) * 64 + lane]);')
    A('        ga[m] = mfma16_split<NT, F16>(a, b, ga[m]);')
    A('      }')
    if ST and not STL:
        A('      asm volatile("" :: "v"(ga[0][0]), "v"(ga[1][0]), "v"(ga[2][0]), "v"(ga[3][0]));')
    S(7)
    A('      if (sidx + 1 < NSUB' + (' && !(diag & 32)' if exp else '') + ') stage_store((buf ^ 1));')
    S(8)
    A(('      if (!(diag & 128)) ' if exp else '      ') + '__syncthreads();')
    S(9)
    A('      buf = (buf ^ 1);')
    A('      ++sidx;')
    A('    }')


def _rev_block_end(cx: _Gen, ci: int):
    """g_xe stores of the block (behind the hoisted slab request), the next block's g_out entries parked, rows rotated"""
    A, exp, STL, HOIST, bsched, gxe_std = cx.A, cx.exp, cx.STL, cx.HOIST, cx.bsched, cx.gxe_std
    S, emit_g_park = cx.S, cx.emit_g_park
    cat, U, ncb, d1, NOXN = cx.block_info(ci)
    if STL:
        A('    stamp(9);   // (light stamps: all sub-steps of the block)')
    if HOIST:   # the next block's first sub-step: its slab request goes out BEFORE this block's stores
        A('    if (sidx + 1 < NSUB' + (' && !(diag & 32)' if exp else '') + ') stage_load(sidx + 1, (buf ^ 1));')
        A('    __builtin_amdgcn_sched_barrier(0);')
    A('    if (GX && g_xe && valid' + (' && !(diag & 16)' if exp else '') + ') {')
    A('      if constexpr (F16) {   // the bodies ran on raw matrix accumulators and scaled g_out entries: back to g_x itself')
    A('#pragma unroll')
    A(f'        for (int u = 0; u < {U}; ++u)')
    A('#pragma unroll')
    A(f'          for (int m = 0; m < {d1}; ++m) gx[u][m] *= gx_fix;')
    A('      }')
    for u in range(U):
        if gxe_std:
            A(f'      {"float *" if u == 0 else ""}o = g_xe + (size_t)e * DX + {cat.x_off} + 16 * ({U} * cb + {u}) + 4 * g;')
        else:
            A(f'      {"float *" if u == 0 else ""}o = g_xe + (size_t)e * DX + {cat.x_off} + {16 * d1} * ({U} * cb + {u}) + 4 * g;   // chunk order [tile][component]: GXE_CHUNK')
        for m in range(d1):
            # streaming stores (kept in L2, partially written lines were evicted before their other half arrived:
            # 8.9 GB written for 5.9 GB of output in round 2)
            A(f'      __builtin_nontemporal_store(gx[{u}][{m}], reinterpret_cast<f32x4 *>(o + {m * cat.mul if gxe_std else 16 * m}));')
    A('    }')
    # park the prefetched entries of the next block in the other buffer (last read one block ago)
    if ncb > 1:
        A(f'    if (cb + 1 < {ncb}) {{')
        emit_g_park('      ', ci, 'gbuf ^ 1')
        A('    }' + (' else {' if ci + 1 < len(bsched) else ''))
        if ci + 1 < len(bsched):
            emit_g_park('      ', ci + 1, 'gbuf ^ 1')
            A('    }')
    elif ci + 1 < len(bsched):
        emit_g_park('    ', ci + 1, 'gbuf ^ 1')
    A('    gbuf ^= 1;')
    A('    __builtin_amdgcn_wave_barrier();')
    if ncb > 1 and NOXN:
        A(f'    if (cb + 1 < {ncb}) {{')
        for u in range(U):
            for m in range(d1):
                A(f'      xr{ci}[{u}][{m}] = *reinterpret_cast<const f32x4 *>(xs{ci} + {16 * U} * (cb + 1) + {16 * u + m * cat.mul});')
        A('    }')
    elif ncb > 1:
        A('#pragma unroll')
        A(f'    for (int u = 0; u < {U}; ++u)')
        A('#pragma unroll')
        A(f'      for (int m = 0; m < {d1}; ++m) xr{ci}[u][m] = xn{ci}[u][m];')
    S(10, '    ')
    A('  }')


def _rev_tail(cx: _Gen):
    """after the last block: g_h2 back to scale, then either stored or -- FusedTail -- the radial MLP's hidden layers reversed on it"""
    A, spec, dead_x, offs_x, S = cx.A, cx.spec, cx.dead_x, cx.offs_x, cx.S
    if dead_x:
        A('  if (GX && g_xe && valid) {  // x blocks that feed no path get a zero gradient')
        for i in dead_x:
            mul_d, l_d, _ = spec.irreps_x[i]
            A(f'    for (int q = 4 * g; q < {mul_d * (2 * l_d + 1)}; q += 16)')
            A(f'      *reinterpret_cast<f32x4 *>(g_xe + (size_t)e * DX + {offs_x[i]} + q) = f32x4{{0.f, 0.f, 0.f, 0.f}};')
        A('  }')
    A('  if constexpr (F16) {  // back to g_h2 itself')
    A('#pragma unroll')
    A('    for (int m = 0; m < 4; ++m) ga[m] *= g_unsc;')
    A('  }')
    A('  if (tail.g_emb == nullptr) {')
    A('    if (valid) {')
    A('#pragma unroll')
    A('      for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4 *>(g_h2 + (size_t)e * 64 + 16 * m + 4 * g) = ga[m];')
    A('    }')
    A('  } else {')
    A('    // The radial MLP\'s two hidden layers, reversed on this tile\'s 16 edges: g_h2 stays in the accumulators it was')
    A('    // summed in.  Every product keeps the transposed layout [hidden unit 16 m + 4 g + r][edge j]; as a B operand')
    A('    // the k slot (g, t) of k-step s then names unit 16 (2 s + (t >> 2)) + 4 g + (t & 3) -- this lane\'s own registers')
    A('    // [2 s + (t >> 2)][t & 3] -- and the host packed the weight fragments (NT terms each) in that k order.  The')
    A('    // fragments pass through the (now idle) slab buffers in two phases, staged by the whole workgroup: per-wave')
    A('    // global fragment loads would add ~20 % to the kernel\'s vector-memory instruction count.')
    A('    constexpr int TA = 12 * NT, TB = 10 * NT;  // 1-KB lines: z1 (4) + z2 (8) fragments | g_a1 (8) + g_emb (2)')
    A('    static_assert(TA <= 2 * LPS, "tail phase does not fit the slab buffers");')
    A('    constexpr int NSA = (TA * 64 + NTH - 1) / NTH, NSB = (TB * 64 + NTH - 1) / NTH;')
    A('    u32x4 *tl = &slab[0][0];')
    A('    const u32x4 *ep = slabs + (size_t)NSUB * LPS * 64;')
    A('    u32x4 sb[NSB];')
    A('#pragma unroll')
    A('    for (int i = 0; i < NSA; ++i)')
    A('      if ((TA * 64) % NTH == 0 || tid + NTH * i < TA * 64) tl[tid + NTH * i] = ep[tid + NTH * i];')
    A('#pragma unroll')
    A('    for (int i = 0; i < NSB; ++i)  // phase-2 fragments wait in registers while phase 1 computes')
    A('      if ((TB * 64) % NTH == 0 || tid + NTH * i < TB * 64) sb[i] = ep[TA * 64 + tid + NTH * i];')
    A('    auto frag = [&](int f, bf16x8 (&a)[NT]) {')
    A('#pragma unroll')
    A('      for (int tm = 0; tm < NT; ++tm) a[tm] = as_bf16x8(tl[(f * NT + tm) * 64 + lane]);')
    A('    };')
    A('    const int nb = tail.nb, act = tail.act;')
    A('    const float cst = tail.cst;')
    A('    // fp16 terms: every dynamic operand of the tail is scaled by a wave-uniform power of two (its largest magnitude')
    A('    // in the tile lands below 2^F16_TOP), the result is multiplied by the inverse together with the weight matrix\'s')
    A('    auto scale8 = [&](float (&q)[8]) -> int {')
    A('      const int k = snet::F16_TOP - snet::bound_exp(snet::wave_max(snet::max8(q)));')
    A('      const float sc = snet::pow2f(k);')
    A('#pragma unroll')
    A('      for (int i = 0; i < 8; ++i) q[i] *= sc;')
    A('      return k;')
    A('    };')
    A('    auto scale16 = [&](f32x4 (&q)[4]) -> int {')
    A('      float m = 0.f;')
    A('#pragma unroll')
    A('      for (int i = 0; i < 4; ++i)')
    A('#pragma unroll')
    A('        for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(q[i][r]));')
    A('      const int k = snet::F16_TOP - snet::bound_exp(snet::wave_max(m));')
    A('      const float sc = snet::pow2f(k);')
    A('#pragma unroll')
    A('      for (int i = 0; i < 4; ++i) q[i] *= sc;')
    A('      return k;')
    A('    };')
    A('    auto unscale16 = [&](f32x4 (&q)[4], int k) {')
    A('      const float us = snet::pow2f(-k);')
    A('#pragma unroll')
    A('      for (int i = 0; i < 4; ++i) q[i] *= us;')
    A('    };')
    A('    float ev[8];')
    A('#pragma unroll')
    A('    for (int t = 0; t < 8; t += 4) {')
    A('      f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};')
    A('      if (8 * g + t < nb) q = *reinterpret_cast<const f32x4 *>(tail.emb + (size_t)e * nb + 8 * g + t);')
    A('      ev[t] = q[0]; ev[t + 1] = q[1]; ev[t + 2] = q[2]; ev[t + 3] = q[3];')
    A('    }')
    A('    int k_e = 0;')
    A('    if constexpr (F16) k_e = scale8(ev);')
    A('    const SplitN<NT> eb = splitn8<NT, F16>(ev);')
    A('    __syncthreads();')
    S(11, '    ')
    A('    f32x4 z1[4], z2[4];')
    A('#pragma unroll')
    A('    for (int m = 0; m < 4; ++m) {')
    A('      bf16x8 a[NT];')
    A('      frag(m, a);')
    A('      z1[m] = mfma16_split<NT, F16>(a, eb, f32x4{0.f, 0.f, 0.f, 0.f});')
    A('      z2[m] = f32x4{0.f, 0.f, 0.f, 0.f};')
    A('    }')
    A('    if constexpr (F16) unscale16(z1, k_e + tail.w0_exp);')
    A('    f32x4 a1[4], d1[4];  // act(z1) cst and cst act\'(z1): one sigmoid for both')
    A('#pragma unroll')
    A('    for (int m = 0; m < 4; ++m)')
    A('#pragma unroll')
    A('      for (int r = 0; r < 4; ++r) {')
    A('        float f_, g_;')
    A('        snet::act_both_fast(z1[m][r], act, f_, g_);')
    A('        a1[m][r] = f_ * cst;')
    A('        d1[m][r] = g_ * cst;')
    A('      }')
    A('    int k_a = 0;')
    A('    if constexpr (F16) k_a = scale16(a1);')
    A('#pragma unroll')
    A('    for (int s = 0; s < 2; ++s) {')
    A('      float v[8];')
    A('#pragma unroll')
    A('      for (int t = 0; t < 8; ++t) v[t] = a1[2 * s + (t >> 2)][t & 3];')
    A('      const SplitN<NT> b = splitn8<NT, F16>(v);')
    A('#pragma unroll')
    A('      for (int m = 0; m < 4; ++m) {')
    A('        bf16x8 a[NT];')
    A('        frag(4 + 2 * m + s, a);')
    A('        z2[m] = mfma16_split<NT, F16>(a, b, z2[m]);')
    A('      }')
    A('    }')
    A('    if constexpr (F16) unscale16(z2, k_a + tail.w1_exp);')
    A('    __syncthreads();  // every wave is done with the phase-1 fragments')
    A('#pragma unroll')
    A('    for (int i = 0; i < NSB; ++i)')
    A('      if ((TB * 64) % NTH == 0 || tid + NTH * i < TB * 64) tl[tid + NTH * i] = sb[i];')
    A('    __syncthreads();')
    A('    f32x4 ga1[4], gz[4];  // gz = dE/dz2 = g_h2 cst act\'(z2)')
    A('#pragma unroll')
    A('    for (int m = 0; m < 4; ++m) {')
    A('      ga1[m] = f32x4{0.f, 0.f, 0.f, 0.f};')
    A('#pragma unroll')
    A('      for (int r = 0; r < 4; ++r) {')
    A('        float f_, g_;')
    A('        snet::act_both_fast(z2[m][r], act, f_, g_);')
    A('        gz[m][r] = ga[m][r] * cst * g_;')
    A('      }')
    A('    }')
    A('    int k_z = 0;')
    A('    if constexpr (F16) k_z = scale16(gz);')
    A('#pragma unroll')
    A('    for (int s = 0; s < 2; ++s) {')
    A('      float v[8];')
    A('#pragma unroll')
    A('      for (int t = 0; t < 8; ++t) v[t] = gz[2 * s + (t >> 2)][t & 3];')
    A('      const SplitN<NT> b = splitn8<NT, F16>(v);')
    A('#pragma unroll')
    A('      for (int m = 0; m < 4; ++m) {')
    A('        bf16x8 a[NT];')
    A('        frag(2 * m + s, a);')
    A('        ga1[m] = mfma16_split<NT, F16>(a, b, ga1[m]);')
    A('      }')
    A('    }')
    A('#pragma unroll')
    A('    for (int m = 0; m < 4; ++m) ga1[m] *= d1[m];  // dE/dz1 (fp16 terms: still times 2^(k_z + w1_exp))')
    A('    int k_d = 0;')
    A('    if constexpr (F16) k_d = scale16(ga1);')
    A('    f32x4 ge = f32x4{0.f, 0.f, 0.f, 0.f};')
    A('#pragma unroll')
    A('    for (int s = 0; s < 2; ++s) {')
    A('      float v[8];')
    A('#pragma unroll')
    A('      for (int t = 0; t < 8; ++t) v[t] = ga1[2 * s + (t >> 2)][t & 3];')
    A('      const SplitN<NT> b = splitn8<NT, F16>(v);')
    A('      bf16x8 a[NT];')
    A('      frag(8 + s, a);')
    A('      ge = mfma16_split<NT, F16>(a, b, ge);')
    A('    }')
    A('    if constexpr (F16) ge *= snet::pow2f(-(k_z + tail.w1_exp + k_d + tail.w0_exp));')
    A('    if (valid && 4 * g < nb) {  // accumulator rows 4 g + r = basis index')
    A('      f32x4 *o = reinterpret_cast<f32x4 *>(tail.g_emb + (size_t)e * nb + 4 * g);')
    A('      *o = *o + ge;')
    A('    }')
    S(12, '    ')
    A('  }')


def _rev_epilogue(cx: _Gen):
    """dE/dY summed over the four channel groups of an edge, folded with dY/dr into g_vec (or written as g_sh for the plug-in)"""
    A, ST, S, NPH = cx.A, cx.ST, cx.S, 16
    A('  // d/d(edge_vec) = sum_i gy_i dY_i/dr (Y_0 is constant).  The 4 channel groups of an edge (lanes j, j+16, j+32,')
    A('  // j+48) are summed with two permlane swaps per value; only the g == 0 lanes then touch dsh and g_vec.')
    if cx.PKGY:
        A('  float gys[NSH];   // the packed bodies\' two partial sums per component')
        A('#pragma unroll')
        A('  for (int i = 0; i < NSH; ++i) gys[i] = gy[i][0] + gy[i][1];')
    else:
        A('  float (&gys)[NSH] = gy;')
    A('#pragma unroll')
    A('  for (int i = 1; i < NSH; ++i) {')
    A('    gys[i] = snet::swap_add16(gys[i], gys[i]);')
    A('    gys[i] = snet::swap_add32(gys[i], gys[i]);')
    A('  }')
    A('  if (tail.g_sh != nullptr) {   // b1 plug-in: the gradient with respect to the harmonics themselves (wave-uniform branch)')
    A('    gys[0] = snet::swap_add16(gys[0], gys[0]);')
    A('    gys[0] = snet::swap_add32(gys[0], gys[0]);')
    A('    if (valid && g == 0) {')
    A('#pragma unroll')
    A('      for (int i = 0; i < NSH; ++i) tail.g_sh[(size_t)e * NSH + i] = gys[i] * gx_fix;')
    A('    }')
    A('  }')
    A('  if (dsh != nullptr && valid && g == 0) {')
    A('    float t0 = 0.f, t1 = 0.f, t2 = 0.f;')
    A('    const float *jd = dsh + (size_t)e * (3 * NSH);')
    A('#pragma unroll')
    A('    for (int i = 1; i < NSH; ++i) {')
    A('      t0 = fmaf(gys[i], jd[3 * i], t0);')
    A('      t1 = fmaf(gys[i], jd[3 * i + 1], t1);')
    A('      t2 = fmaf(gys[i], jd[3 * i + 2], t2);')
    A('    }')
    A('    float *o = g_vec + (size_t)e * 3;')
    A('    o[0] += t0 * gx_fix; o[1] += t1 * gx_fix; o[2] += t2 * gx_fix;   // (gx_fix: the bodies\' common factor, see the prologue)')
    A('  }')
    if ST:
        S(13, '  ')
        A('  if (lane == 0 && live && t_raw < SNET_STAMP_TILES) {')
        A(f'    for (int i = 0; i < {NPH}; ++i) snet_stamps[t_raw * 16 + i] = ph[i];')
        A('  }')
    A('}')
    A('')


def _emit_reverse_kernel(cx: _Gen):
    _rev_prologue(cx)
    for ci, bs in enumerate(cx.bsched):
        _rev_block_top(cx, ci)
        for si_, (ta, tb) in enumerate(bs['steps']):
            _rev_substep(cx, ci, si_, ta, tb)
        _rev_block_end(cx, ci)
    _rev_tail(cx)
    _rev_epilogue(cx)


def _fwd_prologue(cx: _Gen):
    """kernel head, LDS buffers, the wave's node and pass count, slab staging lambdas, output-offset tables"""
    A, L, spec, tag, NSHP = cx.A, cx.L, cx.spec, cx.tag, cx.NSHP
    fgroups, LPB, NOEP, MAXD1, N_OOFF, OOLDS, STF = cx.fgroups, cx.LPB, cx.NOEP, cx.MAXD1, cx.N_OOFF, cx.OOLDS, cx.STF
    SF, out_index, olists, oo_index = cx.SF, cx.out_index, cx.olists, cx.oo_index
    oo_rows = []
    A('template <int NT, bool F16, int NWV, bool GLDS, int OCC>')
    A(f'__global__ __launch_bounds__(64 * NWV, OCC) void conv_fwdf_{tag}(const float *__restrict__ x, const float *__restrict__ sh,')
    A('    const float *__restrict__ h2, const int32_t *__restrict__ w_row, const int32_t *__restrict__ row_ptr,')
    A('    const int32_t *__restrict__ src, int n_nodes, const u32x4 *__restrict__ slabs, float scale, float *__restrict__ out, int w2_exp, int diag) {')
    A(f'  constexpr int LPS = 8 * NT, LPF = 4 * NT, LPB = {LPB}, NTH = 64 * NWV;  // lines per sub-step (all / w part), sub-steps per block')
    A('  constexpr int NSTB = (LPB * LPF * 64 + NTH - 1) / NTH;')
    A('  __shared__ u32x4 slab[2][LPB * LPF * 64];')
    A('  __shared__ int s_pass[NWV];')
    if OOLDS:
        A(f'  __shared__ int32_t s_ooff[{16 * N_OOFF}];')
    A('  __shared__ __attribute__((aligned(16))) float s_ys[NWV][32 * NSHP];  // spherical harmonics of the pass\'s edges (rows padded: NSHP)')
    A(f'  __shared__ __attribute__((aligned(16))) float s_x[NWV][{MAXD1} * 256];  // source-row slice of one tile: [m][r][g][channel]')
    A(f'  __shared__ __attribute__((aligned(16))) float s_o[NWV][{NOEP} * 16];      // output rows of one block: [entry][channel]')
    A('  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar address arithmetic)')
    A('  const int c = lane & 15, g = lane >> 4;')
    A('  const int n_raw = snet::xcd_node(blockIdx.x, gridDim.x) * NWV + wave;')
    if STF:
        A('  unsigned ph[16];')
        A('#pragma unroll')
        A('  for (int i = 0; i < 16; ++i) ph[i] = 0u;')
        A('  unsigned t_prev;')
        A('  { unsigned long long t0_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); t_prev = (unsigned)t0_; }')
        A('  auto stamp = [&](int i) {')
        A('    __builtin_amdgcn_sched_barrier(0);')
        A('    unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");')
        A('    const unsigned tn = (unsigned)t_;')
        A('    ph[i] += tn - t_prev;')
        A('    t_prev = tn;')
        A('    __builtin_amdgcn_sched_barrier(0);')
        A('  };')
    A('  const bool live = n_raw < n_nodes;')
    A('  const int node = __builtin_amdgcn_readfirstlane(live ? n_raw : n_nodes - 1);')
    A('  const int e_beg = row_ptr[node], e_end = row_ptr[node + 1];')
    A('  const int my_pass = live ? (e_end - e_beg + 31) >> 5 : 0;')
    oo_fill_at = len(L)
    A('  if (lane == 0) s_pass[wave] = my_pass;')
    A('  __syncthreads();')
    A('  int n_pass = 1;  // every wave of the block walks the weight stream the same number of times')
    A('#pragma unroll')
    A('  for (int i = 0; i < NWV; ++i) n_pass = max(n_pass, s_pass[i]);')
    SF(0, '  ')
    # block staging: sub-steps s0 .. s0 + n - 1, only their w parts (LPF lines each)
    A('  u32x4 st[GLDS ? 1 : NSTB];')
    A('  auto stage_load = [&](int s0, int n, int b) {  // the w parts of sub-steps s0 .. s0 + n - 1 -> slab[b]')
    A('    if constexpr (GLDS) {')
    A('      for (int l = wave; l < n * LPF; l += NWV)')
    A('        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(slabs + (size_t)(s0 + l / LPF) * (LPS * 64) + (l % LPF) * 64 + lane),')
    A('                                         (__attribute__((address_space(3))) void *)(&slab[b][l * 64]), 16, 0, 0);')
    A('    } else {')
    A('#pragma unroll')
    A('      for (int i = 0; i < NSTB; ++i) {')
    A('        const int f = tid + NTH * i, l = f >> 6;')
    A('        if (l < n * LPF) st[i] = slabs[(size_t)(s0 + l / LPF) * (LPS * 64) + (l % LPF) * 64 + (f & 63)];')
    A('      }')
    A('    }')
    A('  };')
    A('  auto stage_store = [&](int n, int b) {')
    A('    if constexpr (!GLDS) {')
    A('#pragma unroll')
    A('      for (int i = 0; i < NSTB; ++i) if (((tid + NTH * i) >> 6) < n * LPF) slab[b][tid + NTH * i] = st[i];')
    A('    }')
    A('  };')
    A('  float *onode = out + (size_t)node * DOUT + 4 * (lane & 3);')
    A('  const bool has_e = e_end > e_beg;')
    A('  const int e_last = max(e_beg, e_end - 1);  // clamp target for padded rows (their weight is zeroed)')
    # output offsets per (cat, group): entry = 16 k + (lane >> 2)
    for ci, gl in enumerate(fgroups):
        for gi, grp in enumerate(gl):
            ol = [(pi, m3) for _, pr in grp for pi in pr if pi is not None for m3 in range(2 * spec.paths[pi].l3 + 1)]
            olists[(ci, gi)] = ol
            for k in range((len(ol) + 15) // 16):
                offs = [out_index(spec.paths[ol[q][0]], ol[q][1]) if q < len(ol) else -1 for q in range(16 * k, 16 * k + 16)]
                if OOLDS:
                    oo_index[(ci, gi, k)] = len(oo_rows)
                    oo_rows.append(offs)
                else:
                    A(f'  static const int32_t OOFF{ci}_{gi}_{k}[16] = {{' + ', '.join(str(o) for o in offs) + '};')
                    A(f'  const int ooff{ci}_{gi}_{k} = OOFF{ci}_{gi}_{k}[lane >> 2];')
    if OOLDS:
        assert len(oo_rows) == N_OOFF
        L[oo_fill_at] = (f'  static const int32_t OOFF_ALL[{16 * N_OOFF}] = {{' + ', '.join(str(o) for row in oo_rows for o in row) + '};\n'
                         + f'  for (int i = tid; i < {16 * N_OOFF}; i += NTH) s_ooff[i] = OOFF_ALL[i];   // (ordered by the s_pass barrier below)\n' + L[oo_fill_at])
    # flat block schedule: first sub-step index and size of every block, in stream order


def _fwd_pass_top(cx: _Gen):
    """a pass = up to 32 edges of the node: h2 fragments and scales per tile, harmonics to LDS, first slab, first source-row slice"""
    A, FROW, fgroups, MAXD1, SF = cx.A, cx.FROW, cx.fgroups, cx.MAXD1, cx.SF
    A('  for (int pass = 0; pass < n_pass; ++pass) {')
    A('    const int eb = e_beg + 32 * pass;')
    A('    const int n_e = live ? max(0, min(32, e_end - eb)) : 0;  // edges of this pass: tile 0 = [0,16), tile 1 = [16,32)')
    A('    const bool two = n_e > 16;')
    if FROW:
        # Row r of lane group g of a tile holds the tile's edge g R + r, R = ceil(m / 4) for a tile of m edges (R = 4: the identity).
        # A diamond-cubic atom has 28 neighbours: its second tile holds 12 edges in rows {0, 1, 2} of every group, and the
        # tensor-product bodies skip row 3 in all 64 lanes (one eighth of the pass's vector work).
        A('    const int m_t[2] = {min(n_e, 16), max(n_e - 16, 0)};')
        A('    const int rows_t[2] = {(m_t[0] + 3) >> 2, (m_t[1] + 3) >> 2};')
        A('    auto edge_of_row = [&](int tl, int row) { return 16 * tl + (row >> 2) * rows_t[tl] + (row & 3); };  // (rows r >= R alias a neighbour\'s edge: never used)')
    else:
        A('    auto edge_of_row = [&](int tl, int row) { return 16 * tl + row; };')
    # Round 6: (1) the h2 rows of a tile's EMPTY rows are zero, so the matrix product itself returns w = 0 there and the per-path
    # mask of its result (4 selects + 4 multiplies per path tile) is gone; (2) ONE power-of-two scale for the h2 rows of both tiles
    # of the pass, so its inverse rides on the multiply that scales the output rows (o_scale) instead of on every w tile.
    A('    // per tile: A fragments of h2 (row = edge lane & 15; rows without an edge are zero) and the source row this lane stages (edge lane >> 2)')
    A('    SplitN<NT> ha[2][2];')
    A('    int srs[2];')
    A('    float hv[2][2][8];')
    A('#pragma unroll')
    A('    for (int tl = 0; tl < 2; ++tl) {')
    A('      const int ea = min(eb + edge_of_row(tl, c), e_last);')
    if FROW:
        A('      const bool rv = (c & 3) < rows_t[tl] && (c >> 2) * rows_t[tl] + (c & 3) < m_t[tl];   // this lane\'s A row holds an edge')
    else:
        A('      const bool rv = 16 * tl + c < n_e;')
    A('      const int wra = rv ? (w_row ? w_row[ea] : ea) : 0;')
    A('#pragma unroll')
    A('      for (int q = 0; q < 2; ++q) {')
    A('        f32x4 lo4 = f32x4{0.f, 0.f, 0.f, 0.f}, hi4 = lo4;')
    A('        if (rv) {')
    A('          lo4 = *reinterpret_cast<const f32x4 *>(h2 + (size_t)wra * 64 + 32 * q + 8 * g);')
    A('          hi4 = *reinterpret_cast<const f32x4 *>(h2 + (size_t)wra * 64 + 32 * q + 8 * g + 4);')
    A('        }')
    A('#pragma unroll')
    A('        for (int i = 0; i < 4; ++i) { hv[tl][q][i] = lo4[i]; hv[tl][q][4 + i] = hi4[i]; }')
    A('      }')
    A('      srs[tl] = has_e ? src[min(eb + edge_of_row(tl, lane >> 2), e_last)] : 0;')
    A('    }')
    A('    float o_scale = scale;  // fp16 terms: h2 is scaled per pass (wave-uniform power of two), W2 per matrix on the host; both are undone on the output rows')
    A('    if constexpr (F16) {')
    A('      const int kh = snet::F16_TOP - snet::bound_exp(snet::wave_max(fmaxf(fmaxf(snet::max8(hv[0][0]), snet::max8(hv[0][1])), fmaxf(snet::max8(hv[1][0]), snet::max8(hv[1][1])))));')
    A('      const float sc = snet::pow2f(kh);')
    A('#pragma unroll')
    A('      for (int tl = 0; tl < 2; ++tl)')
    A('#pragma unroll')
    A('        for (int q = 0; q < 2; ++q)')
    A('#pragma unroll')
    A('          for (int i = 0; i < 8; ++i) hv[tl][q][i] *= sc;')
    A('      o_scale = scale * snet::pow2f(-(kh + w2_exp));')
    A('    }')
    A('#pragma unroll')
    A('    for (int tl = 0; tl < 2; ++tl) {')
    A('      ha[tl][0] = splitn8<NT, F16>(hv[tl][0]);')
    A('      ha[tl][1] = splitn8<NT, F16>(hv[tl][1]);')
    A('    }')
    A('    for (int i = lane; i < 32 * NSH; i += 64) {')
    A('      const int el = i / NSH;')
    A('      s_ys[wave][el * NSHP + (i - el * NSH)] = has_e ? sh[(size_t)min(eb + edge_of_row(el >> 4, el & 15), e_last) * NSH + (i - el * NSH)] : 0.f;')
    A('    }')
    first_n = len(fgroups[0][0])
    A(f'    stage_load(0, {first_n}, 0);')
    A(f'    stage_store({first_n}, 0);')
    A('    __syncthreads();  // also orders the wave\'s s_ys stores before its reads')
    SF(1, '    ')
    A('    int sidx = 0, buf = 0;')
    A(f'    f32x4 xq[{MAXD1}];  // the next stage\'s slice, in flight: lane L holds channels 4 (L & 3) .. + 3 of edge L >> 2')
    cx.emit_x_loads('    ', 0, '0', 0)


def _fwd_block(cx: _Gen, ci: int, gi: int):
    """one block = up to FG sub-steps of one (x block, channel tile): per tile the staged source-row slice, per path tile the w
    product and the tensor-product body; then the reduction over the four edge groups and the 16-byte output stores"""
    A, spec, cats, fgroups, FROW, STFL, OOLDS = cx.A, cx.spec, cx.cats, cx.fgroups, cx.FROW, cx.STFL, cx.OOLDS
    SF, emit_x_loads, olists, oo_index = cx.SF, cx.emit_x_loads, cx.olists, cx.oo_index
    cat, grp = cats[ci], fgroups[ci][gi]
    d1, nct = 2 * cat.l1 + 1, cat.mul // 16
    n_here = len(grp)
    # what the next block is (for the slab prefetch): next group of this ct, next ct, or the next x block
    if gi + 1 < len(fgroups[ci]):
        nxt = f'{len(fgroups[ci][gi + 1])}'
    else:
        n_first = len(fgroups[ci][0])
        n_next_cat = len(fgroups[ci + 1][0]) if ci + 1 < len(cats) else 0
        nxt = f'(ct + 1 < {nct} ? {n_first} : {n_next_cat})'
    ol = olists[(ci, gi)]
    A('      {')
    A(f'        const int n_next = (sidx + {n_here} < NS) ? {nxt} : 0;')
    A(f'        const int s_next = sidx + {n_here};')
    A('        const u32x4 *sl = slab[buf];')
    for _, pr in grp:
        for pi in pr:
            if pi is not None:
                A(f'        float acc{pi}[{2 * spec.paths[pi].l3 + 1}];')
                A('#pragma unroll')
                A(f'        for (int i = 0; i < {2 * spec.paths[pi].l3 + 1}; ++i) acc{pi}[i] = 0.f;')
    # The 16 source rows' slice of a tile reaches LDS through registers that were loaded ONE STAGE AHEAD (stage =
    # (block, tile)): without the prefetch every stage exposed a full gather latency -- the waves of this kernel
    # spent 50 % of their cycles in s_waitcnt (SQ_WAIT_ANY, profiles/r02_pmc_sq_fused_kernels.txt).
    def emit_next_block_loads(ind):
        if gi + 1 < len(fgroups[ci]):
            emit_x_loads(ind, ci, 'ct', 0)        # the next group of this (x block, channel tile): same slice
            return
        has_next_cat = ci + 1 < len(cats)
        if nct > 1:
            A(f'{ind}if (ct + 1 < {nct}) {{')
            emit_x_loads(ind + '  ', ci, 'ct + 1', 0)
            A(f'{ind}}}' + (' else {' if has_next_cat else ''))
            if has_next_cat:
                emit_x_loads(ind + '  ', ci + 1, '0', 0)
                A(f'{ind}}}')
        elif has_next_cat:
            emit_x_loads(ind, ci + 1, '0', 0)
    for tl in range(2):
        A(f'        if ({"true" if tl == 0 else "two"}) {{  // tile {tl}')
        A('          {  // stage the 16 source rows\' slice: [m][r][g][channel], conflict-free for the reads below')
        A('            __builtin_amdgcn_wave_barrier();  // every read of the previous tile\'s slice has been issued')
        A('            float *xw = &s_x[wave][(((lane >> 2) & 3) * 4 + (lane >> 4)) * 16 + 4 * (lane & 3)];')
        for m in range(d1):
            A(f'            *reinterpret_cast<f32x4 *>(xw + {m * 256}) = xq[{m}];')
        if tl == 0:
            A('            if (two) {')
            emit_x_loads('              ', ci, 'ct', 1)
            A('            } else {')
            emit_next_block_loads('              ')
            A('            }')
            # the next block's weight fragments are requested AFTER the slice prefetch: vmcnt retires in order,
            # and the wait in front of the next slice store must not also wait for a slab that was just requested
            A('            if (n_next) stage_load(s_next, n_next, buf ^ 1);')
        else:
            emit_next_block_loads('            ')
        A('            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch up here')
        A('            __builtin_amdgcn_wave_barrier();')
        A('          }')
        SF(2, '          ')
        A(f'          float xr[4][{d1}];')
        A('#pragma unroll')
        A('          for (int r = 0; r < 4; ++r)')
        A('#pragma unroll')
        A(f'            for (int m = 0; m < {d1}; ++m) xr[r][m] = s_x[wave][m * 256 + r * 64 + lane];')
        A(f'          const float *ysl = &s_ys[wave][(16 * {tl} + 4 * g) * NSHP];')
        SF(3, '          ')
        chain = [(ls, tp, pi) for ls, pr in grp for tp, pi in enumerate(pr) if pi is not None]
        for k, (ls, tp, pi) in enumerate(chain):
            A(f'          {{  // sub-step {ls} of the block, tile {tp}: path {pi}')
            A('            f32x4 wv = f32x4{0.f, 0.f, 0.f, 0.f};')
            A('#pragma unroll')
            A('            for (int q = 0; q < 2; ++q) {')
            A('              bf16x8 bfr[NT];')
            A('#pragma unroll')
            A(f'              for (int tm = 0; tm < NT; ++tm) bfr[tm] = as_bf16x8(sl[(({ls - grp[0][0]} * 4 + {tp} * 2 + q) * NT + tm) * 64 + lane]);')
            A(f'              wv = mfma16_split<NT, F16>(ha[{tl}][q], bfr, wv);')
            A('            }')
            SF(4, '            ', light=False)
            A(f'            if (!(diag & 1)) fwdf_p{pi}(xr, ysl, wv, acc{pi}, {f"rows_t[{tl}]" if FROW else "4"});  // opaque branch: see the reverse kernel')
            SF(5, '            ', light=False)
            A('          }')
        SF(5, '          ', light=STFL)
        A('        }')
    # Reduce over the 4 edge groups, park in LDS, write out with 16-byte stores.  Round 6: the reduction is a TRANSPOSING butterfly over
    # four output values at a time -- swap_add16(v0, v1) and swap_add16(v2, v3), then swap_add32 of the two results: three swaps and
    # three adds leave value q of the quad, fully summed, in lane group q, and every lane parks one word per quad.  (It was two swaps,
    # two adds and two register copies PER VALUE -- swap_add(a, a) needs a copy of its operand -- with the g == 0 lanes parking all of
    # them: ~7 vector instructions per output value, 45 values per channel tile of the SevenNet-0 middle layer.)
    A('        __builtin_amdgcn_wave_barrier();')
    vals = [f'acc{pi}[{m3}]' for (pi, m3) in ol]
    for k in range(0, len(vals), 4):
        quad = vals[k:k + 4] + ['0.f'] * (4 - len(vals[k:k + 4]))
        A(f'        {{ const float t01 = snet::swap_add16({quad[0]}, {quad[1]}), t23 = snet::swap_add16({quad[2]}, {quad[3]});')
        if len(vals) - k >= 4:
            A(f'          s_o[wave][({k} + g) * 16 + c] = snet::swap_add32(t01, t23) * o_scale; }}')
        else:   # (a short last quad: the groups beyond its values park nothing)
            A(f'          const float u = snet::swap_add32(t01, t23) * o_scale;')
            A(f'          if (g < {len(vals) - k}) s_o[wave][({k} + g) * 16 + c] = u; }}')
    A('        __builtin_amdgcn_wave_barrier();')
    for k in range((len(ol) + 15) // 16):
        if OOLDS:
            A(f'        if (const int oo_ = s_ooff[{16 * oo_index[(ci, gi, k)]} + (lane >> 2)]; live && oo_ >= 0) {{')
            A('          float *o = onode + oo_ + 16 * ct;')
        else:
            A(f'        if (live && ooff{ci}_{gi}_{k} >= 0) {{')
            A(f'          float *o = onode + ooff{ci}_{gi}_{k} + 16 * ct;')
        A(f'          f32x4 v = *reinterpret_cast<const f32x4 *>(&s_o[wave][(16 * {k} + (lane >> 2)) * 16 + 4 * (lane & 3)]);')
        A('          if (pass) v += *reinterpret_cast<const f32x4 *>(o);')
        A('          *reinterpret_cast<f32x4 *>(o) = v;')   # (streaming stores here: measured neutral, round 4)
        A('        }')
    SF(6, '        ')
    A('        if (n_next) stage_store(n_next, buf ^ 1);')
    SF(7, '        ')
    A('        __syncthreads();')
    SF(8, '        ')
    A('        buf ^= 1;')
    A(f'        sidx += {n_here};')
    A('      }')


def _emit_forward_kernel(cx: _Gen):
    A = cx.A
    _fwd_prologue(cx)
    _fwd_pass_top(cx)
    for ci, cat in enumerate(cx.cats):
        A(f'    // ---- x block {cat.i_x}: {cat.mul}x l={cat.l1}, {len(cat.paths)} paths')
        A(f'    for (int ct = 0; ct < {cat.mul // 16}; ++ct) {{')
        for gi in range(len(cx.fgroups[ci])):
            _fwd_block(cx, ci, gi)
        A('    }')
    A('  }')
    if cx.STF:
        A('  if (lane == 0 && live && n_raw < SNET_STAMP_TILES) {')
        A('    for (int i = 0; i < 16; ++i) snet_stamps[n_raw * 16 + i] = ph[i];')
        A('  }')
    A('}')
    A('')


def _variants(exp, default, fwd=False):
    combos = [default]
    if exp:
        cand = ((4, 0, 2), (8, 0, 2), (8, 1, 2), (4, 0, 3), (12, 0, 3), (12, 1, 3)) if fwd else \
            ((4, 0, 2), (8, 0, 2), (8, 1, 2), (4, 0, 3), (4, 1, 3), (12, 0, 3), (12, 1, 3), (8, 1, 4), (8, 0, 4), (8, 0, 3))
        combos += [v for v in cand if v != default]
    return combos


def _emit_launch_bwd(cx: _Gen):
    A, tag, exp, def_b, bwd_cfg = cx.A, cx.tag, cx.exp, cx.def_b, cx.bwd_cfg
    A('template <int NT, bool F16, int NWV, bool GLDS, int OCC>')
    A('void launch_bwd_t(const float *x, const float *sh, const float *dsh, const float *h2, const int32_t *w_row,')
    A('                  const int32_t *row_ptr, const int32_t *src, const int32_t *tile_ptr, const int32_t *tile_node, int64_t n_tiles,')
    A('                  const void *slabs, float scale, const float *g_out, float *g_xe, float *g_h2, float *g_vec, snet::FusedTail tail, hipStream_t st) {')
    A('  const unsigned grid = (unsigned)((n_tiles + NWV - 1) / NWV);')
    A('  int diag = 0;')
    if exp:
        A('  if (const char *e = getenv("SNET_FV_DIAG")) diag = atoi(e);')
    A('  // launches without a g_xe output (first layer, last layer with the transposed convolution) take the instantiation that does not form')
    A('  // the source-row gradient: its multiply-adds and its 4 U d1 registers per lane are gone, not just its stores')
    A('  if (g_xe != nullptr)')
    A(f'    conv_bwdf_{tag}<NT, F16, NWV, GLDS, OCC, true><<<dim3(grid), dim3(64 * NWV), 0, st>>>(x, sh, dsh, h2, w_row, row_ptr, src, tile_ptr, tile_node,')
    A('        (int)n_tiles, static_cast<const u32x4 *>(slabs), scale, g_out, g_xe, g_h2, g_vec, tail, diag);')
    A('  else')
    A(f'    conv_bwdf_{tag}<NT, F16, NWV, GLDS, OCC, false><<<dim3(grid), dim3(64 * NWV), 0, st>>>(x, sh, dsh, h2, w_row, row_ptr, src, tile_ptr, tile_node,')
    A('        (int)n_tiles, static_cast<const u32x4 *>(slabs), scale, g_out, g_xe, g_h2, g_vec, tail, diag);')
    A('}')
    A('void launch_bwd(int nt, const float *x, const float *sh, const float *dsh, const float *h2, const int32_t *w_row,')
    A('                const int32_t *row_ptr, const int32_t *src, const int32_t *tile_ptr, const int32_t *tile_node, int64_t n_tiles,')
    A('                const void *slabs, float scale, const float *g_out, float *g_xe, float *g_h2, float *g_vec, snet::FusedTail tail, hipStream_t st) {')
    args_b = 'x, sh, dsh, h2, w_row, row_ptr, src, tile_ptr, tile_node, n_tiles, slabs, scale, g_out, g_xe, g_h2, g_vec, tail, st'
    if exp:
        A('  int vw = %d, vg = %d, vo = %d;' % def_b)
        A('  if (const char *e = getenv("SNET_FV_BWD")) sscanf(e, "%d,%d,%d", &vw, &vg, &vo);')
        for (w, gl, oc) in _variants(exp, def_b):
            for nt_ in (3, 2):
                A(f'  if (nt == {nt_} && vw == {w} && vg == {gl} && vo == {oc}) return launch_bwd_t<{nt_}, false, {w}, {"true" if gl else "false"}, {oc}>({args_b});')
            A(f'  if (nt == 4 && vw == {w} && vg == {gl} && vo == {oc}) return launch_bwd_t<2, true, {w}, {"true" if gl else "false"}, {oc}>({args_b});')
    w, gl, oc = bwd_cfg(2)
    A(f'  if (nt == 4) launch_bwd_t<2, true, {w}, {"true" if gl else "false"}, {oc}>({args_b});')
    for nt_, kw in ((3, 'else if'), (2, 'else if'), (1, 'else')):
        w, gl, oc = bwd_cfg(nt_)
        cond = f' (nt == {nt_})' if kw != 'else' else ''
        A(f'  {kw}{cond} launch_bwd_t<{nt_}, false, {w}, {"true" if gl else "false"}, {oc}>({args_b});')
    A('}')


def _emit_launch_fwd(cx: _Gen):
    A, tag, exp, def_f, fwd_cfg = cx.A, cx.tag, cx.exp, cx.def_f, cx.fwd_cfg
    A('template <int NT, bool F16, int NWV, bool GLDS, int OCC>')
    A('void launch_fwd_t(const float *x, const float *sh, const float *h2, const int32_t *w_row, const int32_t *row_ptr,')
    A('                  const int32_t *src, int64_t n_dst, const void *slabs, float scale, float *out, int w2_exp, hipStream_t st) {')
    A('  unsigned grid = (unsigned)((n_dst + NWV - 1) / NWV);')
    A('  int diag = 0;')
    if exp:
        A('  if (const char *e = getenv("SNET_FV_DIAG")) diag = atoi(e);')
    A(f'  conv_fwdf_{tag}<NT, F16, NWV, GLDS, OCC><<<dim3(grid), dim3(64 * NWV), 0, st>>>(x, sh, h2, w_row, row_ptr, src, (int)n_dst,')
    A('      static_cast<const u32x4 *>(slabs), scale, out, w2_exp, diag);')
    A('}')
    A('void launch_fwd(int nt, const float *x, const float *sh, const float *h2, const int32_t *w_row, const int32_t *row_ptr,')
    A('                const int32_t *src, int64_t n_dst, const void *slabs, float scale, float *out, int w2_exp, hipStream_t st) {')
    args_f = 'x, sh, h2, w_row, row_ptr, src, n_dst, slabs, scale, out, w2_exp, st'
    if exp:
        A('  int vw = %d, vg = %d, vo = %d;' % def_f)
        A('  if (const char *e = getenv("SNET_FV_FWD")) sscanf(e, "%d,%d,%d", &vw, &vg, &vo);')
        for (w, gl, oc) in _variants(exp, def_f, True):
            for nt_ in (3, 2):
                if nt_ == 3 and w > 8:
                    continue   # would not fit the 160-KB LDS
                A(f'  if (nt == {nt_} && vw == {w} && vg == {gl} && vo == {oc}) return launch_fwd_t<{nt_}, false, {w}, {"true" if gl else "false"}, {oc}>({args_f});')
            A(f'  if (nt == 4 && vw == {w} && vg == {gl} && vo == {oc}) return launch_fwd_t<2, true, {w}, {"true" if gl else "false"}, {oc}>({args_f});')
    w, gl, oc = fwd_cfg(2)
    A(f'  if (nt == 4) launch_fwd_t<2, true, {w}, {"true" if gl else "false"}, {oc}>({args_f});')
    for nt_, kw in ((3, 'else if'), (2, 'else if'), (1, 'else')):
        w, gl, oc = fwd_cfg(nt_)
        cond = f' (nt == {nt_})' if kw != 'else' else ''
        A(f'  {kw}{cond} launch_fwd_t<{nt_}, false, {w}, {"true" if gl else "false"}, {oc}>({args_f});')
    A('}')


def _emit_registration(cx: _Gen):
    A, tag, XT, cols_rev = cx.A, cx.tag, cx.XT, cx.cols_b
    A(f'const snet::FusedKernels kernels = {{"{tag}", DX, DOUT, NSH, WN, NS, SUB_COLS, {len(cols_rev)}, SUB_COLS_B, GXE_CHUNK, launch_bwd, launch_fwd, {1 if XT else 0}}};')
    A('const snet::FusedRegistrar registrar(&kernels);')
    A('}  // namespace')
    if cx.stamped:
        A('// out[0 .. 15] = per-phase cycle sums over the tiles of the LAST launch, out[31] = number of tiles that reported')
        A('extern "C" int snet_debug_stamps(unsigned long long *out, int reset) {')
        A('  static unsigned *host = nullptr;')
        A('  const size_t bytes = sizeof(unsigned) * 16 * SNET_STAMP_TILES;')
        A('  if (!host) host = static_cast<unsigned *>(malloc(bytes));')
        A('  if (out) {')
        A('    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(snet_stamps), bytes) != hipSuccess) return 1;')
        A('    for (int i = 0; i < 32; ++i) out[i] = 0;')
        A('    for (int t = 0; t < SNET_STAMP_TILES; ++t) {')
        A('      unsigned long long tot = 0;')
        A('      for (int i = 0; i < 16; ++i) tot += host[t * 16 + i];')
        A('      if (!tot) continue;')
        A('      for (int i = 0; i < 16; ++i) out[i] += host[t * 16 + i];')
        A('      ++out[31];')
        A('    }')
        A('  }')
        A('  if (reset) { void *dev = nullptr; if (hipGetSymbolAddress(&dev, HIP_SYMBOL(snet_stamps)) != hipSuccess || hipMemset(dev, 0, bytes) != hipSuccess) return 1; }')
        A('  return 0;')
        A('}')


def gen_conv_fused(spec: ConvSpec) -> str:
    cx = _Gen(spec)
    _emit_header(cx)
    _emit_path_functions(cx)
    _emit_reverse_kernel(cx)
    _emit_forward_kernel(cx)
    _emit_launch_bwd(cx)
    _emit_launch_fwd(cx)
    _emit_registration(cx)
    return '\n'.join(cx.L) + '\n'
