#!/usr/bin/env python3
"""Per-kernel timing of the C-ABI entry points at the benchmark's scale (MI355X).

    python tools/microbench.py [--reps 23] [--layer 1] [--iters 5]
    SNET_HIP_LIB=/path/to/experimental.so python tools/microbench.py     # kernel variants

Times each hot kernel of one interaction layer of the SevenNet-0 shape on the
~100k-atom Si graph with random features (values do not matter for timing),
reporting ms and achieved GB/s or TFLOP/s against the algorithmic counts of bench.py.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=23)
    ap.add_argument('--layer', type=int, default=1)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--model', default='sevennet_0')
    ap.add_argument('--mlp-mode', default='bf16x6')
    ap.add_argument('--terms', type=int, default=4)
    ap.add_argument('--fv', default='', help='kernel-tuning builds (SNET_CODEGEN_OPTS=fexp=<tag>): semicolon-separated '
                    '"nwv,glds,occ[,diag]" variants of the fused kernels to time, e.g. "4,0,2;4,1,2;4,1,1;4,1,2,1"')
    ap.add_argument('--only', default='', help='substring filter on kernel names')
    ap.add_argument('--zeros', action='store_true', help='all-zero features and gradients (same instruction stream, no data toggling): '
                    'a power-limited kernel runs faster on them (DVFS), a latency- or issue-limited one does not')
    ap.add_argument('--stamps', action='store_true', help='stamp builds (SNET_CODEGEN_OPTS=stamp=<tag>): print the per-phase '
                    'cycle sums the instrumented reverse kernel collected (snet_debug_stamps)')
    ap.add_argument('--order', default='raster', choices=['raster', 'morton', 'random'], help='atom order of the test cell')
    a = ap.parse_args()
    from bench import kernel_model, model_config
    from sevennet_amd import _lib
    from sevennet_amd.engine import HipForceEngine, _ptr, _stream, build_graph
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config(a.model)
    eng_sd = random_state_dict(cfg, 0)
    eng_sd = {k: (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)) for k, v in eng_sd.items()}
    eng = HipForceEngine(cfg, eng_sd, mlp_mode=a.mlp_mode, fused_terms=a.terms)
    lib = eng.lib
    pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.05, 2)
    if a.order == 'random':
        pos = pos[np.random.default_rng(0).permutation(len(pos))]
    elif a.order == 'morton':
        q = np.floor(pos / 2.7).astype(np.int64)  # ~half-cell boxes
        key = np.zeros(len(pos), np.int64)
        for b in range(8):
            for d in range(3):
                key |= ((q[:, d] >> b) & 1) << (3 * b + d)
        pos = pos[np.argsort(key, kind='stable')]
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    g = build_graph(np.zeros(len(pos), np.int64), ei, ev)
    N, E = g.n_local, g.n_edges
    L = eng.layers[a.layer]
    ls = L.spec
    dx, dmid, wn, nsh, nb = ls.conv.irreps_x.dim, ls.conv.irreps_out.dim, ls.conv.weight_numel, eng.nsh, eng.spec.n_basis
    dev = eng.dev
    rnd = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    emb, sh, dsh = eng._new(E, nb), eng._new(E, nsh), eng._new(E, nsh * 3)
    _lib.check(lib.snet_edge_embed_fwd(C.byref(eng.edge_params), eng.coeffs, _ptr(g.edge_vec), E, _ptr(emb), _ptr(sh),
                                       _ptr(dsh), _stream()))
    h, w, g_m = rnd(N, dx), rnd(E, wn), rnd(N, dmid)
    m, g_w, g_vec, g_h, g_emb = eng._new(N, dmid), eng._new(E, wn), torch.zeros(E, 3, device=dev), eng._new(N, dx), torch.zeros(E, nb, device=dev)
    g_xe = eng._new(E, dx)
    from sevennet_amd.conv_plugin import HipUvuConvolution
    plug = None
    try:
        order_ = sorted(range(len(ls.conv.paths)), key=lambda q: (ls.conv.paths[q].out_off, ls.conv.paths[q].out_ch))
        mid_idx = [0] * len(order_)   # block of irreps_mid (one per path, sorted) that each path fills
        for k_, q in enumerate(order_):
            mid_idx[q] = k_
        ins = [(p_.i_x, p_.i_sh, k_, 'uvu', True) for p_, k_ in zip(ls.conv.paths, mid_idx)]
        plug = HipUvuConvolution(str(ls.conv.irreps_x), str(ls.conv.irreps_sh), str(ls.conv.irreps_mid), ins).to(dev)
    except Exception as exc:  # noqa: BLE001
        print('plugin op unavailable:', exc)
    xq, shq, wq = rnd(N, dx).requires_grad_(True), sh.clone().requires_grad_(True), w.clone().requires_grad_(True)
    goq = rnd(N, dmid)

    def plugin_step():
        out = plug(xq, shq, wq, g.src, g.center)
        out.backward(goq)
        xq.grad = shq.grad = wq.grad = None
    fplug = None
    try:   # the WHOLE convolution behind the plug-in point (hidden radial layers + fused kernels, no weight[E, wn])
        if a.only and 'plugin' not in a.only:
            raise RuntimeError('not requested')
        from sevennet_amd.conv_plugin import HipFusedIrrepsConvolution
        fplug = HipFusedIrrepsConvolution(str(ls.conv.irreps_x), str(ls.conv.irreps_sh), str(ls.conv.irreps_out), list(ls.mlp_dims[:-1]),
                                          'silu', 28.0, fused_terms=a.terms,
                                          sort_by_out=[(p_.i_x, p_.i_sh) for p_ in ls.conv.paths] != sorted((p_.i_x, p_.i_sh) for p_ in ls.conv.paths)).to(dev)
        eidx = torch.stack([g.center.long(), g.src.long()])
        xq_mi, embq = rnd(N, dx).requires_grad_(True), emb.clone().requires_grad_(True)
    except Exception as exc:  # noqa: BLE001
        print('fused plugin module not built:', exc)

    def fused_plugin_step():
        out = fplug({'x': xq_mi, 'edge_attr': shq, 'edge_embedding': embq, 'edge_index': eidx})['x']
        out.backward(goq)
        xq_mi.grad = shq.grad = embq.grad = None
    km = kernel_model(ls, N, E)
    st = _stream()
    h2 = rnd(E, 64)
    g_h2 = rnd(E, 64)
    tile_ptr, tile_node, n_tiles = g.tiles(int(lib.snet_fused_plan_tile_mode(L.fplan)) if L.fplan is not None else 0)
    if a.zeros:
        for t_ in (h, g_m, h2, g_h2, sh, dsh, emb):
            t_.zero_()
    x_max, g_max = h.abs().amax(1).contiguous(), g_m.abs().amax(1).contiguous()   # bounds of the fp16-operand mode
    gy_r, y_r, sc_r = rnd(N, ls.si2.dim_out), rnd(N, ls.gate.irreps_in.dim), rnd(N, ls.gate.irreps_in.dim)
    xo_r = rnd(N, ls.gate.irreps_out.dim)
    from sevennet_amd.model_spec import linear_weight_matrices
    rb16 = torch.full((N,), 40.0, device=dev)   # |randn| < 6, so any row's norm bound
    packs16 = {}

    def run16(lin, transpose, A_, C_, a_stride, c_stride):
        """the linear's per-irrep GEMMs as ONE snet_gemm_grouped_f16 launch (timing only: blocks that share an output would
        need the accumulate flags of _Linear._plan)"""
        key = (id(lin), transpose)
        if key not in packs16:
            sd_w = eng_sd[lin.spec.name]
            mats = linear_weight_matrices(lin.spec, sd_w)
            descs, exps, keep = [], [], []
            for b, mt in zip(lin.spec.blocks, mats):
                Bm = np.ascontiguousarray(mt.T if transpose else mt, np.float32)
                K_, N_ = Bm.shape
                buf = np.empty(int(lib.snet_gemm_f16_size(K_, N_)), np.uint8)
                e_ = C.c_int32()
                _lib.check(lib.snet_gemm_f16_pack(Bm.ctypes.data_as(C.c_void_p), K_, N_, buf.ctypes.data_as(C.c_void_p), C.byref(e_)))
                t_ = torch.from_numpy(buf).to(dev)
                keep.append(t_)
                a_off, c_off = (b.out_off, b.in_off) if transpose else (b.in_off, b.out_off)
                descs.append(_lib.GemmDesc(None, t_.data_ptr(), a_off, c_off, 2 * b.l + 1, K_, N_, 0))
                exps.append(e_.value)
            packs16[key] = ((_lib.GemmDesc * len(descs))(*descs), (C.c_int32 * len(exps))(*exps), len(descs), keep)
        d_, e_, n_, _ = packs16[key]
        for i0 in range(0, n_, 8):
            cnt = min(8, n_ - i0)
            sub_d = (_lib.GemmDesc * cnt)(*[d_[i] for i in range(i0, i0 + cnt)])
            sub_e = (C.c_int32 * cnt)(*[e_[i] for i in range(i0, i0 + cnt)])
            _lib.check(lib.snet_gemm_grouped_f16(sub_d, sub_e, cnt, _ptr(A_), _ptr(C_), N, a_stride, c_stride, None, _ptr(rb16), 1.0, st))
    ops = {
        'radial_mlp_hidden_fwd': lambda: lib.snet_radial_mlp_hidden_fwd(L.mlp_plan, _ptr(emb), E, _ptr(h2), st),
        f'conv_fwd_fused[{ls.conv.tag}]': lambda: lib.snet_conv_fwd_fused(L.fplan, _ptr(h), _ptr(sh), _ptr(h2), _ptr(g.w_row), _ptr(g.row_ptr), _ptr(g.src), N, L.scale, _ptr(m), st),
        f'conv_bwd_fused[{ls.conv.tag}]': lambda: lib.snet_conv_bwd_fused(L.fplan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(h2), _ptr(g.w_row), _ptr(g.row_ptr), _ptr(g.src), _ptr(tile_ptr), _ptr(tile_node), n_tiles, L.scale, _ptr(g_m), _ptr(g_xe), _ptr(g_h2), None, None, _ptr(g_vec), _ptr(x_max), _ptr(g_max), st),
        f'conv_bwd_fused_tail[{ls.conv.tag}]': lambda: lib.snet_conv_bwd_fused(L.fplan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(h2), _ptr(g.w_row), _ptr(g.row_ptr), _ptr(g.src), _ptr(tile_ptr), _ptr(tile_node), n_tiles, L.scale, _ptr(g_m), _ptr(g_xe), None, _ptr(emb), _ptr(g_emb), _ptr(g_vec), _ptr(x_max), _ptr(g_max), st),
        f'conv_bwd_fused_no_gxe[{ls.conv.tag}]': lambda: lib.snet_conv_bwd_fused(L.fplan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(h2), _ptr(g.w_row), _ptr(g.row_ptr), _ptr(g.src), _ptr(tile_ptr), _ptr(tile_node), n_tiles, L.scale, _ptr(g_m), None, _ptr(g_h2), None, None, _ptr(g_vec), _ptr(x_max), _ptr(g_max), st),
        'radial_mlp_hidden_bwd': lambda: lib.snet_radial_mlp_hidden_bwd(L.mlp_plan, _ptr(emb), _ptr(g_h2), E, _ptr(g_emb), st),
        f'radial_mlp_fwd[wn={wn}]': lambda: eng._mlp_fwd(L, emb, E),
        f'conv_fwd[{ls.conv.tag}]': lambda: lib.snet_conv_fwd(L.plan, _ptr(h), _ptr(sh), _ptr(w), None, _ptr(g.row_ptr), _ptr(g.src), N, L.scale, _ptr(m), st),
        f'conv_bwd_edge[{ls.conv.tag}]': lambda: lib.snet_conv_bwd_edge_vec(L.plan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(w), None, _ptr(g.row_ptr), _ptr(g.src), N, L.scale, _ptr(g_m), _ptr(g_w), _ptr(g_xe), _ptr(g_vec), st),
        f'conv_bwd_edge_no_gxe[{ls.conv.tag}]': lambda: lib.snet_conv_bwd_edge_vec(L.plan, _ptr(h), _ptr(sh), _ptr(dsh), _ptr(w), None, _ptr(g.row_ptr), _ptr(g.src), N, L.scale, _ptr(g_m), _ptr(g_w), None, _ptr(g_vec), st),
        'segment_sum_rows': lambda: lib.snet_segment_sum_rows(_ptr(g_xe), _ptr(g.col_ptr), _ptr(g.eperm), N, dx, _ptr(g_h), st),
        f'radial_mlp_bwd[wn={wn}]': lambda: eng._mlp_bwd(L, emb, None, g_w, g_emb, E),
        f'conv_bwd_node[{ls.conv.tag}]': lambda: lib.snet_conv_bwd_node(L.plan, _ptr(sh), _ptr(w), None, _ptr(g.col_ptr), _ptr(g.eperm), _ptr(g.center), N, L.scale, _ptr(g_m), _ptr(g_h), st),
        'plugin_fwd_bwd(b1 autograd op)': plugin_step,
        'plugin_fused_module_fwd_bwd(b1, whole convolution)': fused_plugin_step,
        'si2_fwd': lambda: eng._linear(L.si2, m, N, g),
        'si1_fwd': lambda: eng._linear(L.si1, h, N, g),
        'sc_fwd': lambda: eng._linear(L.sc, h, N, g),
        'si1_bwd': lambda: eng._linear_T(L.si1, h, N, g),
        'si2_bwd': lambda: eng._linear_T(L.si2, gy_r, N, g),
        'si2_bwd_f16x3': lambda: run16(L.si2, True, gy_r, m, ls.si2.dim_out, ls.si2.dim_in),
        'sc_bwd_f16x3': lambda: run16(L.sc, True, gy_r, g_h, ls.sc.dim_out, ls.sc.dim_in),
        'si2_fwd_f16x3': lambda: run16(L.si2, False, m, y_r, ls.si2.dim_in, ls.si2.dim_out),
        'si1_fwd_f16x3': lambda: run16(L.si1, False, h, g_h, ls.si1.dim_in, ls.si1.dim_out),
        # the same launches with every node reading / writing row 0 (strides 0): what the kernel costs without its HBM streams
        'si2_fwd_rows_aliased': lambda: eng._run_groups(L.si2.groups_fwd, m, y_r, N, 0, 0, g),
        'si2_fwd_A_aliased': lambda: eng._run_groups(L.si2.groups_fwd, m, y_r, N, 0, ls.si2.dim_out, g),
        'si2_bwd_rows_aliased': lambda: eng._run_groups(L.si2.groups_T, gy_r, m, N, 0, 0, g),
        'si1_fwd_rows_aliased': lambda: eng._run_groups(L.si1.groups_fwd, h, g_h, N, 0, 0, g),
        'sc_bwd': lambda: eng._linear_T(L.sc, gy_r, N, g),
        'gate_fwd': lambda: lib.snet_gate_fwd(_ptr(y_r), _ptr(sc_r), _ptr(xo_r), N, ls.gate.irreps_in.dim, ls.gate.irreps_out.dim, L.gate_segs, len(ls.gate.segs), st),
        'gate_bwd': lambda: lib.snet_gate_bwd_norm(_ptr(y_r), _ptr(xo_r), _ptr(gy_r), N, ls.gate.irreps_in.dim, ls.gate.irreps_out.dim, L.gate_segs, len(ls.gate.segs), 1.0, _ptr(g_max), st),
        'row_absmax': lambda: lib.snet_row_absmax(_ptr(h), N, dx, _ptr(x_max), st),
    }
    if fplug is None:
        ops.pop('plugin_fused_module_fwd_bwd(b1, whole convolution)')
    if plug is None:
        ops.pop('plugin_fwd_bwd(b1 autograd op)')
    print(f'lib={_lib.LIB_PATH} N={N} E={E} layer={a.layer} dx={dx} dmid={dmid} wn={wn}')
    todo = []
    for name, fn in ops.items():
        if a.only and a.only not in name:
            continue
        if a.fv and 'fused' in name and '[' in name:
            for v in a.fv.split(';'):
                todo.append((f'{name} fv={v}', fn, v))
        else:
            todo.append((name, fn, None))
    for name, fn, fv in todo:
        if fv is not None:
            parts = fv.split(',')
            os.environ['SNET_FV_BWD'] = os.environ['SNET_FV_FWD'] = ','.join(parts[:3])
            os.environ['SNET_FV_DIAG'] = parts[3] if len(parts) > 3 else '0'
        fn()
        torch.cuda.synchronize()
        stamps = a.stamps and ('conv_bwd_fused' in name or 'conv_fwd_fused' in name) and hasattr(lib, 'snet_debug_stamps')
        if stamps:
            lib.snet_debug_stamps(None, 1)
        ts = []
        for _ in range(a.iters):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            fn()
            s1.record()
            torch.cuda.synchronize()
            ts.append(s0.elapsed_time(s1))
        ms = float(np.median(ts))
        extra = ''
        if name.split(' fv=')[0] in km:
            k = km[name.split(' fv=')[0]]
            extra = (f"{k['bytes'] / ms / 1e6:8.1f} GB/s (algorithmic)" if k['bound'] == 'hbm'
                     else f"{k['flops'] / ms / 1e9:8.2f} TFLOP/s")
        print(f'{name:34s} {ms:8.3f} ms  {extra}')
        if stamps:
            buf = (C.c_ulonglong * 32)()
            lib.snet_debug_stamps(buf, 1)
            v = np.array(list(buf), np.float64)
            waves = max(v[31], 1.0)
            tot = v[:16].sum()
            names = ['prologue', 'block top', 'sub-step top (slab request)', 'tile 0: w products', 'tile 0: tensor product',
                     'tile 1: w products', 'tile 1: tensor product', 'split + g_h2 products', 'slab park (vmcnt + ds_write)',
                     'workgroup barrier', 'block end (stores, park, rows)', 'tail A', 'tail B', 'dY epilogue', '-', '-']
            if 'conv_fwd_fused' in name:   # stampf / stampfl builds: the forward kernel's phases, one row per destination node
                names = ['prologue (row pointers, pass count)', 'pass top (h2 split, Y, first slab, barrier)', 'slice store + next slice / slab requests',
                         'slice read back (LDS)', 'w products (fragments + matrix)', 'tensor-product bodies', 'reduce, park, output stores',
                         'slab park (vmcnt + ds_write)', 'workgroup barrier'] + ['-'] * 7
            print(f'    stamps: {int(waves)} wave-launches, {tot / waves:9.0f} cycles per wave')
            for i in range(14):
                print(f'    phase {i:2d} {names[i]:34s} {v[i] / waves:9.0f} cycles per wave  {100 * v[i] / tot:5.1f} %')


if __name__ == '__main__':
    main()
