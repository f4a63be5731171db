"""debug: fused reverse kernel with / without the g_xe output (the two instantiations) against the separate kernels, f16x3"""
import ctypes as C
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_ops_gpu as T  # noqa: E402
_p = T._p


def run(model, layer, pairs, terms=4, gscale=1.0):
    L, lib = T._lib()
    dev = 'cuda:0'
    c = T._fused_case(model, layer, 40 + layer, pairs)
    spec, nb, wn, dx, dout, nsh, N, E, R = (c[k] for k in ('spec', 'nb', 'wn', 'dx', 'dout', 'nsh', 'N', 'E', 'R'))
    fp = lambda t: t.numpy().ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    mlp, plan, fplan = C.c_void_p(), C.c_void_p(), C.c_void_p()
    cst = 1.6791767923989418
    L.check(lib.snet_radial_mlp_plan_create(nb, 64, 64, wn, fp(c['W0']), fp(c['W1']), fp(c['W2']), 0, cst, 1, C.byref(mlp)))
    L.check(lib.snet_conv_plan_create(spec.tag.encode(), C.byref(plan)))
    L.check(lib.snet_fused_plan_create(plan, mlp, terms, C.byref(fplan)))
    rp, sr = c['row_ptr'].to(dev), c['src'].to(dev)
    wr = None if c['w_row'] is None else c['w_row'].to(dev)
    x, sh, dsh, emb, g_out = (c[k].to(dev) for k in ('x', 'sh', 'dsh', 'emb', 'g_out'))
    g_out = (g_out * gscale).contiguous()
    scale = 0.25
    w_ref = torch.empty(R, wn, device=dev)
    L.check(lib.snet_radial_mlp_fwd(mlp, _p(emb), R, _p(w_ref), None))
    g_w = torch.empty(E, wn, device=dev)
    g_xe_ref = torch.empty(E, dx, device=dev)
    g_vec_ref = torch.zeros(E, 3, device=dev)
    L.check(lib.snet_conv_bwd_edge_vec(plan, _p(x), _p(sh), _p(dsh), _p(w_ref), _p(wr), _p(rp), _p(sr), N, scale,
                                       _p(g_out), _p(g_w), _p(g_xe_ref), _p(g_vec_ref), None))
    h2 = torch.empty(R, 64, device=dev)
    L.check(lib.snet_radial_mlp_hidden_fwd(mlp, _p(emb), R, _p(h2), None))
    tile_ptr, tile_node, n_tiles = T._work_list(L, lib, fplan, rp, c['row_ptr'], N, E, dev)
    x_max, g_max = torch.empty(c['NT'], device=dev), torch.empty(N, device=dev)
    L.check(lib.snet_row_absmax(_p(x), c['NT'], dx, _p(x_max), None))
    L.check(lib.snet_row_absmax(_p(g_out), N, dout, _p(g_max), None))
    res = {}
    for name, with_gx in (('gx', True), ('nogx', False)):
        g_xe = torch.full((E, dx), float('nan'), device=dev)
        g_h2 = torch.full((E, 64), float('nan'), device=dev)
        g_vec = torch.zeros(E, 3, device=dev)
        L.check(lib.snet_conv_bwd_fused(fplan, _p(x), _p(sh), _p(dsh), _p(h2), _p(wr), _p(rp), _p(sr), _p(tile_ptr), _p(tile_node),
                                        n_tiles.value, scale, _p(g_out), _p(g_xe) if with_gx else None, _p(g_h2), None, None, _p(g_vec),
                                        _p(x_max), _p(g_max), None))
        torch.cuda.synchronize()
        res[name] = (g_h2.cpu().double(), g_vec.cpu().double())
    g_h2_ref = g_w.double().cpu() @ c['W2'].double().T
    gv_ref = g_vec_ref.cpu().double()
    out = f'{model}:{layer} {spec.tag}'
    for name in ('gx', 'nogx'):
        a, b = res[name]
        out += f' | {name}: g_h2 err {(a - g_h2_ref).abs().max() / g_h2_ref.abs().max():.2e} g_vec err {(b - gv_ref).abs().max() / gv_ref.abs().max():.2e}'
    d_h2 = (res['gx'][0] - res['nogx'][0]).abs()
    d_v = (res['gx'][1] - res['nogx'][1]).abs()
    out += f' | gx vs nogx: g_h2 {d_h2.max() / g_h2_ref.abs().max():.2e} ({int((d_h2 > 0).sum())} entries, rows {sorted(set(torch.nonzero(d_h2)[:, 0].tolist()))[:8]}) g_vec {d_v.max() / gv_ref.abs().max():.2e}'
    print(out, flush=True)


if __name__ == '__main__':
    for m, l, p in (('sevennet_0', 1, True), ('sevennet_l3i5', 0, False), ('sevennet_l3i5', 1, True), ('sevennet_l3i5', 4, True),
                    ('sevennet_mf_ompa', 0, False), ('sevennet_mf_ompa', 1, True), ('sevennet_mf_ompa', 2, True), ('sevennet_mf_ompa', 3, True), ('sevennet_mf_ompa', 4, True)):
        try:
            run(m, l, p)
        except Exception as e:  # noqa: BLE001
            print(m, l, 'EXC', repr(e)[:300], flush=True)
