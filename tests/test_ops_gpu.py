"""GPU: individual C-ABI entry points vs plain PyTorch fp32/fp64 references of the same op."""
import ctypes as C

import numpy as np
import pytest
from helpers import packed_tiles_expected
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from sevennet_amd import _lib
    return _lib, _lib.load()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


@pytest.mark.parametrize('shape', [(1000, 1, 8, 64), (777, 3, 64, 64), (513, 5, 352, 32), (300, 1, 224, 224),
                                   (4097, 1, 64, 960), (129, 1, 960, 64), (50, 3, 4, 12), (33, 1, 2, 1)])
def test_gemm_vs_torch(shape):
    L, lib = _lib()
    n, d, K, N = shape
    dev = 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(1)
    a_stride, a_off = d * K + 8, 4
    c_stride, c_off = d * N + 4, 4
    A = torch.randn(n, a_stride, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)
    Cm = torch.randn(n, c_stride, generator=g).to(dev)
    C0 = Cm.clone()
    ref = (A[:, a_off:a_off + d * K].reshape(n, d, K).double() @ B.double()).reshape(n, d * N)
    L.check(lib.snet_gemm(_p(A), _p(B), _p(Cm), n, d, K, N, a_stride, a_off, c_stride, c_off, None, 0, None))
    torch.cuda.synchronize()
    tol = 2e-6 * K ** 0.5 * 8
    assert (Cm[:, c_off:c_off + d * N].double() - ref).abs().max() < tol
    assert torch.equal(Cm[:, :c_off], C0[:, :c_off])  # nothing outside the block is touched
    L.check(lib.snet_gemm(_p(A), _p(B), _p(Cm), n, d, K, N, a_stride, a_off, c_stride, c_off, None, 1, None))
    torch.cuda.synchronize()
    assert (Cm[:, c_off:c_off + d * N].double() - 2 * ref).abs().max() < 2 * tol
    # row indirection (species-grouped rows)
    rows = torch.randperm(n, generator=g)[: n // 2].to(torch.int32).to(dev)
    C2 = torch.zeros_like(Cm)
    L.check(lib.snet_gemm(_p(A), _p(B), _p(C2), rows.numel(), d, K, N, a_stride, a_off, c_stride, c_off, _p(rows), 0, None))
    torch.cuda.synchronize()
    sel = rows.long()
    assert (C2[sel][:, c_off:c_off + d * N].double() - ref[sel]).abs().max() < tol
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    mask[sel] = False
    assert C2[mask].abs().max() == 0


def test_gemm_grouped_matches_separate_launches():
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(8)
    n, din, dout = 1234, 480, 576
    # SevenNet-0 self-connection shapes: 128x0e->224x0e, 64x1e->64x1e, 32x2e->32x2e
    probs = [(1, 128, 224, 0, 0), (3, 64, 64, 128, 224), (5, 32, 32, 320, 416)]
    A = torch.randn(n, din, generator=g).to(dev)
    Bs = [torch.randn(K, N, generator=g).to(dev) for (_, K, N, _, _) in probs]
    C1 = torch.zeros(n, dout, device=dev)
    C2 = torch.zeros(n, dout, device=dev)
    for (d, K, N, ao, co), B in zip(probs, Bs):
        L.check(lib.snet_gemm(_p(A), _p(B), _p(C1), n, d, K, N, din, ao, dout, co, None, 0, None))
    descs = (L.GemmDesc * 3)(*[L.GemmDesc(B.data_ptr(), None, ao, co, d, K, N, 0) for (d, K, N, ao, co), B in zip(probs, Bs)])
    L.check(lib.snet_gemm_grouped(descs, 3, _p(A), _p(C2), n, din, dout, None, None))
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    ref = (A[:, :128].double() @ Bs[0].double())
    assert (C2[:, :224].double() - ref).abs().max() < 1e-3


def _pack_split(L, lib, B):
    import ctypes as C
    Bh = np.ascontiguousarray(B.cpu().numpy(), np.float32)
    K, N = Bh.shape
    buf = np.empty(int(lib.snet_gemm_split_size(K, N)), np.uint8)
    L.check(lib.snet_gemm_split_pack(C.c_void_p(Bh.ctypes.data), K, N, C.c_void_p(buf.ctypes.data)))
    return torch.from_numpy(buf).to(B.device)


@pytest.mark.parametrize('n', [1234, 70000])   # 32 and 64 rows per wave
@pytest.mark.parametrize('shape', ['sevennet0_sc', 'si2_like', 'ragged'])
def test_gemm_split_precision_vs_fp64(shape, n):
    """bf16 x 6 split-precision GEMM on packed weights: fp32-class accuracy (same bar as the fp32 MFMA
    kernel), including K, N that are not multiples of the 16 x 32 fragment, species row lists and
    accumulation"""
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(11)
    if shape == 'sevennet0_sc':
        din, dout, probs = 480, 576, [(1, 128, 224, 0, 0), (3, 64, 64, 128, 224), (5, 32, 32, 320, 416)]
    elif shape == 'si2_like':
        din, dout, probs = 3136, 576, [(1, 224, 224, 0, 0), (3, 384, 64, 224, 224), (5, 352, 32, 1376, 416)]
    else:
        din, dout, probs = 61, 83, [(1, 4, 7, 0, 0), (3, 9, 20, 4, 7), (5, 6, 3, 31, 67)]
    if n > 10000 and shape == 'si2_like':
        n = 20000
    A = torch.randn(n, din, generator=g).to(dev)
    Bs = [torch.randn(K, N, generator=g).to(dev) for (_, K, N, _, _) in probs]
    packed = [_pack_split(L, lib, B) for B in Bs]
    C1 = torch.full((n, dout), 7.0, device=dev)
    descs = (L.GemmDesc * 3)(*[L.GemmDesc(None, P.data_ptr(), ao, co, d, K, N, 0) for (d, K, N, ao, co), P in zip(probs, packed)])
    L.check(lib.snet_gemm_grouped(descs, 3, _p(A), _p(C1), n, din, dout, None, None))
    torch.cuda.synchronize()
    for (d, K, N, ao, co), B in zip(probs, Bs):
        ref = torch.einsum('nmk,kj->nmj', A[:, ao:ao + d * K].reshape(n, d, K).double(), B.double()).reshape(n, d * N)
        got = C1[:, co:co + d * N].double()
        assert (got - ref).abs().max() <= 3e-6 * ref.abs().max(), (shape, d, K, N)
    # accumulate + row list: only the listed rows change, by exactly one more product
    rows = torch.arange(0, n, 3, dtype=torch.int32, device=dev)
    C2 = C1.clone()
    descs_acc = (L.GemmDesc * 3)(*[L.GemmDesc(None, P.data_ptr(), ao, co, d, K, N, 1) for (d, K, N, ao, co), P in zip(probs, packed)])
    L.check(lib.snet_gemm_grouped(descs_acc, 3, _p(A), _p(C2), rows.numel(), din, dout, _p(rows), None))
    torch.cuda.synchronize()
    d, K, N, ao, co = probs[0]
    sel = rows.long()
    assert torch.allclose(C2[sel, co:co + N], 2 * C1[sel, co:co + N], rtol=1e-5, atol=1e-4)
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    mask[sel] = False
    assert torch.equal(C2[mask], C1[mask])
    # mixing packed and fp32 weights in one group is refused
    mixed = (L.GemmDesc * 2)(L.GemmDesc(Bs[0].data_ptr(), None, 0, 0, 1, probs[0][1], probs[0][2], 0),
                             L.GemmDesc(None, packed[1].data_ptr(), probs[1][3], probs[1][4], probs[1][0], probs[1][1], probs[1][2], 0))
    assert lib.snet_gemm_grouped(mixed, 2, _p(A), _p(C2), n, din, dout, None, None) != 0


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('nb,wn,E', [(8, 960, 5000), (8, 224, 333), (8, 12, 100), (12, 60, 257), (8, 384, 128)])
def test_fused_radial_mlp_vs_torch(nb, wn, E, mode):
    """mode 0 = exact fp32 MFMA, mode 1 = bf16 x 6 split products: both against an fp64 reference"""
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(2)
    emb = torch.randn(E, nb, generator=g).to(dev)
    W0 = (torch.randn(nb, 64, generator=g) / nb ** 0.5).contiguous()
    W1 = (torch.randn(64, 64, generator=g) / 8).contiguous()
    W2 = (torch.randn(64, wn, generator=g) / 8).contiguous()
    cst = 1.6791767923989418
    fp = lambda t: t.numpy().ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    plan = C.c_void_p()
    L.check(lib.snet_radial_mlp_plan_create(nb, 64, 64, wn, fp(W0), fp(W1), fp(W2), 0, cst, mode, C.byref(plan)))
    w = torch.empty(E, wn, device=dev)
    L.check(lib.snet_radial_mlp_fwd(plan, _p(emb), E, _p(w), None))
    torch.cuda.synchronize()
    e64 = emb.double().cpu().requires_grad_(True)
    a1 = torch.nn.functional.silu(e64 @ W0.double()) * cst
    a2 = torch.nn.functional.silu(a1 @ W1.double()) * cst
    ref = a2 @ W2.double()
    assert (w.cpu().double() - ref).abs().max() < 5e-6 * ref.abs().max()
    gw = torch.randn(E, wn, generator=g)
    (gref,) = torch.autograd.grad(ref, e64, gw.double())
    gwd = gw.to(dev)
    g_emb = torch.ones(E, nb, device=dev)  # accumulates
    L.check(lib.snet_radial_mlp_bwd(plan, _p(emb), _p(gwd), E, _p(g_emb), None))
    torch.cuda.synchronize()
    assert (g_emb.cpu().double() - 1.0 - gref).abs().max() < 1e-5 * gref.abs().max()
    lib.snet_radial_mlp_plan_destroy(plan)
    with pytest.raises(RuntimeError):
        L.check(lib.snet_radial_mlp_plan_create(nb, 32, 64, wn, fp(W0), fp(W1), fp(W2), 0, cst, mode, C.byref(plan)))


@pytest.mark.parametrize('lmax,normalize,kind', [(1, 0, 0), (2, 0, 1), (2, 1, 0), (3, 1, 0), (3, 0, 1)])
def test_edge_embedding_fwd_bwd_vs_oracle(lmax, normalize, kind):
    from oracle.e3 import spherical_harmonics
    from oracle.model import bessel_basis, poly_cutoff, xplor_cutoff
    L, lib = _lib()
    dev = 'cuda:0'
    E, nb, rc, r_on = 2000, 8, 5.0, 4.5
    g = torch.Generator().manual_seed(3)
    v = torch.randn(E, 3, generator=g)
    v = v / v.norm(dim=1, keepdim=True) * (1.5 + 3.4 * torch.rand(E, 1, generator=g))
    coeffs = torch.tensor([n * np.pi / rc * (1 + 0.01 * n) for n in range(1, nb + 1)])
    P = L.EdgeParams(rc, nb, kind, 6, r_on, lmax, normalize)
    cf = (C.c_float * nb)(*coeffs.tolist())
    nsh = (lmax + 1) ** 2
    vd = v.to(dev)
    emb, sh, dsh = torch.empty(E, nb, device=dev), torch.empty(E, nsh, device=dev), torch.empty(E, nsh, 3, device=dev)
    L.check(lib.snet_edge_embed_fwd(C.byref(P), cf, _p(vd), E, _p(emb), _p(sh), _p(dsh), None))
    v64 = v.double().requires_grad_(True)
    r = v64.norm(dim=1)
    env = poly_cutoff(r, rc, 6) if kind == 0 else xplor_cutoff(r, rc, r_on)
    emb_ref = bessel_basis(r, coeffs.double(), rc) * env.unsqueeze(-1)
    sh_ref = spherical_harmonics(lmax, v64, bool(normalize))
    torch.cuda.synchronize()
    assert (emb.cpu().double() - emb_ref).abs().max() < 2e-6
    assert (sh.cpu().double() - sh_ref).abs().max() < 5e-6 * max(1.0, sh_ref.abs().max().item())
    g_emb = torch.randn(E, nb, generator=g)
    g_sh = torch.randn(E, nsh, generator=g)
    (gref,) = torch.autograd.grad((emb_ref * g_emb.double()).sum() + (sh_ref * g_sh.double()).sum(), v64)
    gv = torch.full((E, 3), 7.0, device=dev)
    g_emb_d, g_sh_d = g_emb.to(dev), g_sh.to(dev)  # keep the device buffers alive across the async calls
    L.check(lib.snet_edge_embed_bwd(C.byref(P), cf, _p(vd), E, _p(g_emb_d), _p(g_sh_d), _p(gv), 0, None))
    torch.cuda.synchronize()
    tol = 2e-5 * max(1.0, gref.abs().max().item())
    assert (gv.cpu().double() - gref).abs().max() < tol
    # Jacobian path: g_vec = dsh^T g_sh + radial part (accumulate)
    gv2 = torch.einsum('eia,ei->ea', dsh, g_sh_d).contiguous()
    L.check(lib.snet_edge_embed_bwd(C.byref(P), cf, _p(vd), E, _p(g_emb_d), None, _p(gv2), 1, None))
    torch.cuda.synchronize()
    assert (gv2.cpu().double() - gref).abs().max() < tol


def _conv_case(cfg_name):
    from sevennet_amd.model_spec import build_model_spec, sevennet_0_config
    from sevennet_amd.shapes import unit_test_config
    if cfg_name == 'unit_l3':
        return build_model_spec(unit_test_config(lmax=3)).layers[1].conv
    if cfg_name == 'unit_l2':
        return build_model_spec(unit_test_config()).layers[1].conv
    if cfg_name == 'jit_c32_l2':   # NOT in sevennet_amd/shapes.py: compiled on demand by the plug-in (sevennet_amd/jit.py)
        return build_model_spec(unit_test_config(channel=32)).layers[1].conv
    return build_model_spec(sevennet_0_config()).layers[1].conv


@pytest.mark.parametrize('cfg_name', ['unit_l2', 'unit_l3', '7net0_mid', 'jit_c32_l2'])
def test_conv_plugin_vs_oracle_autograd(cfg_name):
    """b1 boundary: HipUvuConvolution (mul_ir in/out, unsorted int32 edges, ghost rows) against the
    oracle's e3nn-style tensor product + scatter, forward and all three gradients."""
    from oracle.e3 import Irreps as OIrreps
    from oracle.model import tp_uvu
    from sevennet_amd.conv_plugin import HipUvuConvolution
    spec = _conv_case(cfg_name)
    dev = 'cuda:0'
    ins = [(p.i_x, p.i_sh, k, 'uvu', True) for p, k in zip(spec.paths, _mid_index(spec))]
    conv = HipUvuConvolution(str(spec.irreps_x), str(spec.irreps_sh), str(spec.irreps_mid), ins).to(dev)
    g = torch.Generator().manual_seed(4)
    N, E = 37, 411
    x = torch.randn(N, spec.irreps_x.dim, generator=g)
    sh = torch.randn(E, spec.irreps_sh.dim, generator=g)
    w = torch.randn(E, spec.weight_numel, generator=g)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 5, (E,), generator=g)  # last rows act as ghosts: sources only
    xd, shd, wd = [t.to(dev).requires_grad_(True) for t in (x, sh, w)]
    out = conv(xd, shd, wd, src.to(dev).to(torch.int32), dst.to(dev).to(torch.int32))
    go = torch.randn(N, spec.irreps_out.dim, generator=g)
    out.backward(go.to(dev))
    torch.cuda.synchronize()
    x64, sh64, w64 = [t.double().requires_grad_(True) for t in (x, sh, w)]
    oins = [(i, j, k) for (i, j, k, _, _) in ins]
    msg = tp_uvu(x64[src], sh64, w64, OIrreps(str(spec.irreps_x)), OIrreps(str(spec.irreps_sh)),
                 OIrreps(str(spec.irreps_mid)), oins)
    ref = torch.zeros(N, msg.shape[1], dtype=torch.float64).index_add_(0, dst, msg)
    ref.backward(go.double())
    for a, b, name in ((out, ref, 'out'), (xd.grad, x64.grad, 'g_x'), (shd.grad, sh64.grad, 'g_sh'), (wd.grad, w64.grad, 'g_w')):
        err = (a.detach().cpu().double() - b.detach()).abs().max().item()
        assert err < 3e-5 * max(1.0, b.abs().max().item()), (name, err)


class _RefConvStub:
    """what sevenn.nn.flash_helper.patch_convolution receives: a not yet instantiated IrrepsConvolution -- restated as a plain
    object with the attributes the reference's class sets in its constructor (sevenn/nn/convolution.py:50-105); the reference
    package itself cannot be imported here (e3nn absent)"""

    def __init__(self, spec, ins, hs, act, denominator):
        self.convolution_kwargs = dict(irreps_in1=str(spec.irreps_x), irreps_in2=str(spec.irreps_sh), irreps_out=str(spec.irreps_mid),
                                       instructions=ins, shared_weights=False, internal_weights=False)
        self.weight_nn_kwargs = dict(hs=hs, act=act)
        self.denominator = torch.nn.Parameter(torch.tensor([denominator]), requires_grad=False)
        self.key_x, self.key_filter, self.key_weight_input, self.key_edge_idx = 'x', 'edge_attr', 'edge_embedding', 'edge_index'
        self.is_parallel = False
        self.layer_instantiated = False


@pytest.mark.parametrize('cfg_name,terms', [('7net0_mid', 4), ('7net0_mid', 3), ('jit_c32_l2', 4)])
def test_fused_convolution_module_vs_oracle_autograd(cfg_name, terms):
    """b1, the whole module (VERDICT r4 missing #3 / next #7): `patch_convolution(irreps_convolution)` returns a module that
    replaces the reference's IrrepsConvolution -- same parameter names (weight_nn.layer{0,1,2}.weight, denominator), same
    forward(data) -- and runs hidden radial layers + fused tensor-product kernels: no weight[E, wn] in memory.  Against the
    oracle's restatement of convolution.py:118-141 (FullyConnectedNet -> uvu tensor product -> scatter -> / denominator) in fp64:
    output and the gradients with respect to x, edge_attr and edge_embedding (what the force autograd needs); unsorted int64
    edge_index as the reference's data loaders produce it."""
    from oracle.e3 import Irreps as OIrreps
    from oracle.model import fcn_apply, tp_uvu
    from sevennet_amd.conv_plugin import HipFusedIrrepsConvolution, patch_convolution
    spec = _conv_case(cfg_name)
    dev = 'cuda:0'
    ins = [(p.i_x, p.i_sh, k, 'uvu', True) for p, k in zip(spec.paths, _mid_index(spec))]
    nb, den = 8, 28.0
    g = torch.Generator().manual_seed(11)
    stub = _RefConvStub(spec, ins, [nb, 64, 64, spec.weight_numel], 'silu', den)
    conv = patch_convolution(stub, fused=True, fused_terms=terms).to(dev)
    assert isinstance(conv, HipFusedIrrepsConvolution)
    assert sorted(k for k, _ in conv.state_dict().items()) == ['denominator', 'weight_nn.layer0.weight', 'weight_nn.layer1.weight', 'weight_nn.layer2.weight']
    W = [torch.randn(nb, 64, generator=g), torch.randn(64, 64, generator=g), torch.randn(64, spec.weight_numel, generator=g)]
    conv.load_state_dict({'denominator': torch.tensor([den]), **{f'weight_nn.layer{k}.weight': W[k] for k in range(3)}})
    N, E = 41, 733
    x = torch.randn(N, spec.irreps_x.dim, generator=g)
    sh = torch.randn(E, spec.irreps_sh.dim, generator=g)
    emb = torch.randn(E, nb, generator=g) * 0.5
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 4, (E,), generator=g)
    xd, shd, ed = [t.to(dev).requires_grad_(True) for t in (x, sh, emb)]
    data = {'x': xd, 'edge_attr': shd, 'edge_embedding': ed, 'edge_index': torch.stack([dst, src]).to(dev)}
    out = conv(data)['x']
    go = torch.randn(N, spec.irreps_out.dim, generator=g)
    out.backward(go.to(dev))
    torch.cuda.synchronize()
    x64, sh64, e64 = [t.double().requires_grad_(True) for t in (x, sh, emb)]
    w64 = fcn_apply(e64, [w.double() for w in W], 'silu')
    oins = [(i, j, k) for (i, j, k, _, _) in ins]
    msg = tp_uvu(x64[src], sh64, w64, OIrreps(str(spec.irreps_x)), OIrreps(str(spec.irreps_sh)), OIrreps(str(spec.irreps_mid)), oins)
    ref = torch.zeros(N, msg.shape[1], dtype=torch.float64).index_add_(0, dst, msg) / den
    ref.backward(go.double())
    for a, b, name in ((out, ref, 'out'), (xd.grad, x64.grad, 'g_x'), (shd.grad, sh64.grad, 'g_edge_attr'), (ed.grad, e64.grad, 'g_edge_embedding')):
        err = (a.detach().cpu().double() - b.detach()).abs().max().item()
        assert err < 3e-5 * max(1.0, b.abs().max().item()), (name, err, b.abs().max().item())
    # a second call with the same weights reuses the plans; changed weights rebuild them
    key = conv._plan_key
    conv({'x': xd.detach(), 'edge_attr': shd.detach(), 'edge_embedding': ed.detach(), 'edge_index': data['edge_index']})
    assert conv._plan_key == key
    with torch.no_grad():
        conv.weight_nn.layer2.weight.mul_(2.0)
    out2 = conv({'x': xd.detach(), 'edge_attr': shd.detach(), 'edge_embedding': ed.detach(), 'edge_index': data['edge_index']})['x']
    assert conv._plan_key != key and (out2 - 2.0 * out.detach()).abs().max().item() < 1e-4 * out.detach().abs().max().item()
    # parallel mode (convolution.py:124-125,137-138): ghost rows arrive separately, are sources only, and the output has the local rows
    n_loc = N - 4
    stub_p = _RefConvStub(spec, ins, [nb, 64, 64, spec.weight_numel], 'silu', den)
    stub_p.is_parallel = True
    conv_p = patch_convolution(stub_p, fused=True, fused_terms=terms).to(dev)
    conv_p.load_state_dict(conv.state_dict())
    with torch.no_grad():
        conv_p.weight_nn.layer2.weight.copy_(W[2].to(dev))     # (conv's last layer was doubled above)
    xl, xg = x[:n_loc].to(dev).requires_grad_(True), x[n_loc:].to(dev).requires_grad_(True)
    out_p = conv_p({'x': xl, 'x_ghost': xg, 'edge_attr': shd.detach(), 'edge_embedding': ed.detach(),
                    'edge_index': data['edge_index']})['x']
    assert out_p.shape == (n_loc, spec.irreps_out.dim)
    assert (out_p.detach().cpu().double() - ref.detach()[:n_loc]).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    out_p.backward(go[:n_loc].to(dev))
    torch.cuda.synchronize()
    # (rows n_loc .. N of the reference output are zero -- no edge ends there -- so the full-output gradient above is the same function)
    gx_all = torch.cat([xl.grad, xg.grad]).cpu().double()
    assert (gx_all - x64.grad).abs().max().item() < 3e-5 * max(1.0, x64.grad.abs().max().item())
    # the fused module is inference-only, and says so instead of training a silently frozen radial network: its radial
    # weights are created without requires_grad; turned on by a trainer, forward() raises (no-grad evaluation still works); a source
    # convolution with a trainable denominator is refused (patch_convolution then falls back to the convolution_cls variant); a
    # double backward (force loss) raises instead of returning zeros
    assert not any(p.requires_grad for p in conv.parameters())
    conv.weight_nn.layer1.weight.requires_grad_(True)
    with pytest.raises(RuntimeError, match='inference module'):
        conv({'x': xd.detach(), 'edge_attr': shd.detach(), 'edge_embedding': ed.detach(), 'edge_index': data['edge_index']})
    with torch.no_grad():
        conv({'x': xd.detach(), 'edge_attr': shd.detach(), 'edge_embedding': ed.detach(), 'edge_index': data['edge_index']})
    conv.weight_nn.layer1.weight.requires_grad_(False)
    stub_td = _RefConvStub(spec, ins, [nb, 64, 64, spec.weight_numel], 'silu', den)
    stub_td.denominator.requires_grad_(True)
    with pytest.raises(NotImplementedError, match='train_denominator'):
        HipFusedIrrepsConvolution.from_irreps_convolution(stub_td)
    x2 = x.to(dev).requires_grad_(True)
    o2 = conv({'x': x2, 'edge_attr': shd.detach(), 'edge_embedding': ed.detach(), 'edge_index': data['edge_index']})['x']
    (gx2,) = torch.autograd.grad(o2.sum(), x2, create_graph=True)
    with pytest.raises(RuntimeError):   # (once_differentiable: the returned gradient carries no graph / a node that raises)
        gx2.sum().backward()
    # shapes without fused kernels (multiplicities not multiples of 16) are refused by the fused module itself
    from sevennet_amd.model_spec import build_model_spec
    from sevennet_amd.shapes import unit_test_config
    small = build_model_spec(unit_test_config()).layers[1].conv
    with pytest.raises(NotImplementedError, match='no fused kernels'):
        HipFusedIrrepsConvolution(str(small.irreps_x), str(small.irreps_sh), str(small.irreps_out), [nb, 64, 64])


def _fused_case(model, layer, seed, pairs):
    """random inputs for one convolution shape: ragged degrees (0, 1, 15, 16, 17, 31, 32, 33, 70 ...),
    ghost source rows, optional pair-shared radial rows (w_row)"""
    from sevennet_amd.model_spec import build_model_spec, sevennet_0_config, sevennet_l3i5_config, sevennet_mf_ompa_config
    cfg = {'sevennet_0': sevennet_0_config, 'sevennet_l3i5': sevennet_l3i5_config, 'sevennet_mf_ompa': sevennet_mf_ompa_config}[model]()
    ms = build_model_spec(cfg)
    spec = ms.layers[layer].conv
    nb, wn, dx, dout, nsh = 8, spec.weight_numel, spec.irreps_x.dim, spec.irreps_out.dim, spec.irreps_sh.dim
    g = torch.Generator().manual_seed(seed)
    deg = torch.tensor([0, 1, 15, 16, 17, 31, 32, 33, 70, 0, 28, 28, 5] + [28] * 23 + [0])
    N = len(deg)
    row_ptr = torch.zeros(N + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(deg, 0)
    E = int(row_ptr[-1])
    NT = N + 7  # ghost rows: sources only
    src = torch.randint(0, NT, (E,), generator=g).to(torch.int32)
    R = E
    w_row = None
    if pairs:  # several edges share one radial row, in scrambled order
        R = E // 2 + 3
        w_row = torch.randint(0, R, (E,), generator=g).to(torch.int32)
    return dict(spec=spec, nb=nb, wn=wn, dx=dx, dout=dout, nsh=nsh, N=N, NT=NT, E=E, R=R, row_ptr=row_ptr, src=src,
                w_row=w_row, x=torch.randn(NT, dx, generator=g), sh=torch.randn(E, nsh, generator=g),
                dsh=torch.randn(E, nsh * 3, generator=g), emb=torch.randn(R, nb, generator=g),
                g_out=torch.randn(N, dout, generator=g),
                W0=(torch.randn(nb, 64, generator=g) / nb ** 0.5).contiguous(),
                W1=(torch.randn(64, 64, generator=g) / 8).contiguous(),
                W2=(torch.randn(64, wn, generator=g) / 8).contiguous())



@pytest.mark.gpu
@pytest.mark.parametrize('pattern', ['crystal28', 'zeros_and_ones', 'long_rows', 'random', 'single_row', 'all_empty'])
def test_packed_tiles_builder_matches_its_definition(pattern):
    """snet_edge_tiles_packed against the greedy restatement above: every edge in exactly one tile, <= 16 edges and <= 2 rows per tile,
    rows with no edges skipped, sub-ranges [lo, hi) carry global row ids and chain into one list (tile_e0[k] is both the end of the
    first list and the start of the second: how the interior / boundary split of a brick uses it)"""
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(5)
    N = 203
    deg = {'crystal28': torch.full((N,), 28), 'zeros_and_ones': torch.randint(0, 2, (N,), generator=g),
           'long_rows': torch.randint(0, 3, (N,), generator=g) * 37 + torch.randint(0, 2, (N,), generator=g),
           'random': torch.randint(0, 41, (N,), generator=g), 'single_row': torch.tensor([45]),
           'all_empty': torch.zeros(N, dtype=torch.long)}[pattern].long()
    N = len(deg)
    row_ptr = torch.zeros(N + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
    E = int(row_ptr[-1])
    rp = row_ptr.to(dev)
    center = torch.repeat_interleave(torch.arange(N), deg)

    def build(lo, hi, cap=None):
        cap = cap if cap is not None else (hi - lo) + E // 16 + 1
        te = torch.full((cap + 1,), -7, dtype=torch.int32, device=dev)
        tn = torch.full((2 * max(cap, 1),), -7, dtype=torch.int32, device=dev)
        n = C.c_int64(-1)
        L.check(lib.snet_edge_tiles_packed(_p(rp), lo, hi, _p(te), _p(tn), cap, C.byref(n), None))
        torch.cuda.synchronize()
        return te.cpu().tolist(), tn.cpu().tolist(), n.value

    for lo, hi in ((0, N), (0, N // 3), (N // 3, N), (7 % N, max(7 % N, N - 5))):
        te, tn, nt = build(lo, hi)
        e0, nodes = packed_tiles_expected(row_ptr, lo, hi)
        assert nt == len(e0) - 1
        if hi > lo:
            assert te[:nt + 1] == e0 and tn[:2 * nt] == nodes
        assert te[nt + 1:] == [-7] * (len(te) - nt - 1)          # nothing written past the sentinel
        covered = []
        for t in range(nt):
            assert 0 < e0[t + 1] - e0[t] <= 16
            rows = set(center[e0[t]:e0[t + 1]].tolist())
            assert rows == set(nodes[2 * t:2 * t + 2]) and nodes[2 * t] <= nodes[2 * t + 1]
            covered += list(range(e0[t], e0[t + 1]))
        assert covered == list(range(int(row_ptr[lo]), int(row_ptr[hi])))
        assert nt <= int(((deg[lo:hi] + 15) // 16).sum())           # never more tiles than the per-row list
    if E > 0 and N > 3:   # capacity is checked, not overrun
        with pytest.raises(RuntimeError, match='capacity'):
            build(0, N, cap=max(1, len(packed_tiles_expected(row_ptr, 0, N)[0]) - 2))


def _work_list(L, lib, fplan, rp, row_ptr_cpu, N, E, dev):
    """the reverse kernel's tile list in the format its plan asks for, checked against the format's definition"""
    n_tiles = C.c_int64()
    deg = (row_ptr_cpu[1:] - row_ptr_cpu[:-1]).long()
    if lib.snet_fused_plan_tile_mode(fplan) == 1:
        cap = N + E // 16 + 1
        tile_ptr = torch.full((cap + 1,), -1, dtype=torch.int32, device=dev)
        tile_node = torch.full((2 * cap,), -1, dtype=torch.int32, device=dev)
        L.check(lib.snet_edge_tiles_packed(_p(rp), 0, N, _p(tile_ptr), _p(tile_node), cap, C.byref(n_tiles), None))
        e0, nodes = packed_tiles_expected(row_ptr_cpu, 0, N)
        nt = n_tiles.value
        assert nt == len(e0) - 1 <= int(((deg + 15) // 16).sum())
        assert tile_ptr.cpu()[:nt + 1].tolist() == e0 and tile_node.cpu()[:2 * nt].tolist() == nodes
        center = torch.repeat_interleave(torch.arange(N), deg)
        for t in range(nt):   # every edge once, <= 16 per tile, all of them in row A or row B
            assert 0 < e0[t + 1] - e0[t] <= 16
            assert set(center[e0[t]:e0[t + 1]].tolist()) == set(nodes[2 * t:2 * t + 2])
        return tile_ptr, tile_node, n_tiles
    tile_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    cap = N + E // 16 + 1
    tile_node = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    L.check(lib.snet_edge_tiles(_p(rp), N, _p(tile_ptr), _p(tile_node), cap, C.byref(n_tiles), None))
    assert n_tiles.value == int(((deg + 15) // 16).sum())
    assert torch.equal(tile_ptr.cpu()[1:].long(), torch.cumsum((deg + 15) // 16, 0))
    assert torch.equal(tile_node.cpu()[:n_tiles.value].long(), torch.repeat_interleave(torch.arange(N), (deg + 15) // 16))
    return tile_ptr, tile_node, n_tiles


@pytest.mark.parametrize('model,layer,pairs,terms,gscale', [
    ('sevennet_0', 0, False, 3, 1.0), ('sevennet_0', 1, True, 3, 1.0), ('sevennet_0', 4, True, 3, 1.0),
    ('sevennet_0', 1, False, 2, 1.0), ('sevennet_0', 1, True, 1, 1.0), ('sevennet_l3i5', 1, True, 3, 1.0),
    ('sevennet_l3i5', 0, False, 3, 1.0),
    # f16x3 (engine default): fp32-rounding class, also with gradients 7 orders of magnitude below / 5 above unity
    # (the kernels scale their fp16 operands per tile; fp16 itself spans 2^-24 .. 2^16)
    ('sevennet_0', 0, False, 4, 1.0), ('sevennet_0', 1, True, 4, 1.0), ('sevennet_0', 4, True, 4, 3e5),
    ('sevennet_0', 1, False, 4, 1e-7), ('sevennet_l3i5', 1, True, 4, 1.0), ('sevennet_mf_ompa', 2, True, 4, 40.0)])
def test_conv_fused_matches_separate_kernels(model, layer, pairs, terms, gscale):
    """Radial-MLP last layer inside the tensor-product kernels (w and g_w never in memory) ==
    snet_radial_mlp_fwd + snet_conv_fwd and snet_conv_bwd_edge_vec + snet_radial_mlp_bwd:
    forward rows, per-edge source-row gradients, dE/d(edge_vec) and the radial-embedding gradient.
    terms = 3 (bf16x6) and 4 (f16x3) must agree to fp32 rounding; 2 and 1 to their stated precision.
    gscale multiplies the incoming gradient g_out (every reverse output is linear in it)."""
    L, lib = _lib()
    dev = 'cuda:0'
    c = _fused_case(model, layer, 40 + layer, pairs)
    spec, nb, wn, dx, dout, nsh, N, E, R = (c[k] for k in ('spec', 'nb', 'wn', 'dx', 'dout', 'nsh', 'N', 'E', 'R'))
    fp = lambda t: t.numpy().ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    mlp, plan, fplan = C.c_void_p(), C.c_void_p(), C.c_void_p()
    cst = 1.6791767923989418
    L.check(lib.snet_radial_mlp_plan_create(nb, 64, 64, wn, fp(c['W0']), fp(c['W1']), fp(c['W2']), 0, cst, 1, C.byref(mlp)))
    L.check(lib.snet_conv_plan_create(spec.tag.encode(), C.byref(plan)))
    assert lib.snet_conv_fused_available(plan) == 1
    L.check(lib.snet_fused_plan_create(plan, mlp, terms, C.byref(fplan)))
    rp, sr = c['row_ptr'].to(dev), c['src'].to(dev)
    wr = None if c['w_row'] is None else c['w_row'].to(dev)
    x, sh, dsh, emb, g_out = (c[k].to(dev) for k in ('x', 'sh', 'dsh', 'emb', 'g_out'))
    g_out = (g_out * gscale).contiguous()
    scale = 0.25
    # ---- separate kernels (reference)
    w_ref = torch.empty(R, wn, device=dev)
    out_ref = torch.empty(N, dout, device=dev)
    L.check(lib.snet_radial_mlp_fwd(mlp, _p(emb), R, _p(w_ref), None))
    L.check(lib.snet_conv_fwd(plan, _p(x), _p(sh), _p(w_ref), _p(wr), _p(rp), _p(sr), N, scale, _p(out_ref), None))
    g_w = torch.empty(E, wn, device=dev)
    g_xe_ref = torch.empty(E, dx, device=dev)
    g_vec_ref = torch.full((E, 3), gscale, device=dev)   # accumulated into
    L.check(lib.snet_conv_bwd_edge_vec(plan, _p(x), _p(sh), _p(dsh), _p(w_ref), _p(wr), _p(rp), _p(sr), N, scale,
                                       _p(g_out), _p(g_w), _p(g_xe_ref), _p(g_vec_ref), None))
    emb_e = emb if wr is None else emb[wr.long()].contiguous()   # per directed edge
    g_emb_ref = torch.full((E, nb), gscale, device=dev)
    L.check(lib.snet_radial_mlp_bwd(mlp, _p(emb_e), _p(g_w), E, _p(g_emb_ref), None))
    # ---- fused kernels
    h2 = torch.empty(R, 64, device=dev)
    L.check(lib.snet_radial_mlp_hidden_fwd(mlp, _p(emb), R, _p(h2), None))
    out = torch.full((N, dout), float('nan'), device=dev)
    L.check(lib.snet_conv_fwd_fused(fplan, _p(x), _p(sh), _p(h2), _p(wr), _p(rp), _p(sr), N, scale, _p(out), None))
    tile_ptr, tile_node, n_tiles = _work_list(L, lib, fplan, rp, c['row_ptr'], N, E, dev)
    g_xe = torch.full((E, dx), float('nan'), device=dev)
    g_h2 = torch.full((E, 64), float('nan'), device=dev)
    # row maxima the fp16-operand mode (terms = 4) bounds each edge's g_w with; other modes ignore them
    x_max, g_max = torch.empty(c['NT'], device=dev), torch.empty(N, device=dev)
    L.check(lib.snet_row_absmax(_p(x), c['NT'], dx, _p(x_max), None))
    L.check(lib.snet_row_absmax(_p(g_out), N, dout, _p(g_max), None))
    torch.cuda.synchronize()
    assert torch.equal(x_max, x.abs().amax(1)) and torch.equal(g_max, g_out.abs().amax(1))
    if terms == 4:
        with pytest.raises(RuntimeError, match='x_rowmax'):
            L.check(lib.snet_conv_bwd_fused(fplan, _p(x), _p(sh), _p(dsh), _p(h2), _p(wr), _p(rp), _p(sr), _p(tile_ptr), _p(tile_node),
                                            n_tiles.value, scale, _p(g_out), _p(g_xe), _p(g_h2), None, None, _p(g_xe), None, None, None))
    g_vec = torch.full((E, 3), gscale, device=dev)
    L.check(lib.snet_conv_bwd_fused(fplan, _p(x), _p(sh), _p(dsh), _p(h2), _p(wr), _p(rp), _p(sr), _p(tile_ptr), _p(tile_node),
                                    n_tiles.value, scale, _p(g_out), _p(g_xe), _p(g_h2), None, None, _p(g_vec), _p(x_max), _p(g_max), None))
    g_emb = torch.full((E, nb), gscale, device=dev)
    L.check(lib.snet_radial_mlp_hidden_bwd(mlp, _p(emb_e), _p(g_h2), E, _p(g_emb), None))
    # ... and with the MLP's hidden layers reversed inside the same kernel (g_h2 never written)
    assert lib.snet_fused_plan_has_mlp_tail(fplan) == 1
    g_emb_t = torch.full((E, nb), gscale, device=dev)
    g_xe_t = torch.full((E, dx), float('nan'), device=dev)
    g_vec_t = torch.full((E, 3), gscale, device=dev)
    L.check(lib.snet_conv_bwd_fused(fplan, _p(x), _p(sh), _p(dsh), _p(h2), _p(wr), _p(rp), _p(sr), _p(tile_ptr), _p(tile_node),
                                    n_tiles.value, scale, _p(g_out), _p(g_xe_t), None, _p(emb_e), _p(g_emb_t), _p(g_vec_t), _p(x_max), _p(g_max), None))
    with pytest.raises(RuntimeError):  # exactly one of g_h2 / g_emb
        L.check(lib.snet_conv_bwd_fused(fplan, _p(x), _p(sh), _p(dsh), _p(h2), _p(wr), _p(rp), _p(sr), _p(tile_ptr), _p(tile_node),
                                        n_tiles.value, scale, _p(g_out), _p(g_xe_t), _p(g_h2), _p(emb_e), _p(g_emb_t), _p(g_vec_t), _p(x_max), _p(g_max), None))
    torch.cuda.synchronize()
    a1 = torch.nn.functional.silu(c['emb'].double() @ c['W0'].double()) * cst
    a2 = torch.nn.functional.silu(a1 @ c['W1'].double()) * cst
    assert (h2.cpu().double() - a2).abs().max() < 5e-6 * a2.abs().max()
    for t in (out, g_xe, g_h2, g_vec, g_emb, g_emb_t):
        assert not torch.isnan(t).any()
    # the fused kernel keeps the 16-channel chunks of a g_xe row in its own order (snet_fused_plan_gxe_chunks): undo it,
    # and check that snet_segment_sum_rows_chunked returns the standard-order sums bit for bit
    cp = (C.c_int32 * (dx // 16))()
    L.check(lib.snet_fused_plan_gxe_chunks(fplan, cp, dx // 16))
    cpos = torch.tensor(list(cp), dtype=torch.long)
    assert sorted(cpos.tolist()) == list(range(dx // 16))
    col = (cpos[:, None] * 16 + torch.arange(16)[None, :]).reshape(-1).to(dev)
    g_xe_raw, g_xe = g_xe, g_xe[:, col].contiguous()
    order = torch.argsort(sr.long(), stable=True).to(torch.int32)
    col_ptr = torch.zeros(c['NT'] + 1, dtype=torch.int32, device=dev)
    col_ptr[1:] = torch.cumsum(torch.bincount(sr.long(), minlength=c['NT']), 0).to(torch.int32)
    s_std, s_chk = torch.empty(c['NT'], dx, device=dev), torch.empty(c['NT'], dx, device=dev)
    L.check(lib.snet_segment_sum_rows(_p(g_xe), _p(col_ptr), _p(order), c['NT'], dx, _p(s_std), None))
    L.check(lib.snet_segment_sum_rows_chunked(_p(g_xe_raw), _p(col_ptr), _p(order), c['NT'], dx,
                                              _p(torch.tensor(list(cp), dtype=torch.int32, device=dev)), _p(s_chk), None))
    torch.cuda.synchronize()
    assert torch.equal(s_std, s_chk)
    g_xe_t = g_xe_t[:, col].contiguous()
    assert torch.equal(g_xe_t, g_xe) and torch.equal(g_vec_t, g_vec)
    # the tail multiplies in the kernel's own precision class (`terms`), the separate hidden-layer kernel in bf16x6
    assert (g_emb_t - g_emb).abs().max().item() <= {4: 2e-6, 3: 2e-6, 2: 1e-4, 1: 4e-2}[terms] * max(gscale, g_emb.abs().max().item())
    tol = {4: 2e-5, 3: 2e-5, 2: 1e-4, 1: 4e-2}[terms]
    # g_h2 against the fp64 contraction of the separate kernel's g_w with W2^T
    g_h2_ref = g_w.double().cpu() @ c['W2'].double().T
    for name, a, b in (('out', out, out_ref), ('g_xe', g_xe, g_xe_ref), ('g_vec', g_vec, g_vec_ref),
                       ('g_h2', g_h2.cpu().double(), g_h2_ref), ('g_emb', g_emb, g_emb_ref)):
        err = (a.double().cpu() - b.double().cpu()).abs().max().item()
        floor = 1.0 if name == 'out' else gscale
        assert err <= tol * max(floor, b.abs().max().item()), (name, err, b.abs().max().item())
    assert out[0].abs().max() == 0 and out[9].abs().max() == 0 and out[N - 1].abs().max() == 0   # nodes without edges
    # g_xe is optional (first layer: inputs depend on species only)
    g_h2b = torch.empty_like(g_h2)
    g_vecb = torch.full((E, 3), gscale, device=dev)
    L.check(lib.snet_conv_bwd_fused(fplan, _p(x), _p(sh), _p(dsh), _p(h2), _p(wr), _p(rp), _p(sr), _p(tile_ptr), _p(tile_node),
                                    n_tiles.value, scale, _p(g_out), None, _p(g_h2b), None, None, _p(g_vecb), _p(x_max), _p(g_max), None))
    torch.cuda.synchronize()
    # (two instantiations of one source: the same arithmetic, but hipcc contracts / orders it per instantiation -- since round 6,
    # where the bodies run on the raw matrix accumulators, a last-bit difference in g_h2 on the lmax-3 middle shape; both are
    # checked against the fp64 contraction above / here, and each HOST uses one instantiation per layer, so hosts stay bit-identical)
    assert torch.equal(g_vec, g_vecb)
    assert (g_h2b.double().cpu() - g_h2_ref).abs().max().item() <= tol * max(gscale, g_h2_ref.abs().max().item())
    assert (g_h2b - g_h2).abs().max().item() <= 1e-6 * g_h2.abs().max().item()
    lib.snet_fused_plan_destroy(fplan)
    lib.snet_conv_plan_destroy(plan)
    lib.snet_radial_mlp_plan_destroy(mlp)


def test_conv_plugin_edge_plan_cache_and_sorted_input():
    """the reference passes the same (center-sorted) edge tensors to every layer of a step: the CSR plan
    is built once and shared between modules, sorted input takes the no-permutation path, an in-place
    change of the edge tensors invalidates the plan"""
    from sevennet_amd import conv_plugin
    from sevennet_amd.conv_plugin import HipUvuConvolution
    spec = _conv_case('unit_l2')
    dev = 'cuda:0'
    ins = [(p.i_x, p.i_sh, k, 'uvu', True) for p, k in zip(spec.paths, _mid_index(spec))]
    a = HipUvuConvolution(str(spec.irreps_x), str(spec.irreps_sh), str(spec.irreps_mid), ins).to(dev)
    b = HipUvuConvolution(str(spec.irreps_x), str(spec.irreps_sh), str(spec.irreps_mid), ins).to(dev)
    g = torch.Generator().manual_seed(9)
    N, E = 29, 300
    x = torch.randn(N, spec.irreps_x.dim, generator=g).to(dev)
    sh = torch.randn(E, spec.irreps_sh.dim, generator=g).to(dev)
    w = torch.randn(E, spec.weight_numel, generator=g).to(dev)
    dst = torch.sort(torch.randint(0, N, (E,), generator=g)).values.to(torch.int32).to(dev)   # sorted by center
    src = torch.randint(0, N, (E,), generator=g).to(torch.int32).to(dev)
    conv_plugin._PLAN_CACHE.clear()
    out_a = a(x, sh, w, src, dst)
    assert len(conv_plugin._PLAN_CACHE) == 1
    plan = next(iter(conv_plugin._PLAN_CACHE.values()))
    assert plan.order is None                                    # already sorted: rows are used in place
    out_b = b(x, sh, w, src[:], dst[:])                          # new view objects of the same storage, other module
    assert len(conv_plugin._PLAN_CACHE) == 1 and next(iter(conv_plugin._PLAN_CACHE.values())) is plan
    assert torch.equal(out_a, out_b)
    # same numbers through the unsorted path
    perm = torch.randperm(E, generator=g).to(dev)
    out_p = a(x, sh[perm], w[perm], src[perm].contiguous(), dst[perm].contiguous())
    assert (out_p - out_a).abs().max() <= 1e-5 * out_a.abs().max()
    # in-place edit of the edge list -> new plan, different result
    src2 = src.clone()
    out_c = a(x, sh, w, src2, dst)
    src2[0] = (src2[0] + 1) % N
    out_d = a(x, sh, w, src2, dst)
    torch.cuda.synchronize()
    assert torch.equal(out_c, out_a) and not torch.equal(out_d, out_c)


def test_segment_sum_rows():
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(6)
    E, N, dim = 999, 57, 480
    x = torch.randn(E, dim, generator=g).to(dev)
    src = torch.randint(0, N, (E,), generator=g)
    perm = torch.sort(src, stable=True).indices.to(torch.int32).to(dev)
    ptr = torch.zeros(N + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.bincount(src, minlength=N), 0)
    ptr = ptr.to(torch.int32).to(dev)
    out = torch.empty(N, dim, device=dev)
    L.check(lib.snet_segment_sum_rows(_p(x), _p(ptr), _p(perm), N, dim, _p(out), None))
    torch.cuda.synchronize()
    ref = torch.zeros(N, dim, dtype=torch.float64).index_add_(0, src, x.cpu().double())
    assert (out.cpu().double() - ref).abs().max() < 1e-5


def test_elementwise_node_kernels_beyond_the_grid_cap():
    """the elementwise node kernels run on a capped grid (8192 x 256 threads): every one of them has to
    stride over inputs larger than that (the multi-modal bias add did not: MF-ompa above ~2500 atoms)"""
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(8)
    n, dim = 2744, 832  # > 2^21 elements
    y0 = torch.randn(n, dim, generator=g).to(dev)
    bias = torch.randn(dim, generator=g).to(dev)
    y = y0.clone()
    L.check(lib.snet_add_row_bias(_p(y), _p(bias), n, dim, None))
    assert torch.equal(y, y0 + bias[None])
    x = torch.randn(n, dim, generator=g).to(dev)
    y = y0.clone()
    L.check(lib.snet_add_inplace(_p(y), _p(x), n * dim, None))
    assert torch.equal(y, y0 + x)
    table = torch.randn(7, dim, generator=g).to(dev)
    types = torch.randint(0, 7, (n,), generator=g).to(torch.int32).to(dev)
    out = torch.empty(n, dim, device=dev)
    L.check(lib.snet_embed_rows(_p(table), _p(types), _p(out), n, dim, None))
    assert torch.equal(out, table[types.long()])
    idx = torch.randperm(n, generator=g).to(torch.int32).to(dev)
    L.check(lib.snet_gather_rows(_p(y0), _p(idx), _p(out), n, dim, None))
    assert torch.equal(out, y0[idx.long()])
    a = torch.zeros(n, dim, device=dev)
    L.check(lib.snet_act_fwd(_p(y0), _p(a), n * dim, 0, 1.5, None))
    torch.cuda.synchronize()
    assert bool((a[-1] != 0).any()) and bool(torch.isfinite(a).all())


def _mid_index(spec):
    """index of each path's block inside irreps_mid (sorted, one block per path)"""
    offs = {}
    out = []
    # blocks of irreps_mid with the same irrep appear in the order their paths fill the merged block
    mid = list(spec.irreps_mid)
    used = [False] * len(mid)
    order = sorted(range(len(spec.paths)), key=lambda q: (spec.paths[q].out_off, spec.paths[q].out_ch))
    k = 0
    idx = [0] * len(spec.paths)
    for q in order:
        idx[q] = k
        k += 1
    return idx


def test_gate_and_halo_kernels():
    from oracle.model import GateSpec
    from oracle.e3 import Irreps as OIrreps
    from sevennet_amd.irreps import Irreps, mulir_to_irmul_index, irmul_to_mulir_index
    from sevennet_amd.model_spec import ACT_CST, make_gate
    L, lib = _lib()
    dev = 'cuda:0'
    irr = '4x0o+6x0e+3x1o+5x1e+2x2e'
    act = {'e': 'silu', 'o': 'tanh'}
    gs = make_gate(Irreps(irr), act, act)
    og = GateSpec(OIrreps(irr), act, act)
    N = 301
    g = torch.Generator().manual_seed(5)
    y = torch.randn(N, gs.irreps_in.dim, generator=g)
    y64 = y.double().requires_grad_(True)
    ref = og.apply(y64)
    go = torch.randn(N, gs.irreps_out.dim, generator=g)
    (gref,) = torch.autograd.grad(ref, y64, go.double())
    segs = (L.GateSeg * len(gs.segs))()
    inv = {0: 'silu', 1: 'tanh'}
    for i, s in enumerate(gs.segs):
        segs[i] = L.GateSeg(s.kind, s.in_off, s.out_off, s.mul, s.l, s.gate_off, s.act, ACT_CST[inv[s.act]])
    to_im_in = torch.as_tensor(mulir_to_irmul_index(gs.irreps_in))
    to_im_out = torch.as_tensor(mulir_to_irmul_index(gs.irreps_out))
    yd = y[:, to_im_in].contiguous().to(dev)
    out = torch.empty(N, gs.irreps_out.dim, device=dev)
    L.check(lib.snet_gate_fwd(_p(yd), None, _p(out), N, gs.irreps_in.dim, gs.irreps_out.dim, segs, len(gs.segs), None))
    gy = torch.empty_like(yd)
    god = go[:, to_im_out].contiguous().to(dev)
    L.check(lib.snet_gate_bwd(_p(yd), _p(god), _p(gy), N, gs.irreps_in.dim, gs.irreps_out.dim, segs, len(gs.segs), None))
    torch.cuda.synchronize()
    back_out = torch.as_tensor(irmul_to_mulir_index(gs.irreps_out))
    back_in = torch.as_tensor(irmul_to_mulir_index(gs.irreps_in))
    assert (out.cpu()[:, back_out].double() - ref.detach()).abs().max() < 2e-6
    assert (gy.cpu()[:, back_in].double() - gref).abs().max() < 5e-6
    # fused self-connection add: gate(y + a), and y is updated in place to y + a
    a_ = torch.randn(N, gs.irreps_in.dim, generator=g).to(dev)
    y2, out2 = (yd - a_).contiguous(), torch.empty_like(out)
    L.check(lib.snet_gate_fwd(_p(y2), _p(a_), _p(out2), N, gs.irreps_in.dim, gs.irreps_out.dim, segs, len(gs.segs), None))
    torch.cuda.synchronize()
    assert (y2 - yd).abs().max() < 1e-6 and (out2 - out).abs().max() < 1e-5
    # halo pack / unpack
    x = torch.randn(50, 7, generator=g).to(dev)
    idx = torch.randperm(50, generator=g)[:20].to(torch.int32).to(dev)
    o = torch.empty(20, 7, device=dev)
    L.check(lib.snet_gather_rows(_p(x), _p(idx), _p(o), 20, 7, None))
    yb = torch.zeros(50, 7, device=dev)
    L.check(lib.snet_scatter_add_rows(_p(o), _p(idx), _p(yb), 20, 7, None))
    torch.cuda.synchronize()
    assert torch.equal(o, x[idx.long()])
    assert torch.equal(yb[idx.long()], x[idx.long()])
    mask = torch.ones(50, dtype=torch.bool, device=dev)
    mask[idx.long()] = False
    assert yb[mask].abs().max() == 0


@pytest.mark.parametrize('cfg_name', ['7net0_mid', 'unit_l3'])
def test_conv_bwd_edge_ragged_degrees_vs_autograd(cfg_name):
    """reverse per-edge kernels on nodes with 0, 1, 2, 3, 31..33, 63..65, 70 and 129 edges (odd counts
    and > 64-edge multi-pass rows matter for the blocks that run two edges per wavefront): g_w, g_xe and
    d/dY against autograd of the oracle's tensor product, and the Jacobian-contracted variant against it"""
    from oracle.e3 import Irreps as OIrreps
    from oracle.model import tp_uvu
    from sevennet_amd.irreps import irmul_to_mulir_index, mulir_to_irmul_index
    L, lib = _lib()
    dev = 'cuda:0'
    spec = _conv_case(cfg_name)
    dx, dout, nsh, wn = spec.irreps_x.dim, spec.irreps_out.dim, spec.irreps_sh.dim, spec.weight_numel
    g = torch.Generator().manual_seed(17)
    deg = torch.tensor([0, 1, 2, 3, 31, 32, 33, 63, 64, 65, 70, 129, 0, 5])
    N = len(deg)
    row_ptr = torch.zeros(N + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(deg, 0)
    E = int(row_ptr[-1])
    NT = N + 3
    src = torch.randint(0, NT, (E,), generator=g).to(torch.int32)
    dst = torch.repeat_interleave(torch.arange(N), deg)
    x = torch.randn(NT, dx, generator=g)
    sh = torch.randn(E, nsh, generator=g)
    w = torch.randn(E, wn, generator=g)
    go = torch.randn(N, dout, generator=g)
    dsh = torch.randn(E, nsh, 3, generator=g)
    dsh[:, 0] = 0.0                                    # Y_0 is constant: the kernel skips its Jacobian row
    # oracle (mul_ir layouts) -> convert to the engine's ir_mul
    to_x = torch.as_tensor(mulir_to_irmul_index(spec.irreps_x))
    to_o = torch.as_tensor(mulir_to_irmul_index(spec.irreps_out))
    from_x, from_o = torch.as_tensor(irmul_to_mulir_index(spec.irreps_x)), torch.as_tensor(irmul_to_mulir_index(spec.irreps_out))
    x64 = x[:, from_x].double().requires_grad_(True)   # x given in ir_mul; oracle wants mul_ir
    sh64, w64 = sh.double().requires_grad_(True), w.double().requires_grad_(True)
    ins = [(p.i_x, p.i_sh, k) for p, k in zip(spec.paths, _mid_index(spec))]
    msg = tp_uvu(x64[src.long()], sh64, w64, OIrreps(str(spec.irreps_x)), OIrreps(str(spec.irreps_sh)),
                 OIrreps(str(spec.irreps_mid)), ins)
    out = torch.zeros(N, dout, dtype=torch.float64).index_add_(0, dst, msg)
    (out * go[:, from_o].double()).sum().backward()
    plan = C.c_void_p()
    L.check(lib.snet_conv_plan_create(spec.tag.encode(), C.byref(plan)))
    xd, shd, wd, god, dshd = x.to(dev), sh.to(dev), w.to(dev), go.to(dev), dsh.reshape(E, nsh * 3).contiguous().to(dev)
    rp, sr = row_ptr.to(dev), src.to(dev)
    g_w, g_xe, g_sh = torch.empty(E, wn, device=dev), torch.empty(E, dx, device=dev), torch.zeros(E, nsh, device=dev)
    L.check(lib.snet_conv_bwd_edge(plan, _p(xd), _p(shd), _p(wd), None, _p(rp), _p(sr), N, 1.0, _p(god), _p(g_w), _p(g_xe),
                                   _p(g_sh), None))
    g_w2, g_xe2, g_vec = torch.empty_like(g_w), torch.empty_like(g_xe), torch.zeros(E, 3, device=dev)
    L.check(lib.snet_conv_bwd_edge_vec(plan, _p(xd), _p(shd), _p(dshd), _p(wd), None, _p(rp), _p(sr), N, 1.0, _p(god),
                                       _p(g_w2), _p(g_xe2), _p(g_vec), None))
    torch.cuda.synchronize()
    tol = lambda ref: 3e-5 * max(1.0, ref.abs().max().item())  # noqa: E731
    assert (g_w.cpu().double() - w64.grad).abs().max() < tol(w64.grad)
    assert (g_sh.cpu().double() - sh64.grad).abs().max() < tol(sh64.grad)
    gx_ref = torch.zeros(NT, dx, dtype=torch.float64).index_add_(0, src.long(), g_xe.cpu().double())
    assert (gx_ref - x64.grad[:, to_x]).abs().max() < tol(x64.grad)
    assert torch.equal(g_w, g_w2) and torch.equal(g_xe, g_xe2)
    gv_ref = torch.einsum('ei,eia->ea', sh64.grad, dsh.double())
    assert (g_vec.cpu().double() - gv_ref).abs().max() < tol(gv_ref)
    lib.snet_conv_plan_destroy(plan)


# --------------------------------------------------------------------------- #
# HIP kernels vs outputs of the reference's own pure-torch modules
# (tests/golden/ref_torch_modules.npz, oracle/tools/make_golden_torch_modules.py)
# --------------------------------------------------------------------------- #
def _ref_modules():
    from helpers import GOLDEN
    return np.load(f'{GOLDEN}/ref_torch_modules.npz')


@pytest.mark.parametrize('tag', ['rc5', 'rc6', 'rc4'])
def test_edge_embedding_vs_reference_modules(tag):
    """snet_edge_embed_fwd == BesselBasis * {PolynomialCutoff, XPLORCutoff} of the reference
    (edge_embedding.py:101-103,125-132,150-160) on its own fp32 outputs, incl. r_on and r_cut"""
    L, lib = _lib()
    dev = 'cuda:0'
    d = _ref_modules()
    rc, r_on, p = (float(v) for v in d[f'{tag}_params'])
    r = torch.tensor(d[f'{tag}_r'], dtype=torch.float32)
    E, nb = r.numel(), 8
    u = torch.nn.functional.normalize(torch.randn(E, 3, generator=torch.Generator().manual_seed(1)), dim=1)
    vd = (u * r.unsqueeze(1)).to(dev)
    r_act = vd.norm(dim=1).cpu().double()        # the length the kernel actually sees
    cf = (C.c_float * nb)(*[float(v) for v in d[f'{tag}_coeffs_f32']])
    for kind, ref in ((0, d[f'{tag}_poly_f64']), (1, d[f'{tag}_xplor_f64'])):
        P = L.EdgeParams(rc, nb, kind, int(p), r_on, 1, 0)
        emb, sh, dsh = torch.empty(E, nb, device=dev), torch.empty(E, 4, device=dev), torch.empty(E, 4, 3, device=dev)
        L.check(lib.snet_edge_embed_fwd(C.byref(P), cf, _p(vd), E, _p(emb), _p(sh), _p(dsh), None))
        torch.cuda.synchronize()
        want = d[f'{tag}_bessel_f64'] * ref[:, None]
        # |d(emb)/dr| <= ~10 / A: allow for the fp32 rounding of r itself
        slack = 2e-6 + 12.0 * (r_act - torch.tensor(d[f'{tag}_r'])).abs().max().item()
        assert np.abs(emb.cpu().double().numpy() - want).max() <= slack, (tag, kind)


def test_edge_force_vs_reference_module():
    """snet_edge_force == ForceStressOutputFromEdge.forward (force_output.py:171-230) on the reference's own
    inputs / outputs: forces, per-atom virial (xx,yy,zz,xy,yz,zx at edge_index[1]), total virial"""
    L, lib = _lib()
    dev = 'cuda:0'
    d = _ref_modules()
    ei = torch.tensor(d['fs_edge_index'])
    n = d['fs_force_f32'].shape[0]
    order = torch.sort(ei[0], stable=True).indices       # the kernels want edges sorted by center (edge_index[0])
    ei = ei[:, order]
    g = torch.tensor(d['fs_gij_f32'])[order].contiguous().to(dev)
    rij = torch.tensor(d['fs_rij_f32'])[order].contiguous().to(dev)
    E = ei.shape[1]
    row_ptr = torch.zeros(n + 1, dtype=torch.int64)
    row_ptr[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n), 0)
    col_ptr = torch.zeros(n + 1, dtype=torch.int64)
    col_ptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n), 0)
    eperm = torch.sort(ei[1], stable=True).indices
    rp, cp, ep = (t.to(torch.int32).to(dev) for t in (row_ptr, col_ptr, eperm))
    F = torch.empty(n, 3, device=dev)
    va = torch.empty(n, 6, device=dev)
    vt = torch.empty(6, dtype=torch.float64, device=dev)
    L.check(lib.snet_edge_force(_p(g), _p(rij), _p(rp), _p(cp), _p(ep), n, E, _p(F), _p(va), _p(vt), None))
    torch.cuda.synchronize()
    assert np.abs(F.cpu().numpy() - d['fs_force_f64']).max() <= 1e-5 * np.abs(d['fs_force_f64']).max()
    assert np.abs(va.cpu().numpy() - d['fs_atomic_virial_f64']).max() <= 1e-5 * np.abs(d['fs_atomic_virial_f64']).max()
    stress = vt.cpu().numpy() / d['fs_volume'][0]
    assert np.abs(stress - d['fs_stress_f64']).max() <= 1e-5 * np.abs(d['fs_stress_f64']).max()


def test_rescale_reduce_vs_reference_modules():
    """snet_rescale_reduce == Rescale / SpeciesWiseRescale / ModalWiseRescale.forward + AtomReduce
    (scale.py:53-56,155-162,341-363; linear.py:127-141)"""
    L, lib = _lib()
    dev = 'cuda:0'
    d = _ref_modules()
    e = torch.tensor(d['rs_in']).to(dev)
    types = torch.tensor(d['rs_types']).to(torch.int32).to(dev)
    n = e.shape[0]

    def run(scale, shift):
        sc = torch.tensor(np.asarray(scale, np.float32)).reshape(-1).to(dev)
        sh = torch.tensor(np.asarray(shift, np.float32)).reshape(-1).to(dev)
        ea = torch.empty(n, device=dev)
        tot = torch.empty(1, dtype=torch.float64, device=dev)
        L.check(lib.snet_rescale_reduce(_p(e), _p(types), _p(sc), _p(sh), sc.numel(), n, _p(ea), _p(tot), None))
        torch.cuda.synchronize()
        return ea.cpu().numpy(), tot.item()
    sh0, sc0 = d['rs_global_params']
    ea, _ = run([sc0], [sh0])
    assert np.abs(ea - d['rs_global'][:, 0]).max() <= 1e-6
    ea, tot = run(d['rs_species_scale'], d['rs_species_shift'])
    assert np.abs(ea - d['rs_species'][:, 0]).max() <= 1e-6
    assert abs(tot - float(d['reduce_total'][0])) <= 2e-5
    # modal-wise: the engine selects the fidelity channel's row at load time (model_spec.rescale_vectors)
    sm, cm = d['rs_modal_shift'], d['rs_modal_scale']
    for modal in range(sm.shape[0]):
        for tag, shift, scale in (('mm', sm[modal], cm[modal]), ('ms', sm[modal], d['rs_species_scale']),
                                  ('sm', d['rs_species_shift'], cm[modal])):
            ea, _ = run(scale, shift)
            assert np.abs(ea - d[f'rs_modal_{tag}_{modal}'][:, 0]).max() <= 1e-6, (tag, modal)


def test_edge_vectors_from_positions():
    """snet_edge_vectors: r_j - r_i + image offset, subtracted in fp64 (positions near 125 A have an fp32 ulp of 8e-6 A:
    an fp32 subtraction would miss the reference hosts' edge vectors by 1e-5 A)"""
    L, lib = _lib()
    dev = 'cuda:0'
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    pos, cell = diamond_cubic(5.431, (3, 3, 3), 0.05, 1)
    pos = pos + 110.0          # far from the origin
    ei, ev, S = neighbor_list(pos, cell, [True] * 3, 5.0)
    shift = S.astype(np.float64) @ cell
    assert np.abs(pos[ei[1]] - pos[ei[0]] + shift - ev).max() < 1e-9
    pd = torch.from_numpy(pos).to(dev)
    c, s_ = (torch.from_numpy(ei[k]).to(dev, torch.int32) for k in (0, 1))
    sd = torch.from_numpy(np.ascontiguousarray(shift)).to(dev)
    out = torch.empty(ei.shape[1], 3, device=dev)
    L.check(lib.snet_edge_vectors(_p(pd), _p(c), _p(s_), _p(sd), ei.shape[1], _p(out), None))
    torch.cuda.synchronize()
    want = torch.from_numpy(ev).to(torch.float32)
    assert (out.cpu() - want).abs().max() <= 2.4e-7 * 5.0        # one fp32 ulp of a cutoff-length vector
    assert (out.cpu() == want).float().mean() > 0.99
    naive = (pd.float()[s_.long()] - pd.float()[c.long()] + sd.float()).cpu()
    assert (naive - want).abs().max() > 3e-6                      # what the fp64 subtraction avoids


def test_edges_by_source_and_transposed_scalar_convolution():
    """snet_edges_by_source (destination atom / radial row of every edge in source-grouped order) and the last layer's
    source-row gradient as a forward convolution of the transposed product (snet_conv_plan_transposed):
    == g_xe rows of the fused reverse kernel summed per source atom, ragged degrees and atoms without edges included"""
    L, lib = _lib()
    dev = 'cuda:0'
    from sevennet_amd.engine import build_graph
    from sevennet_amd.model_spec import build_model_spec, transposed_scalar_conv
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    pos, cell = diamond_cubic(5.431, (3, 2, 2), 0.07, 3)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, 4.4)
    keep = np.ones(ei.shape[1], bool)
    keep[(ei[0] == 5) | (ei[1] == 7)] = False   # an atom without in-edges, one that is nobody's source
    keep[::11] = False                          # ragged degrees, one-directional pairs
    ei, ev = ei[:, keep], ev[keep]
    g = build_graph(np.zeros(len(pos), np.int64), ei, ev, device=dev, share_pairs=False)
    E, N = g.n_edges, g.n_local
    ct, wt = torch.empty_like(g.eperm), torch.empty_like(g.eperm)
    L.check(lib.snet_edges_by_source(_p(g.row_ptr), N, _p(g.eperm), None, E, _p(ct), _p(wt), None))
    torch.cuda.synchronize()
    assert torch.equal(ct, g.center[g.eperm.long()]) and torch.equal(wt, g.eperm)
    w_row = torch.randint(0, E, (E,), device=dev, dtype=torch.int32)
    L.check(lib.snet_edges_by_source(_p(g.row_ptr), N, _p(g.eperm), _p(w_row), E, _p(ct), _p(wt), None))
    torch.cuda.synchronize()
    assert torch.equal(wt, w_row[g.eperm.long()])
    # the transposed shape of SevenNet-0's last layer reproduces sum_e w_e T Y_e g_out[center(e)] per source atom
    conv = build_model_spec(sevennet_0_config()).layers[-1].conv
    spec_t, kappa = transposed_scalar_conv(conv)
    tag = C.create_string_buffer(13)
    plan, plan_t = C.c_void_p(), C.c_void_p()
    L.check(lib.snet_conv_plan_create(conv.tag.encode(), C.byref(plan)))
    col = np.zeros(conv.weight_numel, np.float32)
    L.check(lib.snet_conv_plan_transposed(plan, tag, col.ctypes.data_as(C.c_void_p), None, 0, None))
    assert tag.value.decode() == spec_t.tag
    L.check(lib.snet_conv_plan_create(spec_t.tag.encode(), C.byref(plan_t)))
    dx, dout, wn, nsh = conv.irreps_x.dim, conv.irreps_out.dim, conv.weight_numel, conv.irreps_sh.dim
    gen = torch.Generator(device='cpu').manual_seed(5)
    x, g_out = torch.randn(N, dx, generator=gen).to(dev), torch.randn(N, dout, generator=gen).to(dev)
    sh, w = torch.randn(E, nsh, generator=gen).to(dev), torch.randn(E, wn, generator=gen).to(dev)
    dsh = torch.zeros(E, nsh, 3, device=dev)
    g_w, g_xe, g_vec = torch.empty(E, wn, device=dev), torch.empty(E, dx, device=dev), torch.zeros(E, 3, device=dev)
    L.check(lib.snet_conv_bwd_edge_vec(plan, _p(x), _p(sh), _p(dsh), _p(w), None, _p(g.row_ptr), _p(g.src), N, 0.25, _p(g_out),
                                       _p(g_w), _p(g_xe), _p(g_vec), None))
    want = torch.zeros(N, dx, device=dev).index_add_(0, g.src.long(), g_xe)
    ep = g.eperm.long()
    sh_t, w_t = sh[ep].contiguous(), (w * torch.from_numpy(col).to(dev))[ep].contiguous()
    got = torch.zeros(N, dx, device=dev)
    L.check(lib.snet_conv_fwd(plan_t, _p(g_out), _p(sh_t), _p(w_t), None, _p(g.col_ptr), _p(ct), N, 0.25, _p(got), None))
    torch.cuda.synchronize()
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    assert want[7].abs().max().item() == 0.0 and got[7].abs().max().item() == 0.0
    lib.snet_conv_plan_destroy(plan)
    lib.snet_conv_plan_destroy(plan_t)


@pytest.mark.gpu
@pytest.mark.parametrize('nb,acts,E', [(8, (0, 0, 0, 0, 0), 1000), (8, (0, 1, 4), 333), (6, (0, 0), 130), (8, (0,), 5)])
def test_radial_mlp_hidden_layers_one_launch(nb, acts, E):
    """snet_radial_mlp_hidden_fwd_layers: the hidden activations of several interaction layers' radial MLPs from ONE launch ==
    one snet_radial_mlp_hidden_fwd per layer, bit for bit, and == act(act(emb W0) c W1) c in fp64 (nn/convolution.py:124 feeds the
    same edge embedding to every layer's weight_nn).  Covers the compile-time (silu, 8 basis functions) and the run-time
    (mixed activations, 6 basis functions) instantiations and a ragged edge count."""
    L, lib = _lib()
    dev = 'cuda:0'
    rng = np.random.default_rng(nb * 100 + E)
    cst_of = {0: 1.6791767923989418, 1: 1.5937334472592695, 4: 1.6822012}
    emb = torch.from_numpy(rng.normal(0, 0.6, (E, nb)).astype(np.float32)).to(dev)
    plans, Ws = [], []
    for a in acts:
        W0 = (rng.normal(0, 1, (nb, 64)) / np.sqrt(nb)).astype(np.float32)
        W1 = (rng.normal(0, 1, (64, 64)) / 8).astype(np.float32)
        W2 = (rng.normal(0, 1, (64, 32)) / 8).astype(np.float32)
        fp = lambda t: t.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        mlp = C.c_void_p()
        L.check(lib.snet_radial_mlp_plan_create(nb, 64, 64, 32, fp(W0), fp(W1), fp(W2), a, cst_of[a], 1, C.byref(mlp)))
        plans.append(mlp)
        Ws.append((W0, W1))
    n = len(acts)
    one = [torch.full((E, 64), float('nan'), device=dev) for _ in acts]
    for p, o in zip(plans, one):
        L.check(lib.snet_radial_mlp_hidden_fwd(p, _p(emb), E, _p(o), None))
    many = [torch.full((E, 64), float('nan'), device=dev) for _ in acts]
    L.check(lib.snet_radial_mlp_hidden_fwd_layers((C.c_void_p * n)(*plans), n, _p(emb), E, (C.c_void_p * n)(*[o.data_ptr() for o in many]), None))
    torch.cuda.synchronize()

    def act(z, a):
        if a == 0:
            return z / (1 + np.exp(-z))
        if a == 1:
            return np.tanh(z)
        return np.logaddexp(z, 0) - np.log(2.0)
    e64 = emb.cpu().numpy().astype(np.float64)
    for a, (W0, W1), o, mny in zip(acts, Ws, one, many):
        assert torch.equal(o, mny)
        ref = act(act(e64 @ W0.astype(np.float64), a) * cst_of[a] @ W1.astype(np.float64), a) * cst_of[a]
        err = np.abs(mny.cpu().numpy() - ref).max()
        assert err < 3e-6 * max(1.0, np.abs(ref).max()), (a, err)     # fp32 rounding class (bf16 x6 products, hardware exp2 / rcp for silu)
    with pytest.raises(RuntimeError, match='1 .. 8 layers'):
        L.check(lib.snet_radial_mlp_hidden_fwd_layers((C.c_void_p * n)(*plans), 0, _p(emb), E, (C.c_void_p * n)(*[o.data_ptr() for o in many]), None))
    if nb == 8 and n > 1:   # layers with different basis counts cannot share an edge embedding
        other = C.c_void_p()
        W0 = np.zeros((6, 64), np.float32)
        L.check(lib.snet_radial_mlp_plan_create(6, 64, 64, 32, fp(W0), fp(Ws[0][1]), fp(np.zeros((64, 32), np.float32)), 0, cst_of[0], 1, C.byref(other)))
        with pytest.raises(RuntimeError, match='same n_basis'):
            L.check(lib.snet_radial_mlp_hidden_fwd_layers((C.c_void_p * 2)(plans[0], other), 2, _p(emb), E, (C.c_void_p * 2)(many[0].data_ptr(), many[1].data_ptr()), None))
        lib.snet_radial_mlp_plan_destroy(other)
    for p in plans:
        lib.snet_radial_mlp_plan_destroy(p)


@pytest.mark.gpu
def test_row_absmax_multi_equals_per_matrix_calls():
    """snet_row_absmax_multi: row maxima of several matrices of different shapes from one launch (ragged row counts, a row
    length that is not a multiple of 4, an empty matrix in the middle) == torch, exactly (a maximum has no rounding)."""
    L, lib = _lib()
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(5)
    shapes = [(1001, 480), (3, 128), (0, 64), (517, 30), (64, 1152)]
    xs = [torch.randn(r, d, generator=g).to(dev) * (10.0 ** (i - 2)) for i, (r, d) in enumerate(shapes)]
    outs = [torch.full((r,), float('nan'), device=dev) for r, _ in shapes]
    n = len(shapes)
    L.check(lib.snet_row_absmax_multi((C.c_void_p * n)(*[x.data_ptr() if x.numel() else None for x in xs]), (C.c_int64 * n)(*[r for r, _ in shapes]),
                                      (C.c_int32 * n)(*[d for _, d in shapes]), (C.c_void_p * n)(*[o.data_ptr() if o.numel() else None for o in outs]), n, None))
    torch.cuda.synchronize()
    for x, o in zip(xs, outs):
        if x.numel():
            assert torch.equal(o, x.abs().amax(1))
    with pytest.raises(RuntimeError, match='1 .. 8 matrices'):
        L.check(lib.snet_row_absmax_multi((C.c_void_p * n)(), (C.c_int64 * n)(), (C.c_int32 * n)(), (C.c_void_p * n)(), 9, None))
