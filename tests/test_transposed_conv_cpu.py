"""CPU: the transposed scalar convolution (model_spec.transposed_scalar_conv) against autograd of the oracle's tensor product.

For a convolution whose outputs are all scalars, dL/dx[j] = sum over the edges that have j as their source of a uvu product
of the DESTINATION's output gradient with the same spherical harmonics and radial weights, one constant per path.  The engine
runs that product on its forward kernel; here the identity and the constants are checked in fp64 with the
oracle's own tensor product (oracle/model.py::tp_uvu, e3nn semantics) and torch.autograd, no HIP involved."""
import numpy as np
import pytest
import torch

from oracle.e3 import Irreps as OIrreps
from oracle.model import conv_instructions, tp_uvu


@pytest.mark.parametrize('irreps_x,lmax', [('8x0e+4x1e+4x2e', 2), ('4x0e+2x1e+2x2e+2x3e', 3), ('4x0e+3x1o+2x2e+2x3o', 3)])
def test_transposed_product_equals_autograd_of_the_forward_product(irreps_x, lmax):
    from sevennet_amd.irreps import Irreps
    from sevennet_amd.model_spec import make_conv, transposed_scalar_conv
    parity = 'o' in irreps_x
    sh_str = '+'.join(f'1x{l}{"eo"[l % 2] if parity else "e"}' for l in range(lmax + 1))
    ix, ish = Irreps(irreps_x), Irreps(sh_str)
    n_scalar = sum(mul for mul, _, _ in ix)
    conv = make_conv(ix, ish, Irreps(f'{n_scalar}x0e'), sort_by_out=True)
    assert all(p.l3 == 0 for p in conv.paths) and len(conv.paths) == len(ix)
    spec_t, kappa = transposed_scalar_conv(conv)
    for p, k in zip(conv.paths, kappa):
        assert k == pytest.approx((2 * p.l1 + 1) ** -0.5, rel=1e-12)

    # ---- forward product in the oracle's (e3nn) terms
    ox, osh = OIrreps(irreps_x), OIrreps(sh_str)
    mid, ins, wn = conv_instructions(ox, osh, OIrreps(f'{n_scalar}x0e'), sort_by_out=True)
    assert wn == conv.weight_numel and len(ins) == len(conv.paths)
    g = torch.Generator().manual_seed(3)
    n, E = 7, 40
    src, dst = torch.randint(0, n, (E,), generator=g), torch.randint(0, n, (E,), generator=g)
    x = torch.randn(n, ox.dim, generator=g, dtype=torch.float64, requires_grad=True)
    Y = torch.randn(E, osh.dim, generator=g, dtype=torch.float64)
    w = torch.randn(E, wn, generator=g, dtype=torch.float64)
    G = torch.randn(n, mid.dim, generator=g, dtype=torch.float64)          # dL/d(out), scalars only
    msg = tp_uvu(x[src], Y, w, ox, osh, mid, ins)
    out = torch.zeros(n, mid.dim, dtype=torch.float64).index_add_(0, dst, msg)
    (g_x,) = torch.autograd.grad((out * G).sum(), x)

    # ---- the same gradient as a FORWARD product of the transposed shape over the edges grouped by source:
    # "x" = G[dst] (one scalar block per path, in the order of the forward product's output blocks), weights = w * kappa
    # (same columns: the transposed paths keep the forward paths' order), output = one (mul, l1) block per path
    order = sorted(range(len(conv.paths)), key=lambda k: (conv.paths[k].out_off, conv.paths[k].out_ch))
    g_irreps = OIrreps([(conv.paths[k].mul, (0, 1)) for k in order])
    assert g_irreps.dim == mid.dim
    block_of = {k: i for i, k in enumerate(order)}
    t_mid = OIrreps([(p.mul, ox[p.i_x][1]) for p in conv.paths])           # path k writes block k = the x block it read
    t_ins = [(block_of[k], p.i_sh, k) for k, p in enumerate(conv.paths)]
    col = torch.ones(wn, dtype=torch.float64)
    for p, kp in zip(conv.paths, kappa):
        col[p.w_off:p.w_off + p.mul] = kp
    # the engine's transposed ConvSpec says the same thing
    assert [(q.i_x, q.i_sh, q.l1, q.l3, q.w_off) for q in spec_t.paths] == \
           [(block_of[k], p.i_sh, 0, p.l1, p.w_off) for k, p in enumerate(conv.paths)]
    msg_t = tp_uvu(G[dst], Y, w * col, g_irreps, osh, t_mid, t_ins)
    got_blocks = torch.zeros(n, t_mid.dim, dtype=torch.float64).index_add_(0, src, msg_t)
    # blocks of t_mid are in path order, x blocks in irreps order: put them back
    got = torch.zeros_like(g_x)
    sl_x, sl_t = ox.slices(), t_mid.slices()
    for k, p in enumerate(conv.paths):
        got[:, sl_x[p.i_x]] = got_blocks[:, sl_t[k]]
    assert torch.allclose(got, g_x, rtol=1e-12, atol=1e-12), float((got - g_x).abs().max())
    assert float(g_x.abs().max()) > 0.1


def test_shapes_with_a_non_scalar_output_have_no_transposed_form():
    from sevennet_amd.model_spec import build_model_spec, sevennet_0_config, transposed_scalar_conv
    layers = build_model_spec(sevennet_0_config()).layers
    assert transposed_scalar_conv(layers[1].conv) is None and transposed_scalar_conv(layers[0].conv) is None
    assert transposed_scalar_conv(layers[-1].conv) is not None
