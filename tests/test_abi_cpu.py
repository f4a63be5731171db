"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/snet_hip.h declares; host-side model description logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, 'include', 'snet_hip.h')) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(snet_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from sevennet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from sevennet_amd.build import build
        build(verbose=False)
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in snet_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.snet_abi_version() == _lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, 'include', 'snet_hip.h')).read()
    assert int(re.search(r'#define SNET_ABI_VERSION (\d+)', hdr).group(1)) == _lib.ABI_VERSION   # header, library, binding in step


def test_library_exports_the_reference_d3_binding():
    """include/snet_d3_ref.h: the ten `pair_*` functions of the reference's D3 library (pair_d3_for_ase.cu:2034-2082), and
    the parameter blob they read sits next to the library"""
    from sevennet_amd import _lib
    with open(os.path.join(ROOT, 'include', 'snet_d3_ref.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    names = sorted(set(re.findall(r'\b(pair_[a-z_]+)\s*\(', text)))
    assert names == sorted(['pair_init', 'pair_set_atom', 'pair_set_domain', 'pair_run_settings', 'pair_run_coeff',
                            'pair_run_compute', 'pair_get_energy', 'pair_get_force', 'pair_get_stress', 'pair_fin',
                            'pair_failed'])   # (pair_failed: the one extension, reads the sticky failure flag)
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    blob = os.path.join(os.path.dirname(_lib.LIB_PATH), 'data', 'd3_params.bin')
    assert os.path.exists(blob)
    raw = open(blob, 'rb').read()
    z = np.load(os.path.join(os.path.dirname(_lib.LIB_PATH), 'data', 'd3_params.npz'))
    n_c6 = int(np.frombuffer(raw[8:16], '<i8')[0])
    assert raw[:8] == b'SNETD3P1' and n_c6 == z['c6ab'].shape[0]
    body = np.frombuffer(raw[16:16 + 8 * (94 * 94 + 5 * n_c6 + 188)], '<f8')
    assert np.array_equal(body[:94 * 94], z['r0ab'].ravel()) and np.array_equal(body[-94:], z['rcov'])
    # without a GPU the shims fail loudly (sticky error, zero results), they do not crash
    lib.pair_init.restype = C.c_void_p
    lib.pair_get_energy.restype = C.c_double
    lib.pair_get_energy.argtypes = [C.c_void_p]
    lib.pair_fin.argtypes = [C.c_void_p]
    lib.pair_get_force.restype = C.c_void_p
    lib.pair_get_force.argtypes = [C.c_void_p]
    lib.pair_get_stress.restype = C.c_void_p
    lib.pair_get_stress.argtypes = [C.c_void_p]
    lib.pair_failed.argtypes = [C.c_void_p]
    p = lib.pair_init()
    assert p
    # a failed handle never hands out results a caller could mistake for "zero dispersion" (ADVICE r4): NaN energy, NULL arrays
    if lib.pair_failed(p):
        assert np.isnan(lib.pair_get_energy(p)) and lib.pair_get_force(p) is None and lib.pair_get_stress(p) is None
    else:   # (a GPU is present: an unknown functional name is the failure that must stick)
        lib.pair_run_settings.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_char_p, C.c_char_p]
        lib.pair_run_settings(p, 9000.0, 1600.0, b'damp_bj', b'no-such-functional')
        assert lib.pair_failed(p) and np.isnan(lib.pair_get_energy(p)) and lib.pair_get_stress(p) is None
    lib.pair_fin(p)


def test_every_registered_model_shape_is_compiled():
    from sevennet_amd import _lib
    from sevennet_amd.shapes import aot_conv_specs
    tags = set(_lib.compiled_conv_tags())
    for tag in aot_conv_specs():
        assert tag in tags


def test_unknown_shape_fails_loudly():
    import ctypes as C
    from sevennet_amd import _lib
    lib = _lib.load()
    p = C.c_void_p()
    rc = lib.snet_conv_plan_create(b'0123456789ab', C.byref(p))
    assert rc != 0 and b'not compiled' in lib.snet_last_error()


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = unit_test_config()
    with pytest.raises(RuntimeError):
        HipForceEngine(cfg, random_state_dict(cfg))


def test_layout_index_roundtrip_and_gate_layout():
    from sevennet_amd.irreps import Irreps, irmul_to_mulir_index, mulir_to_irmul_index
    from sevennet_amd.model_spec import make_gate
    irr = Irreps('4x0o+12x0e+4x1o+4x1e')
    f, b = mulir_to_irmul_index(irr), irmul_to_mulir_index(irr)
    x = np.arange(irr.dim)
    assert (x[f][b] == x).all()
    # e3nn Gate input layout verified on the reference's deployed model:
    # scalars 4x0o+4x0e, gates 8x0e, gated 4x1o+4x1e -> [0o | 0e scalars | 0e gates | 1o | 1e]
    g = make_gate(Irreps('4x0o+4x0e+4x1o+4x1e'), {'e': 'silu', 'o': 'tanh'}, {'e': 'silu', 'o': 'tanh'})
    assert str(g.irreps_in) == '4x0o+12x0e+4x1o+4x1e'
    segs = {(s.kind, s.in_off): s for s in g.segs}
    assert segs[(0, 0)].act == 1 and segs[(0, 4)].act == 0           # tanh on odd scalars, silu on even
    assert segs[(1, 16)].gate_off == 8 and segs[(1, 28)].gate_off == 12


def test_neighbor_list_edge_count_pins():
    """tests/unit_tests/test_data.py:48 of the reference (cutoff 4.0): bulk NaCl 36, H2O 6, H 0, Cu 18"""
    from sevennet_amd.neighbor import neighbor_list
    a = 5.63
    cell = np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]])
    assert neighbor_list(np.array([[0, 0, 0], [a / 2] * 3]), cell, [1, 1, 1], 4.0)[0].shape[1] == 36
    a = 3.61
    cell = np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]])
    ei, ev, S = neighbor_list(np.zeros((1, 3)), cell, [1, 1, 1], 4.0)
    assert ei.shape[1] == 18 and (ei[0] == ei[1]).all() and (np.abs(S).sum(1) > 0).all()
    h2o = np.array([[0, 0, 0.119262], [0, 0.763239, -0.477047], [0, -0.763239, -0.477047]])
    assert neighbor_list(h2o, np.zeros((3, 3)), [0, 0, 0], 4.0)[0].shape[1] == 6
    assert neighbor_list(np.zeros((1, 3)), np.zeros((3, 3)), [0, 0, 0], 4.0)[0].shape[1] == 0


def test_gemm_split_pack_is_an_exact_three_term_bf16_sum():
    """host packing of weights for the split-precision GEMM: every value = hi + mid + lo (bf16 each)
    to within 2^-24 relative, laid out as MFMA B fragments with zero padding"""
    import ctypes as C
    import numpy as np
    from sevennet_amd import _lib
    lib = _lib.load()
    K, N = 21, 37
    B = np.random.default_rng(0).standard_normal((K, N)).astype(np.float32)
    size = lib.snet_gemm_split_size(K, N)
    assert size == 2 * 2 * 3 * 64 * 16
    buf = np.zeros(size, np.uint8)
    assert lib.snet_gemm_split_pack(C.c_void_p(B.ctypes.data), K, N, C.c_void_p(buf.ctypes.data)) == 0
    h = buf.view(np.uint16).reshape(2, 2, 3, 64, 8)   # [tile][q][term][lane][i]
    f = (h.astype(np.uint32) << 16).view(np.float32)
    total = f.sum(axis=2, dtype=np.float64)            # [tile][q][lane][i]
    for t in range(2):
        for q in range(2):
            for lane in range(64):
                for i in range(8):
                    k, n = 16 * q + 8 * (lane >> 5) + i, 32 * t + (lane & 31)
                    want = float(B[k, n]) if (k < K and n < N) else 0.0
                    assert abs(total[t, q, lane, i] - want) <= 2.0 ** -22 * abs(want)
    assert lib.snet_gemm_split_pack(None, K, N, C.c_void_p(buf.ctypes.data)) != 0


def test_md_nodes_numbering_host_only():
    """snet_md_nodes (pure host code): owned atoms in ilist order, then one node per ghost identity owned
    elsewhere in first-seen atom order -- the numbering snet_md_compute uses (tag_to_graph_idx of
    pair_e3gnn_parallel.cpp:262-287), which a pair style needs before an evaluation to lay out its halo"""
    import ctypes as C
    import numpy as np
    from sevennet_amd import _lib
    lib = _lib.load()
    tag = np.array([5, 6, 7, 8, 9, 5, 9, 11, 6, 11, 12], np.int32)   # 4 owned (ilist), images of owned 5 / 6, ghosts 9, 11, 12
    ilist = np.array([2, 0, 3, 1], np.int32)
    out = np.full(len(tag), -1, np.int32)
    n = C.c_int64()
    P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(lib.snet_md_nodes(4, P(ilist), len(tag), P(tag), 4, 1, P(out), C.byref(n)))
    assert n.value == 7 and out[:7].tolist() == [2, 0, 3, 1, 4, 7, 10]   # ghosts: tag 9 (atom 4), 11 (atom 7), 12 (atom 10)
    _lib.check(lib.snet_md_nodes(4, P(ilist), len(tag), P(tag), 4, 0, P(out), C.byref(n)))
    assert n.value == 4
    tag8 = tag.astype(np.int64)
    _lib.check(lib.snet_md_nodes(4, P(ilist), len(tag), P(tag8), 8, 1, P(out), C.byref(n)))
    assert n.value == 7 and out[4:7].tolist() == [4, 7, 10]
    with pytest.raises(RuntimeError):
        _lib.check(lib.snet_md_nodes(0, P(ilist), len(tag), P(tag), 4, 1, P(out), C.byref(n)))


def test_unlisted_shape_is_compiled_on_demand(tmp_path, monkeypatch):
    """b1: the reference's hook builds `convolution_cls(**kwargs)` for any irreps (convolution.py:237-247).  A shape that
    is not in sevennet_amd/shapes.py is generated, cross-compiled by hipcc into a shape library under $SNET_JIT_CACHE
    and registered with the running libsnet_hip.so (snet_conv_register_library); the second request is a cache hit."""
    from sevennet_amd import _lib, jit
    from sevennet_amd.model_spec import build_model_spec
    from sevennet_amd.shapes import aot_conv_specs, unit_test_config
    monkeypatch.setenv('SNET_JIT_CACHE', str(tmp_path))
    spec = build_model_spec(unit_test_config(channel=16, lmax=1)).layers[1].conv
    assert spec.tag not in aot_conv_specs()
    lib = _lib.load()
    if spec.tag not in _lib.compiled_conv_tags():
        plan = C.c_void_p()
        assert lib.snet_conv_plan_create(spec.tag.encode(), C.byref(plan)) != 0
    assert jit.ensure_conv_shape(spec) == spec.tag
    assert spec.tag in _lib.compiled_conv_tags()
    plan = C.c_void_p()
    _lib.check(lib.snet_conv_plan_create(spec.tag.encode(), C.byref(plan)), 'snet_conv_plan_create')
    assert lib.snet_conv_fused_available(plan) == 1      # channel multiplicities % 16 == 0: fused kernels too
    lib.snet_conv_plan_destroy(plan)
    files = [f for f in os.listdir(tmp_path) if f.endswith('.so')]
    assert len(files) == 1 and files[0].startswith(spec.tag)
    assert jit.compile_shape(spec) == os.path.join(str(tmp_path), files[0])   # cache hit: same file
    with pytest.raises(RuntimeError, match='dlopen'):
        _lib.check(lib.snet_conv_register_library(str(tmp_path / 'missing.so').encode()), 'snet_conv_register_library')


def test_scalar_output_shapes_carry_their_transposed_form():
    """last-layer shapes (paths (l, l -> 0) only) name the transposed product + W2 column factors; others do not"""
    from sevennet_amd.model_spec import (build_model_spec, sevennet_0_config, sevennet_l3i5_config, sevennet_mf_ompa_config,
                                         transposed_scalar_conv)
    from sevennet_amd import _lib
    lib = _lib.load()
    for cfg in (sevennet_0_config(), sevennet_l3i5_config(), sevennet_mf_ompa_config()):
        layers = build_model_spec(cfg).layers
        for ls in (layers[1], layers[-1]):
            plan = C.c_void_p()
            _lib.check(lib.snet_conv_plan_create(ls.conv.tag.encode(), C.byref(plan)), 'snet_conv_plan_create')
            tag, col = C.create_string_buffer(13), np.zeros(ls.conv.weight_numel, np.float32)
            dead, nd = (C.c_int32 * 32)(), C.c_int32(-1)
            _lib.check(lib.snet_conv_plan_transposed(plan, tag, col.ctypes.data_as(C.c_void_p), C.cast(dead, C.c_void_p), 32,
                                                     C.byref(nd)), 'snet_conv_plan_transposed')
            tr = transposed_scalar_conv(ls.conv)
            if ls is layers[1]:
                assert tr is None and tag.value == b'' and nd.value == 0
                continue
            spec_t, kappa = tr
            assert tag.value.decode() == spec_t.tag and spec_t.tag in _lib.compiled_conv_tags()
            # the transposed product swaps the roles of x and out; same weight columns
            assert spec_t.irreps_out.dim == ls.conv.irreps_x.dim and spec_t.irreps_x.dim == ls.conv.irreps_out.dim
            assert spec_t.weight_numel == ls.conv.weight_numel
            for p, k in zip(ls.conv.paths, kappa):
                l = ls.conv.irreps_x[p.i_x][1]
                assert k == pytest.approx((2 * l + 1) ** -0.5, rel=1e-12)
                assert np.all(col[p.w_off:p.w_off + p.mul] == np.float32(k))
            fed = {p.i_x for p in ls.conv.paths}
            want = [(off, m * (2 * l + 1)) for i, (off, (m, l, _)) in
                    enumerate(zip(ls.conv.irreps_x.offsets(), ls.conv.irreps_x)) if i not in fed]
            assert [(dead[2 * i], dead[2 * i + 1]) for i in range(nd.value)] == want
            lib.snet_conv_plan_destroy(plan)


def test_row_map_reciprocal_is_exact_in_its_stated_range():
    """csrc/snet_gemm.hip::RowMap divides (m0 + off) by d with a 16-bit reciprocal, m0 < d, off < 256, and the entry points
    refuse d > 150: the identity ((x * (65536 / d + 1)) >> 16) == x / d must hold for every such x"""
    src = open(os.path.join(ROOT, 'sevennet_amd', 'csrc', 'snet_gemm.hip')).read()
    assert 'inv(65536u / (uint32_t)d_ + 1u)' in src and '(x * inv) >> 16' in src and 'p.d <= 150' in src and 'd <= 150' in src
    for d in range(1, 151):
        inv = 65536 // d + 1
        assert all(((x * inv) >> 16) == x // d for x in range(d + 256)), d


def test_fused_plugin_module_host_side():
    """b1, the whole-convolution module (sevennet_amd.conv_plugin.HipFusedIrrepsConvolution) without a GPU: what a reference
    maintainer relies on before any kernel runs -- parameter names and shapes of the reference's IrrepsConvolution (so that a reference
    state_dict loads), construction from the keyword tables patch_convolution receives, refusal of shapes without fused kernels and of
    radial networks the fused tail does not cover, and a loud error instead of a CPU fallback"""
    import torch
    from sevennet_amd.conv_plugin import HipFusedIrrepsConvolution, patch_convolution
    from sevennet_amd.model_spec import build_model_spec, sevennet_0_config
    from sevennet_amd.shapes import unit_test_config
    spec = build_model_spec(sevennet_0_config()).layers[1].conv
    # (sevennet_0_config is a pre-0.11 model: e3nn instruction order; sort_by_out=True is the >= 0.11 order, another weight-column
    # order and therefore another compiled shape)
    m = HipFusedIrrepsConvolution(str(spec.irreps_x), str(spec.irreps_sh), str(spec.irreps_out), [8, 64, 64], 'silu', 28.0, sort_by_out=False)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        'weight_nn.layer0.weight': (8, 64), 'weight_nn.layer1.weight': (64, 64), 'weight_nn.layer2.weight': (64, spec.weight_numel),
        'denominator': (1,)}
    assert m.spec.tag == spec.tag and m.spec.weight_numel == spec.weight_numel == 960
    assert [(p.i_x, p.i_sh, p.l3) for p in m.spec.paths] == [(p.i_x, p.i_sh, p.l3) for p in spec.paths]
    m.load_state_dict({'weight_nn.layer0.weight': torch.zeros(8, 64), 'weight_nn.layer1.weight': torch.zeros(64, 64),
                       'weight_nn.layer2.weight': torch.zeros(64, 960), 'denominator': torch.tensor([35.989574])})
    assert abs(float(m.denominator) - 35.989574) < 1e-6
    with pytest.raises(RuntimeError, match='no CPU path'):
        m({'x': torch.zeros(4, spec.irreps_x.dim), 'edge_attr': torch.zeros(3, 9), 'edge_embedding': torch.zeros(3, 8),
           'edge_index': torch.zeros(2, 3, dtype=torch.long)})
    # shapes / networks outside the fused kernels' domain are refused by name
    small = build_model_spec(unit_test_config()).layers[1].conv
    with pytest.raises(NotImplementedError, match='no fused kernels'):
        HipFusedIrrepsConvolution(str(small.irreps_x), str(small.irreps_sh), str(small.irreps_out), [8, 64, 64])
    with pytest.raises(NotImplementedError, match='radial network'):
        HipFusedIrrepsConvolution(str(spec.irreps_x), str(spec.irreps_sh), str(spec.irreps_out), [8, 32, 32], sort_by_out=False)
    with pytest.raises(NotImplementedError, match='radial activation'):
        HipFusedIrrepsConvolution(str(spec.irreps_x), str(spec.irreps_sh), str(spec.irreps_out), [8, 64, 64], 'gelu', sort_by_out=False)

    class Stub:   # the attributes sevenn/nn/convolution.py:50-105 sets; e3nn-order instructions re-sorted by output block (:78-82)
        def __init__(self, conv):
            order = sorted(range(len(conv.paths)), key=lambda q: (conv.paths[q].out_off, conv.paths[q].out_ch))
            k_of = {q: k for k, q in enumerate(order)}
            self.convolution_kwargs = dict(irreps_in1=str(conv.irreps_x), irreps_in2=str(conv.irreps_sh), irreps_out=str(conv.irreps_mid),
                                           instructions=[(p.i_x, p.i_sh, k_of[q], 'uvu', True) for q, p in enumerate(conv.paths)])
            self.weight_nn_kwargs = dict(hs=[8, 64, 64, conv.weight_numel], act=torch.nn.functional.silu)
            self.denominator = torch.nn.Parameter(torch.tensor([28.0]), requires_grad=False)
            self.key_x, self.key_filter, self.key_weight_input, self.key_edge_idx = 'x', 'edge_attr', 'edge_embedding', 'edge_index'
            self.is_parallel, self.layer_instantiated = False, False
    got = patch_convolution(Stub(spec))
    assert isinstance(got, HipFusedIrrepsConvolution) and got.spec.tag == spec.tag and got.act == 'silu' and got.layer_instantiated
    with pytest.warns(UserWarning, match='no fused kernels'):
        with pytest.raises(ModuleNotFoundError):     # the fallback is the reference's own class: its package is not installed here
            patch_convolution(Stub(small))
