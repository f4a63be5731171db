"""CPU: the oracle restatement against the reference's own data."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_ts_golden, oracle_model


def test_w3j_matches_reference_checkpoint_buffers():
    """8 real Wigner-3j tensors stored by e3nn in tests/data/checkpoints/cp_0.pth."""
    from oracle.e3 import wigner_3j
    g = np.load(f'{GOLDEN}/w3j_cp0.npz')
    assert len(g.files) == 8
    for k in g.files:
        l1, l2, l3 = map(int, k.split('_'))
        assert np.abs(wigner_3j(l1, l2, l3).numpy() - g[k]).max() < 5e-8, k


def test_product_cg_tables_match_golden_and_oracle():
    from oracle.e3 import wigner_3j
    from sevennet_amd.irreps import real_wigner_3j
    g = np.load(f'{GOLDEN}/w3j_cp0.npz')
    for k in g.files:
        l1, l2, l3 = map(int, k.split('_'))
        assert np.abs(real_wigner_3j(l1, l2, l3) - g[k]).max() < 5e-8
    for l1 in range(4):
        for l2 in range(4):
            for l3 in range(abs(l1 - l2), min(3, l1 + l2) + 1):
                assert np.abs(real_wigner_3j(l1, l2, l3) - wigner_3j(l1, l2, l3).numpy()).max() < 1e-12


@pytest.mark.parametrize('cf,ref', [
    ({}, 20642), ({'num_convolution_layer': 4}, 33458), ({'lmax': 3}, 26866), ({'channel': 2}, 16883),
    ({'is_parity': False}, 20386), ({'self_connection_type': 'linear'}, 20114)])
def test_num_params_pins(cf, ref):
    """tests/unit_tests/test_model.py:164-182 of the reference."""
    from oracle.model import OracleModel
    from sevennet_amd.model_spec import build_model_spec
    from sevennet_amd.shapes import unit_test_config
    cfg = unit_test_config(**cf)
    assert OracleModel(cfg, None).num_weights() == ref
    assert build_model_spec(cfg).num_weights() == ref


_MODAL = dict(use_modality=True, _number_of_modalities=2, _modal_map={'x1': 0, 'x2': 1}, use_modal_node_embedding=False,
              use_modal_self_inter_intro=False, use_modal_self_inter_outro=False, use_modal_output_block=False,
              use_modal_wise_shift=False, use_modal_wise_scale=False)


@pytest.mark.parametrize('cf,ref', [
    ({}, 20642), ({'use_modal_node_embedding': True}, 20642 + 8), ({'use_modal_self_inter_intro': True}, 20642 + 2 * 4 * 3),
    ({'use_modal_self_inter_outro': True}, 20642 + 2 * (12 + 20 + 4)), ({'use_modal_output_block': True}, 20642 + 2 * 4 // 2)])
def test_modal_num_params_pins(cf, ref):
    """multi-modal variants, tests/unit_tests/test_model.py:185-212 of the reference"""
    from oracle.model import OracleModel
    from sevennet_amd.model_spec import build_model_spec
    from sevennet_amd.shapes import unit_test_config
    cfg = unit_test_config(**dict(_MODAL, **cf))
    assert OracleModel(cfg, None, modal='x1').num_weights() == ref
    assert build_model_spec(cfg).num_weights() == ref


def test_modal_onehot_is_a_channel_dependent_bias():
    """oracle (literal one-hot concatenation, linear.py:72-92) vs the product's load-time folding
    (bias rows + rescaled alpha, model_spec.linear_modal_bias): same linear map for each channel;
    modal-wise shift picks its row (scale.py:341-363); a missing / unknown modal is refused"""
    import torch
    from oracle.e3 import Irreps as OIrreps
    from oracle.model import OracleModel, linear_apply
    from sevennet_amd.irreps import Irreps
    from sevennet_amd.model_spec import build_model_spec, linear_modal_bias, linear_weight_matrices, make_linear
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    rng = np.random.default_rng(0)
    irr_in, irr_out, M = '4x0e+3x1o+2x0e', '5x0e+2x1o+1x0o', 3
    spec = make_linear('w', Irreps(irr_in), Irreps(irr_out), n_modal=M)
    flat = rng.standard_normal(spec.numel)
    x = torch.as_tensor(rng.standard_normal((7, Irreps(irr_in).dim)))
    for m in range(M):
        oh = torch.zeros(7, M, dtype=torch.float64)
        oh[:, m] = 1.0
        ref = linear_apply(torch.cat([x, oh], 1), OIrreps(irr_in) + OIrreps(f'{M}x0e'), OIrreps(irr_out), torch.as_tensor(flat))
        from helpers import irmul_to_mulir  # noqa: F401  (layouts: oracle mul_ir, product ir_mul)
        got = np.zeros((7, spec.dim_out))
        xin = x.numpy()
        from sevennet_amd.irreps import mulir_to_irmul_index
        xi = xin[:, mulir_to_irmul_index(Irreps(irr_in))]
        for b, w in zip(spec.blocks, linear_weight_matrices(spec, flat)):
            d = 2 * b.l + 1
            blk = xi[:, b.in_off:b.in_off + d * b.mul_in].reshape(7, d, b.mul_in) @ w.astype(np.float64)
            got[:, b.out_off:b.out_off + d * b.mul_out] += blk.reshape(7, -1)
        got += linear_modal_bias(spec, flat, m)[None, :]
        ref_im = ref.numpy()[:, mulir_to_irmul_index(Irreps(irr_out))]
        assert np.abs(got - ref_im).max() < 1e-6
    cfg = unit_test_config(**dict(_MODAL, use_modal_self_inter_intro=True, use_modal_wise_shift=True, shift=0.0, scale=1.0))
    sd = random_state_dict(cfg, seed=2)
    sd['rescale_atomic_energy.shift'] = np.array([[0.0, 1.0, 2.0, 3.0], [10.0, 11.0, 12.0, 13.0]], np.float32)
    sp = build_model_spec(cfg)
    sc, sh = sp.rescale_vectors(sd, sp.modal_index('x2'))
    assert np.allclose(sh, [10, 11, 12, 13]) and np.allclose(sc, 1.0)
    with pytest.raises(ValueError):
        sp.modal_index(None)
    with pytest.raises(KeyError):
        sp.modal_index('nope')
    with pytest.raises(ValueError):
        OracleModel(cfg, sd)


def test_cp0_state_dict_shapes():
    """every tensor of the reference's test checkpoint has the shape the engine expects"""
    import json
    from sevennet_amd.model_spec import build_model_spec
    d = np.load(f'{GOLDEN}/cp0_state.npz')
    cfg = json.loads(str(d['__config__']))
    shapes = build_model_spec(cfg).param_shapes()
    for k, shp in shapes.items():
        assert k in d.files, k
        assert int(np.prod(d[k].shape)) == int(np.prod(shp)), (k, d[k].shape, shp)
    extra = [k for k in d.files if k not in shapes and not k.startswith('__') and d[k].size > 0]
    assert extra == [], extra


@pytest.mark.parametrize('name', ['hfo2_12', 'hfo_rs64', 'hfo2_96'])
def test_oracle_vs_reference_torchscript_serial(name):
    """E / F / atomic energies / stress of deployed_serial.pt (fp32) vs the fp64 restatement."""
    d, cfg, sd = load_ts_golden(name)
    m = oracle_model(cfg, sd)
    out = m.forward(d['types'], d['edge_index'], d['out_edge_vec'].astype(np.float64), keep=True)
    n = len(d['types'])
    assert abs(float(out['energy']) - float(d['out_energy'])) / n < 5e-6
    assert np.abs(out['atomic_energy'].numpy() - d['out_atomic_energy']).max() < 2e-5
    assert np.abs(out['forces'].numpy() - d['out_forces']).max() < 1e-5
    vol = abs(np.linalg.det(d['cell']))
    stress = (out['virial'].numpy() / vol)  # = -sum(r x g)/V, order xx,yy,zz,xy,yz,zx
    assert np.abs(stress - d['out_stress']).max() < 1e-6
    assert np.abs(out['inter']['edge_embedding'].numpy() - d['out_edge_embedding']).max() < 1e-6
    assert np.abs(out['inter']['edge_attr'].numpy() - d['out_edge_attr']).max() < 1e-6
    from oracle.model import linear_apply
    hid = linear_apply(out['inter']['3_x'], m.irreps_final, m.irreps_hidden, m.p['reduce_input_to_hidden.linear.weight'])
    assert np.abs(hid.numpy() - d['out_x_final']).max() < 2e-5  # data['x'] after reduce_input_to_hidden


@pytest.mark.parametrize('name', ['hfo2_12', 'hfo_rs64'])
def test_oracle_vs_reference_torchscript_segments(name):
    """per-segment node features and dE/dr of the four deployed_parallel segments"""
    d, cfg, sd = load_ts_golden(name, parallel=True)
    m = oracle_model(cfg, sd)
    out = m.forward(d['types'], d['edge_index'], d['out_edge_vec'].astype(np.float64), keep=True)
    for k in range(3):
        ls = m.layers[k + 1]
        x = out['inter'][f'{k}_x']
        onehot = torch.nn.functional.one_hot(torch.as_tensor(d['types']), 2).double()
        assert np.abs(m.si1(ls, x).numpy() - d[f'seg{k}_x']).max() < 2e-5
        assert np.abs(m.sc_intro(ls, x, onehot).numpy() - d[f'seg{k}_self_cont_tmp']).max() < 2e-5
    assert np.abs(out['dE_dr'].numpy() - d['par_dE_dr']).max() < 1e-5
    assert abs(float(out['energy']) - float(d['par_energy'])) / len(d['types']) < 5e-6


def test_oracle_forces_match_finite_differences_l3():
    """l=3 / normalised SH / XPLOR / linear self-connection are unpinned by reference data:
    check the oracle's autograd forces against central differences in fp64."""
    from sevennet_amd.synthetic import random_state_dict
    from sevennet_amd.shapes import unit_test_config
    from helpers import synthetic_system
    cfg = unit_test_config(lmax=3, self_connection_type='linear', num_convolution_layer=2, _number_of_species=2,
                           cutoff_function={'cutoff_function_name': 'XPLOR', 'cutoff_on': 3.5})
    sd = random_state_dict(cfg, 3)
    m = oracle_model(cfg, sd)
    types, pos, cell, ei, ev = synthetic_system((1, 1, 1), sigma=0.1, seed=5, cutoff=4.0, n_species=2)
    out = m.forward(types, ei, ev)
    g = out['dE_dr'].numpy()
    rng = np.random.default_rng(0)
    for e in rng.choice(ei.shape[1], 4, replace=False):
        for a in range(3):
            h = 1e-5
            evp, evm = ev.copy(), ev.copy()
            evp[e, a] += h
            evm[e, a] -= h
            fd = (float(m.forward(types, ei, evp)['energy']) - float(m.forward(types, ei, evm)['energy'])) / (2 * h)
            assert abs(fd - g[e, a]) < 1e-6 * max(1.0, abs(fd)), (e, a, fd, g[e, a])


def test_oracle_rotation_equivariance_l3():
    from sevennet_amd.synthetic import random_state_dict
    from sevennet_amd.shapes import unit_test_config
    from helpers import synthetic_system
    cfg = unit_test_config(lmax=3, num_convolution_layer=2, _number_of_species=2)
    sd = random_state_dict(cfg, 4)
    m = oracle_model(cfg, sd)
    types, pos, cell, ei, ev = synthetic_system((1, 1, 1), sigma=0.1, seed=6, cutoff=4.0, n_species=2)
    q, _ = np.linalg.qr(np.random.default_rng(1).standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    a = m.forward(types, ei, ev)
    b = m.forward(types, ei, ev @ q.T)
    assert abs(float(a['energy']) - float(b['energy'])) < 1e-9 * abs(float(a['energy']))
    assert np.abs(a['forces'].numpy() @ q.T - b['forces'].numpy()).max() < 1e-9
    # improper rotation (parity model): energy invariant as well
    c = m.forward(types, ei, -ev)
    assert abs(float(a['energy']) - float(c['energy'])) < 1e-9 * abs(float(a['energy']))


# --------------------------------------------------------------------------- #
# the reference's own pure-torch modules, executed from /root/reference in the build container
# (oracle/tools/make_golden_torch_modules.py) -> tests/golden/ref_torch_modules.npz
# --------------------------------------------------------------------------- #
def _ref_modules():
    return np.load(f'{GOLDEN}/ref_torch_modules.npz')


@pytest.mark.parametrize('tag', ['rc5', 'rc6', 'rc4'])
def test_oracle_radial_basis_and_cutoffs_vs_reference_modules(tag):
    """BesselBasis / PolynomialCutoff / XPLORCutoff.forward (edge_embedding.py:101-103,125-132,150-160),
    incl. the points around r_on and r_cut: fp64 restatement vs the reference classes run in fp64"""
    from oracle.model import bessel_basis, poly_cutoff, xplor_cutoff
    d = _ref_modules()
    rc, r_on, p = d[f'{tag}_params']
    r = torch.tensor(d[f'{tag}_r'])
    b = bessel_basis(r, torch.tensor(d[f'{tag}_coeffs_f64']), rc)
    assert np.abs(b.numpy() - d[f'{tag}_bessel_f64']).max() <= 1e-13 * np.abs(d[f'{tag}_bessel_f64']).max()
    assert np.abs(poly_cutoff(r, rc, int(p)).numpy() - d[f'{tag}_poly_f64']).max() <= 1e-13
    assert np.abs(xplor_cutoff(r, rc, r_on).numpy() - d[f'{tag}_xplor_f64']).max() <= 1e-13
    # fp32 arithmetic of the same formulae stays within fp32 rounding of the reference's fp32 run
    r32 = r.float()
    assert np.abs(xplor_cutoff(r32, float(rc), float(r_on)).numpy() - d[f'{tag}_xplor_f32']).max() <= 2e-6
    assert np.abs(poly_cutoff(r32, float(rc), int(p)).numpy() - d[f'{tag}_poly_f32']).max() <= 2e-5


def test_oracle_force_virial_vs_reference_module():
    """ForceStressOutputFromEdge.forward (force_output.py:171-230): force sign rule, virial component order
    (xx, yy, zz, xy, yz, zx), atomic virial assigned to edge_index[1], stress = -sum / volume"""
    from oracle.model import force_virial_from_edge
    d = _ref_modules()
    g, rij = torch.tensor(d['fs_gij_f64']), torch.tensor(d['fs_rij_f64'])
    ei = torch.tensor(d['fs_edge_index'])
    n = d['fs_force_f64'].shape[0]
    out = force_virial_from_edge(g, rij, ei, n)
    assert np.abs(out['forces'].numpy() - d['fs_force_f64']).max() <= 1e-12 * np.abs(d['fs_force_f64']).max()
    assert np.abs(out['atomic_virial'].numpy() - d['fs_atomic_virial_f64']).max() <= 1e-12 * np.abs(d['fs_atomic_virial_f64']).max()
    stress = out['virial'].numpy() / d['fs_volume'][0]
    assert np.abs(stress - d['fs_stress_f64']).max() <= 1e-12 * np.abs(d['fs_stress_f64']).max()


def test_oracle_rescale_vs_reference_modules():
    """Rescale / SpeciesWiseRescale / ModalWiseRescale.forward (scale.py:53-56,155-162,341-363), AtomReduce
    (linear.py:127-141), OnehotEmbedding (node_embedding.py:44-53)"""
    from oracle.model import rescale_apply
    d = _ref_modules()
    e, types = torch.tensor(d['rs_in']), torch.tensor(d['rs_types'])
    sh, sc = d['rs_global_params']
    out = rescale_apply(e, types, torch.tensor([sc], dtype=torch.float32), torch.tensor([sh], dtype=torch.float32))
    assert np.abs(out.numpy() - d['rs_global']).max() <= 1e-6
    ss, cs = torch.tensor(d['rs_species_shift']), torch.tensor(d['rs_species_scale'])
    out = rescale_apply(e, types, cs, ss)
    assert np.abs(out.numpy() - d['rs_species']).max() <= 1e-6
    assert abs(float(out.sum()) - float(d['reduce_total'][0])) <= 1e-4
    sm, cm = torch.tensor(d['rs_modal_shift']), torch.tensor(d['rs_modal_scale'])
    for modal in range(sm.shape[0]):
        for tag, shift, scale in (('mm', sm, cm), ('ms', sm, cs), ('sm', ss, cm)):
            out = rescale_apply(e, types, scale, shift, modal)
            assert np.abs(out.numpy() - d[f'rs_modal_{tag}_{modal}']).max() <= 1e-6, (tag, modal)
    oh = torch.nn.functional.one_hot(types, d['onehot'].shape[1]).float().numpy()
    assert np.array_equal(oh, d['onehot'])
