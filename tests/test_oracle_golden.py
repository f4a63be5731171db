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
