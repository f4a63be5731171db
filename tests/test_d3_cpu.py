"""CPU: the D3 parameter blob and the fp64 oracle of the dispersion term (oracle/d3.py), pinned by the reference's own
known answers (tests/unit_tests/test_calculator.py:192-236: D3Calculator() = PBE, Becke-Johnson damping, cutoffs 9000 /
1600 bohr^2, on the 2-atom NaCl cell and on H2O in the automatically generated box)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOB = os.path.join(ROOT, 'sevennet_amd', 'data', 'd3_params.npz')

NACL = dict(numbers=[11, 17], positions=[[0.0, 0.0, 0.0], [2.815, 0.0, 0.0]],
            cell=[[1.0, 2.815, 2.815], [2.815, 0.0, 2.815], [2.815, 2.815, 0.0]], pbc=[True] * 3)
NACL_REF = dict(energy=-0.531393751583389,
                forces=[[-0.00570205, 0.00107457, 0.00107459], [0.00570205, -0.00107457, -0.00107459]],
                stress=[1.52403705e-02, 1.50417333e-02, 1.50417321e-02, -3.22684163e-05, -5.05532863e-05, -5.05586994e-05])
H2O_POS = np.array([[0.0, 0.2, 0.12], [0.0, 0.76, -0.48], [0.0, -0.76, -0.48]])
H2O_REF = dict(energy=-0.009889134535170716,
               forces=[[0.0, 2.04263840e-03, 1.27477674e-03], [0.0, -9.90038901e-05, 1.18046682e-06],
                       [0.0, -1.94363451e-03, -1.27595721e-03]])
# The reference evaluates pairs in float32 and sums ~1e5 lattice translations per pair in float32 registers: its
# literals carry that rounding (NaCl: 4.5e-5 on the energy against this fp64 restatement, 6e-7 for the molecule)
RTOL = 1e-4


def h2o_box():
    """sevenn/calculator.py:533-548: cell generated for a molecule"""
    return np.eye(3) * (H2O_POS.max(0) - H2O_POS.min(0) + np.sqrt(9000.0) * 0.52917726 + 1.0)


def voigt(s):
    return np.array([s[0, 0], s[1, 1], s[2, 2], s[1, 2], s[0, 2], s[0, 1]])


def test_blob_holds_the_published_tables():
    z = np.load(BLOB)
    assert z['r0ab'].shape == (94, 94) and z['c6ab'].shape == (32385, 5) and z['r2r4'].shape == (94,) and z['rcov'].shape == (94,)
    assert np.allclose(z['r0ab'], z['r0ab'].T) and abs(z['r0ab'][0, 0] - 2.1823) < 1e-12
    assert abs(z['rcov'][0] - 0.80628308) < 1e-12 and abs(z['r2r4'][0] - 2.00734898) < 1e-12   # hydrogen
    names = z['damp_bj_names'].tolist()
    assert np.allclose(z['damp_bj_params'][names.index('pbe')], [1.0, 0.4289, 0.7875, 4.4407, 14.0])
    zn = z['damp_zero_names'].tolist()
    assert np.allclose(z['damp_zero_params'][zn.index('pbe')], [1.0, 1.217, 0.722, 1.0, 14.0])
    # a statement after `break` in the reference's switch is dead code there: b2-plyp / BJ keeps s6 = 1
    assert z['damp_bj_params'][names.index('b2-plyp')][0] == 1.0
    assert z['damp_zero_params'][zn.index('b2-plyp')][0] == 0.64
    c = z['c6ab']
    assert (c[:, 0] > 0).all() and c[:, 1].max() < 600 and (c[:, 3:] >= 0).all()


def test_oracle_matches_reference_known_answers_periodic():
    from oracle.d3 import d3
    r = d3(**NACL)
    assert abs(r['energy'] - NACL_REF['energy']) < RTOL * abs(NACL_REF['energy'])
    assert np.abs(r['forces'] - np.array(NACL_REF['forces'])).max() < RTOL * np.abs(NACL_REF['forces']).max()
    assert np.abs(voigt(r['stress']) - np.array(NACL_REF['stress'])).max() < RTOL * np.abs(NACL_REF['stress']).max()
    assert np.abs(r['forces'].sum(0)).max() < 1e-12


def test_oracle_matches_reference_known_answers_molecule():
    from oracle.d3 import d3
    r = d3([8, 1, 1], H2O_POS, h2o_box(), [True] * 3)
    assert abs(r['energy'] - H2O_REF['energy']) < 2e-6 * abs(H2O_REF['energy'])
    assert np.abs(r['forces'] - np.array(H2O_REF['forces'])).max() < RTOL * np.abs(H2O_REF['forces']).max()


def test_oracle_forces_and_stress_are_derivatives_of_the_energy():
    """central differences of the oracle's own energy (zero damping too: its derivative is the autograd one)"""
    from oracle.d3 import d3
    rng = np.random.default_rng(0)
    cell = np.array([[6.0, 0.3, 0.0], [0.2, 5.5, 0.4], [0.0, 0.5, 6.5]])
    pos = rng.uniform(0, 1, (5, 3)) @ cell
    Z = [6, 8, 1, 14, 8]
    for damp in ('damp_bj', 'damp_zero'):
        kw = dict(damping=damp, vdw_cutoff=900.0, cn_cutoff=400.0)
        r = d3(Z, pos, cell, [True, True, False], **kw)
        h = 1e-4
        p1, p2 = pos.copy(), pos.copy()
        p1[2, 1] += h
        p2[2, 1] -= h
        fd = -(d3(Z, p1, cell, [True, True, False], **kw)['energy'] - d3(Z, p2, cell, [True, True, False], **kw)['energy']) / (2 * h)
        assert abs(fd - r['forces'][2, 1]) < 1e-6 * max(1.0, abs(r['forces']).max() * 1e3)
