"""GPU: the HIP D3 dispersion (csrc/snet_d3.hip through the C-ABI snet_d3_*) against the fp64 oracle and against the
reference's known answers (tests/unit_tests/test_calculator.py:192-236)."""
import numpy as np
import pytest

from test_d3_cpu import H2O_POS, H2O_REF, NACL, NACL_REF, RTOL, h2o_box, voigt

pytestmark = pytest.mark.gpu


def test_d3_calculator_known_answers():
    from sevennet_amd.d3 import D3Calculator
    c = D3Calculator()
    r = c.compute(NACL['numbers'], NACL['positions'], NACL['cell'], NACL['pbc'])
    assert abs(r['energy'] - NACL_REF['energy']) < RTOL * abs(NACL_REF['energy']) and r['free_energy'] == r['energy']
    assert np.abs(r['forces'] - np.array(NACL_REF['forces'])).max() < RTOL * np.abs(NACL_REF['forces']).max()
    assert np.abs(r['stress'] - np.array(NACL_REF['stress'])).max() < RTOL * np.abs(NACL_REF['stress']).max()
    # molecule without a cell: the calculator generates the reference's box (calculator.py:533-548)
    r = c.compute([8, 1, 1], H2O_POS, np.zeros((3, 3)), [False] * 3)
    assert abs(r['energy'] - H2O_REF['energy']) < 2e-6 * abs(H2O_REF['energy'])
    assert np.abs(r['forces'] - np.array(H2O_REF['forces'])).max() < RTOL * np.abs(H2O_REF['forces']).max()
    with pytest.raises(ValueError, match='Invalid damping'):
        D3Calculator(damping_type='damp_foo')
    with pytest.raises(ValueError, match='Functional name unknown'):
        D3Calculator(functional_name='no-such-functional')


@pytest.mark.parametrize('damp,func', [('damp_bj', 'pbe'), ('damp_zero', 'pbe'), ('damp_bj', 'b3-lyp'), ('damp_zero', 'b2-plyp')])
def test_d3_hip_vs_oracle(damp, func):
    """triclinic mixed cell, partial periodicity, atoms outside the cell (wrapped), both damping functions: every output
    of the kernels against the fp64 oracle to 1e-9 relative (both are fp64; only the summation order differs)"""
    from oracle.d3 import d3
    from sevennet_amd.d3 import D3Engine
    rng = np.random.default_rng(3)
    cell = np.array([[7.0, 0.4, 0.0], [0.3, 6.5, 0.5], [0.2, 0.6, 8.0]])
    pos = rng.uniform(-0.3, 1.2, (9, 3)) @ cell
    Z = [6, 8, 1, 14, 8, 22, 1, 1, 79]
    for pbc in ([True, True, True], [True, True, False], [False, False, False]):
        ref = d3(Z, pos, cell, pbc, damping=damp, functional=func, vdw_cutoff=1600.0, cn_cutoff=900.0)
        out = D3Engine(damp, func, 1600.0, 900.0).compute(Z, pos, cell, pbc)
        assert np.abs(out['cn'] - ref['cn']).max() < 1e-10 * max(1.0, np.abs(ref['cn']).max())
        assert abs(out['energy'] - ref['energy']) < 1e-9 * abs(ref['energy'])
        assert np.abs(out['forces'] - ref['forces']).max() < 1e-9 * np.abs(ref['forces']).max()
        assert np.abs(out['stress'] - ref['stress']).max() < 1e-9 * np.abs(ref['stress']).max()
        assert np.abs(out['forces'].sum(0)).max() < 1e-12


def test_d3_is_reproducible_and_handles_a_larger_cell():
    """deterministic reductions: two evaluations are bit-identical; 216 Si atoms (cutoffs shorter than the defaults)"""
    from sevennet_amd.d3 import D3Engine
    from sevennet_amd.neighbor import diamond_cubic
    pos, cell = diamond_cubic(5.431, (3, 3, 3), 0.05, 0)
    eng = D3Engine('damp_bj', 'pbe', 2500.0, 900.0)
    a = eng.compute([14] * len(pos), pos, cell, [True] * 3)
    b = eng.compute([14] * len(pos), pos, cell, [True] * 3)
    assert a['energy'] == b['energy'] and np.array_equal(a['forces'], b['forces']) and np.array_equal(a['stress'], b['stress'])
    assert a['energy'] < 0 and np.abs(a['forces'].sum(0)).max() < 1e-9 * max(1.0, np.abs(a['forces']).max() * len(pos))
    assert 3.5 < a['cn'].min() and a['cn'].max() < 4.6, (a['cn'].min(), a['cn'].max())   # bulk silicon: four-fold, CN ~ 3.96
