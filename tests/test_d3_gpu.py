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


def _reference_stub(lib):
    """the ctypes declarations of the reference's D3Calculator._load_cuda_library (sevenn/calculator.py:430-483), restated:
    argtypes / restypes exactly as the reference binds its libpair_d3.so"""
    import ctypes

    class PairD3(ctypes.Structure):
        pass

    P = ctypes.POINTER(PairD3)
    lib.pair_init.restype = P
    lib.pair_set_atom.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    lib.pair_set_atom.restype = None
    lib.pair_set_domain.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double), ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.pair_set_domain.restype = None
    lib.pair_run_settings.argtypes = [P, ctypes.c_double, ctypes.c_double, ctypes.c_char_p, ctypes.c_char_p]
    lib.pair_run_settings.restype = None
    lib.pair_run_coeff.argtypes = [P, ctypes.POINTER(ctypes.c_int)]
    lib.pair_run_coeff.restype = None
    lib.pair_run_compute.argtypes = [P]
    lib.pair_run_compute.restype = None
    lib.pair_get_energy.argtypes = [P]
    lib.pair_get_energy.restype = ctypes.c_double
    lib.pair_get_force.argtypes = [P]
    lib.pair_get_force.restype = ctypes.POINTER(ctypes.c_double)
    lib.pair_get_stress.argtypes = [P]
    lib.pair_get_stress.restype = ctypes.POINTER(ctypes.c_double * 6)
    lib.pair_fin.argtypes = [P]
    lib.pair_fin.restype = None


def _reference_calculate(lib, numbers, positions, cell, pbc, damp=b'damp_bj', func=b'pbe', rthr=9000.0, cnthr=1600.0):
    """D3Calculator.calculate of the reference (sevenn/calculator.py:527-612), restated over the same `pair_*` calls:
    ASE cell -> LAMMPS box + rotation, 1-based types in first-seen order, results rotated back, stress = -virial / V"""
    import ctypes
    cell = np.asarray(cell, float)
    qtrans, ltrans = np.linalg.qr(cell.T, mode='complete')
    lc = ltrans.T
    signs = np.sign(np.diag(lc))
    lc, qtrans = lc * signs, qtrans * signs
    box = lc[(0, 1, 2, 1, 2, 2), (0, 1, 2, 0, 0, 1)]
    rot = qtrans.T
    uniq = list(dict.fromkeys(numbers))
    natoms, ntypes = len(numbers), len(uniq)
    types = (ctypes.c_int * natoms)(*[uniq.index(z) + 1 for z in numbers])
    x = (ctypes.c_double * (3 * natoms))(*(np.asarray(positions, float) @ rot.T).flatten())
    zs = (ctypes.c_int * ntypes)(*uniq)
    pair = lib.pair_init()
    lib.pair_set_atom(pair, natoms, ntypes, types, x)
    lib.pair_set_domain(pair, int(pbc[0]), int(pbc[1]), int(pbc[2]), (ctypes.c_double * 3)(0.0, 0.0, 0.0),
                        (ctypes.c_double * 3)(box[0], box[1], box[2]), box[3], box[4], box[5])
    lib.pair_run_settings(pair, rthr, cnthr, damp, func)
    lib.pair_run_coeff(pair, zs)
    lib.pair_run_compute(pair)
    e = lib.pair_get_energy(pair)
    f = np.array(np.ctypeslib.as_array(lib.pair_get_force(pair), shape=(3 * natoms,))).reshape(natoms, 3) @ rot
    s = np.array(lib.pair_get_stress(pair).contents)
    t = np.array([[s[0], s[3], s[4]], [s[3], s[1], s[5]], [s[4], s[5], s[2]]])
    t = rot.T @ t @ rot
    stress = -np.array([t[0, 0], t[1, 1], t[2, 2], t[1, 2], t[0, 2], t[0, 1]]) / abs(np.linalg.det(cell))
    lib.pair_fin(pair)
    return e, f, stress


def test_reference_pair_d3_binding_known_answers():
    """the reference's OWN binding (`pair_init` .. `pair_fin`, pair_d3_for_ase.cu:2034-2082) exported by libsnet_hip.so:
    driven exactly the way sevenn.calculator.D3Calculator drives its CUDA library, it returns the reference's known answers
    (tests/unit_tests/test_calculator.py:192-236) and agrees with the snet_d3_* engine it wraps"""
    import ctypes
    from sevennet_amd import _lib
    from sevennet_amd.d3 import D3Engine
    lib = ctypes.CDLL(_lib.LIB_PATH)     # a plain handle, like the reference's _load('pair_d3')
    _reference_stub(lib)
    e, f, s = _reference_calculate(lib, NACL['numbers'], NACL['positions'], NACL['cell'], NACL['pbc'])
    assert abs(e - NACL_REF['energy']) < RTOL * abs(NACL_REF['energy'])
    assert np.abs(f - np.array(NACL_REF['forces'])).max() < RTOL * np.abs(NACL_REF['forces']).max()
    assert np.abs(s - np.array(NACL_REF['stress'])).max() < RTOL * np.abs(NACL_REF['stress']).max()
    e, f, s = _reference_calculate(lib, [8, 1, 1], H2O_POS, h2o_box(), [True] * 3)
    assert abs(e - H2O_REF['energy']) < 2e-6 * abs(H2O_REF['energy'])
    assert np.abs(f - np.array(H2O_REF['forces'])).max() < RTOL * np.abs(H2O_REF['forces']).max()
    # a triclinic mixed cell with partial periodicity, zero damping: shim == engine to rounding of the frame rotation
    rng = np.random.default_rng(5)
    cell = np.array([[7.0, 0.4, 0.0], [0.3, 6.5, 0.5], [0.2, 0.6, 8.0]])
    pos = rng.uniform(0.0, 1.0, (7, 3)) @ cell
    Z = [6, 8, 1, 14, 8, 22, 1]
    ref = D3Engine('damp_zero', 'b3-lyp', 1600.0, 900.0).compute(Z, pos, cell, [True, True, False])
    e, f, s = _reference_calculate(lib, Z, pos, cell, [True, True, False], b'damp_zero', b'b3-lyp', 1600.0, 900.0)
    assert abs(e - ref['energy']) < 1e-9 * abs(ref['energy'])
    assert np.abs(f - ref['forces']).max() < 1e-8 * np.abs(ref['forces']).max()
    assert np.abs(s - voigt(ref['stress'])).max() < 1e-8 * np.abs(ref['stress']).max()


def test_reference_pair_d3_binding_failures_are_not_silent():
    """a failed handle must not hand out zero-filled results.  With a GPU present the failure that matters is a bad name
    (the reference aborts via error->all, pair_d3.cu:261-285): the shim's failure is sticky, pair_get_force / pair_get_stress return NULL,
    pair_get_energy NaN, pair_failed reads 1 -- also after a later, otherwise valid, compute sequence on the same handle."""
    import ctypes
    from sevennet_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    _reference_stub(lib)
    lib.pair_failed.argtypes = [ctypes.c_void_p]
    lib.pair_failed.restype = ctypes.c_int
    lib.pair_get_force.restype = ctypes.c_void_p       # (NULL must be visible as None)
    lib.pair_get_stress.restype = ctypes.c_void_p
    for damp, func in ((b'damp_bj', b'no-such-functional'), (b'damp_foo', b'pbe')):
        pair = lib.pair_init()
        assert not lib.pair_failed(pair)
        types = (ctypes.c_int * 2)(1, 2)
        x = (ctypes.c_double * 6)(0, 0, 0, 2.815, 0, 0)
        lib.pair_set_atom(pair, 2, 2, types, x)
        lib.pair_set_domain(pair, 1, 1, 1, (ctypes.c_double * 3)(0, 0, 0), (ctypes.c_double * 3)(5.63, 5.63, 5.63), 0.0, 0.0, 0.0)
        lib.pair_run_settings(pair, 9000.0, 1600.0, damp, func)
        assert lib.pair_failed(pair)
        lib.pair_run_coeff(pair, (ctypes.c_int * 2)(11, 17))
        lib.pair_run_compute(pair)
        assert lib.pair_failed(pair) and np.isnan(lib.pair_get_energy(pair))
        assert lib.pair_get_force(pair) is None and lib.pair_get_stress(pair) is None
        lib.pair_fin(pair)
    # a healthy handle next to them is unaffected, and repeated settings with unchanged arguments are no-ops
    lib.pair_get_force.restype = ctypes.POINTER(ctypes.c_double)
    lib.pair_get_stress.restype = ctypes.POINTER(ctypes.c_double * 6)
    e1, f1, s1 = _reference_calculate(lib, NACL['numbers'], NACL['positions'], NACL['cell'], NACL['pbc'])
    e2, f2, s2 = _reference_calculate(lib, NACL['numbers'], NACL['positions'], NACL['cell'], NACL['pbc'])
    assert e1 == e2 and np.array_equal(f1, f2) and abs(e1 - NACL_REF['energy']) < RTOL * abs(NACL_REF['energy'])


@pytest.mark.parametrize('damp', ['damp_bj', 'damp_zero'])
def test_d3_forces_and_stress_are_derivatives_of_the_energy_at_1000_atoms(damp):
    """f4 through size-independent properties (the reference's kernels carry no such test; its known answers are 2 .. 96 atoms): a
    1 000-atom two-species cell at the DEFAULT cutoffs (vdw 9000, cn 1600 bohr^2: ~25 k pairs per atom, several periodic images of the
    cell), fp64 kernels: central differences of the energy along a collective displacement reproduce -sum F.d, along a symmetric strain
    reproduce V sigma:eps (pair_d3_for_ase.cu:1420-1780 accumulates both in the same pair loop), a rigid rotation leaves the energy and
    rotates forces and stress."""
    from sevennet_amd.d3 import D3Engine
    from sevennet_amd.neighbor import diamond_cubic
    pos, cell = diamond_cubic(5.431, (5, 5, 5), 0.08, 3)
    cell = np.asarray(cell, np.float64)
    z = np.where(np.arange(len(pos)) % 2 == 0, 14, 8).tolist()
    eng = D3Engine(damp, 'pbe', 9000.0, 1600.0)
    ev = lambda p, c: eng.compute(z, p, c, [True] * 3)   # noqa: E731
    a = ev(pos, cell)
    f, s, vol = a['forces'], a['stress'], abs(np.linalg.det(cell))
    assert np.abs(f.sum(0)).max() < 1e-9 * np.abs(f).max() * len(pos)
    d = f / np.abs(f).max()
    h = 1e-4
    lhs = (ev(pos + h * d, cell)['energy'] - ev(pos - h * d, cell)['energy']) / (2 * h)
    # (D3 cuts pair and coordination-number sums off SHARPLY at the two radii, as the reference does: the ~300 pairs that cross the 50-A
    # shell between the two displaced configurations each leave e_pair / 2h in the difference quotient -- 4.5e-6 relative measured)
    assert abs(lhs + (f * d).sum()) <= 2e-5 * abs((f * d).sum()), (lhs, -(f * d).sum())
    eps = np.array([[0.3, 0.1, -0.2], [0.1, -0.5, 0.4], [-0.2, 0.4, 0.7]]) * 1e-4     # symmetric strain
    def strained(sign):   # noqa: E306
        m = np.eye(3) + sign * eps
        return ev(pos @ m, cell @ m)['energy']
    lhs = (strained(+1) - strained(-1)) / 2
    rhs = vol * (s * eps).sum()
    # (under a strain every pair distance moves in proportion to r: the count of pairs crossing the sharp cutoff grows with the strain
    # itself, so their share of the difference quotient does not vanish with it -- 2.1e-4 relative measured; the analytic stress, like the
    # reference's, has no surface term)
    assert abs(lhs - rhs) <= 1e-3 * abs(rhs) + 3e-6, (lhs, rhs)   # (the surface term is 1.3e-6 / 1.7e-6 eV for this strain, both dampings)
    q, r = np.linalg.qr(np.random.default_rng(5).normal(size=(3, 3)))
    R = q * np.sign(np.diag(r))
    if np.linalg.det(R) < 0:
        R[:, 0] = -R[:, 0]
    b = ev(pos @ R.T, cell @ R.T)
    assert abs(b['energy'] - a['energy']) <= 1e-10 * abs(a['energy'])
    assert np.abs(b['forces'] - f @ R.T).max() <= 1e-9 * np.abs(f).max()
    assert np.abs(b['stress'] - R @ s @ R.T).max() <= 1e-9 * np.abs(s).max()
