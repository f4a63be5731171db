"""CPU: a second opinion on oracle/e3.py for l = 3 (VERDICT r4 missing #5 / next #6b).  The reference's data pins the oracle's real
Wigner-3j tensors and spherical harmonics for l <= 2 only (tests/golden/w3j_cp0.npz, the TorchScript goldens); every lmax-3 number
(SevenNet-l3i5, MF-ompa) rested on the same GENERATOR continued to l = 3.  Here the l <= 3 tables are rebuilt from independent
sources -- sympy.physics.wigner (complex 3j symbols, real Gaunt coefficients) and scipy.special's spherical harmonics -- with the
real-basis convention written out from scratch in this file, and the convention itself is fixed on the golden-pinned l <= 2 cases:

* spherical harmonics: e3nn's real harmonics are the standard real harmonics WITHOUT Condon-Shortley phase, polar axis y
  ((x', y', z') = (z, x, y)), component normalisation sqrt(4 pi) Y -- a statement that reproduces the pinned l <= 2 polynomials, and
  then must reproduce the oracle's l = 3 harmonics (raw and normalised input vectors);
* even l1 + l2 + l3: the unit-norm real 3j tensor is + the normalised real Gaunt tensor (integral of three real harmonics) -- sign
  included; holds on the golden-pinned tensors, asserted for every triple up to l = 3;
* odd l1 + l2 + l3 (no Gaunt integral exists): equal to the complex 3j symbols in the real basis up to ONE overall sign per tensor
  (e3nn's choice of phase, pinned by reference data for l <= 2 only) + the permutation symmetries that tie the l = 3 tensors to
  each other."""
import itertools

import numpy as np
import pytest
import torch

sympy = pytest.importorskip('sympy')
from sympy.physics.wigner import real_gaunt, wigner_3j as w3j_complex   # noqa: E402

GOLDEN = __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'golden')
TRIPLES = [t for t in itertools.product(range(4), repeat=3) if abs(t[0] - t[1]) <= t[2] <= t[0] + t[1]]


def _real_from_complex(l):
    """U[m_real, m_complex]: S_{l,m} = sum_m' U[m, m'] Y_l^{m'} for the real harmonics without Condon-Shortley phase
    (S_{l,m>0} = sqrt2 (-1)^m Re Y_l^m, S_{l,m<0} = sqrt2 (-1)^m Im Y_l^{|m|}, conj Y_l^m = (-1)^m Y_l^{-m})"""
    U = np.zeros((2 * l + 1, 2 * l + 1), complex)
    s = 1 / np.sqrt(2)
    U[l, l] = 1
    for m in range(1, l + 1):
        U[l + m, l + m], U[l + m, l - m] = (-1) ** m * s, s
        U[l - m, l + m], U[l - m, l - m] = (-1) ** m * s / 1j, -s / 1j
    return U


def _invariant_tensor_from_sympy(l1, l2, l3):
    W = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                W[l1 + m1, l2 + m2, l3 - m1 - m2] = float(w3j_complex(l1, l2, l3, m1, m2, -m1 - m2))
    T = np.einsum('ai,bj,ck,ijk->abc', _real_from_complex(l1).conj(), _real_from_complex(l2).conj(),
                  _real_from_complex(l3).conj(), W.astype(complex))
    T = T.imag if np.abs(T.imag).max() > np.abs(T.real).max() else T.real   # (purely real or purely imaginary)
    return T / np.linalg.norm(T)


@pytest.mark.parametrize('l1,l2,l3', TRIPLES)
def test_w3j_equals_sympy_in_the_real_basis_up_to_the_tensor_sign(l1, l2, l3):
    from oracle.e3 import wigner_3j
    C = wigner_3j(l1, l2, l3).numpy()
    T = _invariant_tensor_from_sympy(l1, l2, l3)
    assert abs(np.linalg.norm(C) - 1.0) < 1e-12
    assert min(np.abs(C - T).max(), np.abs(C + T).max()) < 1e-12, (l1, l2, l3)


@pytest.mark.parametrize('l1,l2,l3', [t for t in TRIPLES if sum(t) % 2 == 0])
def test_even_w3j_is_the_normalised_real_gaunt_tensor_sign_included(l1, l2, l3):
    from oracle.e3 import wigner_3j
    G = np.array([[[float(real_gaunt(l1, l2, l3, a - l1, b - l2, c - l3)) for c in range(2 * l3 + 1)]
                   for b in range(2 * l2 + 1)] for a in range(2 * l1 + 1)])
    G /= np.linalg.norm(G)
    assert np.abs(wigner_3j(l1, l2, l3).numpy() - G).max() < 1e-12, (l1, l2, l3)


def test_the_gaunt_sign_rule_holds_on_the_reference_pinned_tensors():
    """the rule of the test above is not an assumption: the 3j buffers e3nn stored in the reference's checkpoint obey it"""
    g = np.load(f'{GOLDEN}/w3j_cp0.npz')
    n_even = 0
    for k in g.files:
        l1, l2, l3 = map(int, k.split('_'))
        if (l1 + l2 + l3) % 2:
            continue
        G = np.array([[[float(real_gaunt(l1, l2, l3, a - l1, b - l2, c - l3)) for c in range(2 * l3 + 1)]
                       for b in range(2 * l2 + 1)] for a in range(2 * l1 + 1)])
        assert np.abs(g[k] - G / np.linalg.norm(G)).max() < 5e-8, k
        n_even += 1
    assert n_even >= 4


def test_w3j_permutation_symmetries_tie_the_l3_tensors_together():
    """C_{l2 l1 l3}[b, a, c] = (-1)^(l1 + l2 + l3) C_{l1 l2 l3}[a, b, c] and C_{l2 l3 l1}[b, c, a] = C_{l1 l2 l3}[a, b, c]: the odd-sum
    l = 3 tensors, whose overall sign no offline data pins, are at least consistent with each other and with the l <= 2 ones"""
    from oracle.e3 import wigner_3j
    for l1, l2, l3 in TRIPLES:
        C = wigner_3j(l1, l2, l3).numpy()
        assert np.abs(wigner_3j(l2, l1, l3).numpy().transpose(1, 0, 2) - (-1) ** (l1 + l2 + l3) * C).max() < 1e-12
        assert np.abs(wigner_3j(l2, l3, l1).numpy().transpose(2, 0, 1) - C).max() < 1e-12


def _scipy_real_harmonics(lmax, v):
    """sqrt(4 pi) x real spherical harmonics without Condon-Shortley phase, polar axis y, from scipy's complex Y_l^m"""
    from scipy import special
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xs, ys, zs = z, x, y                                      # standard frame: polar axis z' = y, x' = z, y' = x
    r = np.sqrt(xs ** 2 + ys ** 2 + zs ** 2)
    polar, azim = np.arccos(zs / r), np.arctan2(ys, xs)
    out = []
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            if hasattr(special, 'sph_harm_y'):
                Y = special.sph_harm_y(l, abs(m), polar, azim)
            else:   # scipy < 1.15: sph_harm(m, l, azimuth, polar)
                Y = special.sph_harm(abs(m), l, azim, polar)
            S = Y.real if m == 0 else np.sqrt(2) * (-1) ** m * (Y.real if m > 0 else Y.imag)
            out.append(np.sqrt(4 * np.pi) * S)
    return np.stack(out, -1)


def test_spherical_harmonics_up_to_l3_equal_scipy():
    from oracle.e3 import spherical_harmonics
    v = np.random.default_rng(0).normal(size=(200, 3))
    ref = _scipy_real_harmonics(3, v)
    got = spherical_harmonics(3, torch.tensor(v), normalize=True).numpy()
    assert got.shape == (200, 16)
    assert np.abs(got[:, :9] - ref[:, :9]).max() < 1e-12          # the convention, on the reference-pinned degrees l <= 2
    assert np.abs(got[:, 9:] - ref[:, 9:]).max() < 1e-12          # ... carried to l = 3
    # un-normalised input (old checkpoints, backward_compatibility.py:38-39): degree-l homogeneity, Y_l(v) = |v|^l Y_l(v / |v|)
    raw = spherical_harmonics(3, torch.tensor(v), normalize=False).numpy()
    r = np.linalg.norm(v, axis=1, keepdims=True)
    scale = np.concatenate([r ** l * np.ones((1, 2 * l + 1)) for l in range(4)], 1)
    assert np.abs(raw - ref * scale).max() < 1e-11 * np.abs(raw).max()
    # component normalisation: ||Y_l(unit)||^2 = 2 l + 1
    for l in range(4):
        assert np.abs((got[:, l * l:(l + 1) ** 2] ** 2).sum(1) - (2 * l + 1)).max() < 1e-12


def test_generated_hip_harmonics_header_matches_scipy_too():
    """the product side: the polynomials sevennet_amd/codegen.py writes into csrc/generated/sh_generated.h (evaluated here through
    the same symbolic table the generator prints) agree with scipy for l <= 3"""
    from sevennet_amd.irreps import sh_polynomials
    v = np.random.default_rng(1).normal(size=(50, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ref = _scipy_real_harmonics(3, v)
    k = 0
    polys = sh_polynomials(3)
    for l in range(4):
        for poly in polys[l]:
            val = sum(c * v[:, 0] ** e[0] * v[:, 1] ** e[1] * v[:, 2] ** e[2] for e, c in poly.items())
            assert np.abs(val - ref[:, k]).max() < 1e-12, (l, k)
            k += 1
