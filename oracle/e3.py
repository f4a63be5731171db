"""Irreps algebra, real Wigner-3j and spherical harmonics (oracle side).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the e3nn conventions the reference relies on at
  sevenn/nn/convolution.py:61-82   (irreps_mid sort, instruction order)
  sevenn/util.py:199-221           (infer_irreps_out via FullTensorProduct)
  sevenn/nn/edge_embedding.py:164-185 (SphericalHarmonics, 'component' norm)
  sevenn/nn/cue_helper.py:36-45    (wigner_3j + sort rule `(l, p)`, odd first)
The generators are pinned against data in tests/golden (see make_golden.py).
"""
from __future__ import annotations

import math
import re
from fractions import Fraction
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np
import torch

Irrep = Tuple[int, int]  # (l, parity) with parity in {+1, -1}


# --------------------------------------------------------------------------- #
# Irreps
# --------------------------------------------------------------------------- #
class Irreps:
    """Ordered list of (mul, (l, p)); feature layout is e3nn `mul_ir`."""

    def __init__(self, spec=()):
        if isinstance(spec, Irreps):
            self.items = list(spec.items)
        elif isinstance(spec, str):
            self.items = []
            spec = spec.strip()
            if spec:
                for tok in spec.split('+'):
                    m = re.fullmatch(r'\s*(?:(\d+)x)?(\d+)([eo])\s*', tok)
                    if not m:
                        raise ValueError(f'bad irreps token {tok!r}')
                    mul = int(m.group(1)) if m.group(1) else 1
                    self.items.append((mul, (int(m.group(2)), 1 if m.group(3) == 'e' else -1)))
        else:
            self.items = [(int(mul), (int(ir[0]), int(ir[1]))) for mul, ir in spec]

    # -- basic protocol ----------------------------------------------------- #
    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def __add__(self, other):
        return Irreps(self.items + Irreps(other).items)

    def __eq__(self, other):
        return self.items == Irreps(other).items

    def __contains__(self, ir: Irrep):
        return any(i == tuple(ir) for _, i in self.items)

    def __repr__(self):
        return '+'.join(f"{m}x{l}{'e' if p == 1 else 'o'}" for m, (l, p) in self.items)

    @property
    def dim(self) -> int:
        return sum(m * (2 * l + 1) for m, (l, _) in self.items)

    @property
    def num_irreps(self) -> int:
        return sum(m for m, _ in self.items)

    @property
    def lmax(self) -> int:
        return max(l for _, (l, _) in self.items)

    def slices(self) -> List[slice]:
        out, o = [], 0
        for m, (l, _) in self.items:
            out.append(slice(o, o + m * (2 * l + 1)))
            o += m * (2 * l + 1)
        return out

    # -- e3nn semantics ----------------------------------------------------- #
    @staticmethod
    def _key(ir: Irrep):
        # e3nn Irrep ordering: by l, then odd (-1) before even (+1)
        return (ir[0], ir[1])

    def sort(self):
        """Stable sort by (l, p).  Returns (irreps, p, inv) like e3nn:
        p[i] = new position of old block i; inv[j] = old index at new pos j."""
        inv = sorted(range(len(self.items)), key=lambda i: self._key(self.items[i][1]))
        p = [0] * len(inv)
        for new, old in enumerate(inv):
            p[old] = new
        return Irreps([self.items[i] for i in inv]), p, inv

    def simplify(self):
        out = []
        for m, ir in self.items:
            if m == 0:
                continue
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + m, ir)
            else:
                out.append((m, ir))
        return Irreps(out)

    @staticmethod
    def spherical_harmonics(lmax: int, p: int = -1) -> 'Irreps':
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])


def irrep_product(a: Irrep, b: Irrep) -> List[Irrep]:
    return [(l, a[1] * b[1]) for l in range(abs(a[0] - b[0]), a[0] + b[0] + 1)]


def full_tensor_product_irreps(x: Irreps, y: Irreps) -> Irreps:
    """irreps_out of e3nn FullTensorProduct (sorted), before simplify."""
    out = []
    for m1, ir1 in x:
        for m2, ir2 in y:
            for ir in irrep_product(ir1, ir2):
                out.append((m1 * m2, ir))
    return Irreps(out).sort()[0]


def infer_irreps_out(x: Irreps, operand: Irreps, drop_l=False, parity_mode='full',
                     fix_multiplicity=False) -> Irreps:
    """Follows sevenn/util.py:199-221."""
    assert parity_mode in ('full', 'even', 'sph')
    full = full_tensor_product_irreps(x, operand).simplify()
    elems = []
    for mul, (l, p) in full:
        if drop_l is not False and l > drop_l:
            continue
        if parity_mode == 'even' and p == -1:
            continue
        if parity_mode == 'sph' and p != (-1) ** l:
            continue
        elems.append((fix_multiplicity if fix_multiplicity else mul, (l, p)))
    return Irreps(elems)


# --------------------------------------------------------------------------- #
# Wigner 3j in e3nn's real basis
# --------------------------------------------------------------------------- #
def _fact(n) -> int:
    n = int(round(n))
    assert n >= 0
    return math.factorial(n)


def _su2_cg(j1, m1, j2, m2, j3, m3) -> float:
    """Condon-Shortley <j1 m1 j2 m2 | j3 m3> (Racah's closed form), exact
    rational part via Fractions, one final sqrt."""
    if m3 != m1 + m2 or not (abs(j1 - j2) <= j3 <= j1 + j2):
        return 0.0
    pref = Fraction(
        (2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3),
        _fact(j1 + j2 + j3 + 1),
    ) * Fraction(
        _fact(j3 + m3) * _fact(j3 - m3) * _fact(j1 - m1) * _fact(j1 + m1)
        * _fact(j2 - m2) * _fact(j2 + m2), 1)
    s = Fraction(0)
    for k in range(0, j1 + j2 - j3 + 1):
        d = [k, j1 + j2 - j3 - k, j1 - m1 - k, j2 + m2 - k, j3 - j2 + m1 + k, j3 - j1 - m2 + k]
        if min(d) < 0:
            continue
        den = 1
        for t in d:
            den *= _fact(t)
        s += Fraction((-1) ** k, den)
    return float(s) * math.sqrt(float(pref))


def _real_to_complex(l: int) -> np.ndarray:
    """Unitary taking e3nn's real (y,z,x)-ordered basis to complex |l m>,
    with the global (-i)^l phase that makes the coupled coefficients real."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    s = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s
        q[l + m, l - abs(m)] = -1j * s
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _w3j_np(l1: int, l2: int, l3: int) -> np.ndarray:
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                C[l1 + m1, l2 + m2, l3 + m3] = _su2_cg(l1, m1, l2, m2, l3, m3)
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    R = np.einsum('ij,kl,mn,ikn->jlm', Q1, Q2, np.conj(Q3.T), C.astype(np.complex128))
    assert np.abs(R.imag).max() < 1e-9, (l1, l2, l3)
    R = R.real
    R[np.abs(R) < 1e-14] = 0.0
    return R / np.linalg.norm(R)


def wigner_3j(l1: int, l2: int, l3: int, dtype=torch.float64) -> torch.Tensor:
    """Real, unit-Frobenius-norm 3j tensor of shape (2l1+1, 2l2+1, 2l3+1)."""
    return torch.tensor(_w3j_np(l1, l2, l3), dtype=dtype)


# --------------------------------------------------------------------------- #
# Spherical harmonics, normalization='component'
# --------------------------------------------------------------------------- #
def _sh_unit_norm(lmax: int, v: torch.Tensor) -> List[torch.Tensor]:
    """Homogeneous harmonic polynomials Y_l(v) with ||Y_l|| = ||v||^l.

    Y_0 = 1, Y_1 = v, Y_{l+1} = c_l * sum_{a,j} w3j(l,1,l+1)[a,j,:] Y_l[a] v[j]
    with c_l > 0 fixed by the norm condition (verified against
    the stored w3j_112 for l=2)."""
    ys = [torch.ones_like(v[..., :1])]
    if lmax >= 1:
        ys.append(v)
    for l in range(1, lmax):
        w = wigner_3j(l, 1, l + 1, dtype=v.dtype).to(v.device)
        y = torch.einsum('ajk,...a,...j->...k', w, ys[l], v)
        # constant: evaluate at the polar axis (0,1,0) where Y_l = e_m0
        pole = torch.zeros(3, dtype=torch.float64)
        pole[1] = 1.0
        yp = [torch.ones(1, dtype=torch.float64), pole]
        for ll in range(1, l + 1):
            ww = wigner_3j(ll, 1, ll + 1)
            t = torch.einsum('ajk,a,j->k', ww, yp[ll], pole)
            yp.append(t / t.norm())
        ww = wigner_3j(l, 1, l + 1)
        c = 1.0 / torch.einsum('ajk,a,j->k', ww, yp[l], pole).norm().item()
        ys.append(y * c)
    return ys


def spherical_harmonics(lmax: int, vec: torch.Tensor, normalize: bool) -> torch.Tensor:
    """e3nn o3.SphericalHarmonics(0..lmax, normalize, normalization='component').

    Output [..., (lmax+1)^2]; ||Y_l(unit)||^2 = 2l+1.  With normalize=False the
    degree-l polynomial is evaluated on the raw vector (old checkpoints)."""
    v = vec
    if normalize:
        v = v / v.norm(dim=-1, keepdim=True)
    ys = _sh_unit_norm(lmax, v)
    return torch.cat([math.sqrt(2 * l + 1) * ys[l] for l in range(lmax + 1)], dim=-1)


# --------------------------------------------------------------------------- #
# normalize2mom constants (e3nn.math.normalize2mom, seed 0, 1e6 fp64 samples)
# --------------------------------------------------------------------------- #
ACT_CST = {
    # values as baked into the reference's deployed TorchScript models
    'silu': 1.6791767923989418,
    'tanh': 1.5937334472592695,
}


def normalize2mom_const(name: str) -> float:
    if name in ACT_CST:
        return ACT_CST[name]
    gen = torch.Generator().manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    f = {'ssp': lambda x: torch.nn.functional.softplus(x) - math.log(2.0),
         'abs': torch.abs, 'relu': torch.relu, 'sigmoid': torch.sigmoid, 'elu': torch.nn.functional.elu}[name]
    return float(f(z).pow(2).mean().pow(-0.5))
