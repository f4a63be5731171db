"""Irreps bookkeeping and real Clebsch-Gordan tables for the HIP force engine.

Host-side mirror of the e3nn conventions SevenNet builds on (e3nn is a
third-party dependency of the reference, pinned `e3nn>=0.5.0`,
pyproject.toml:24).  Used by the kernel generator (`codegen.py`) and the model
description (`model_spec.py`).  Conventions restated from the reference's call
sites:
  * block order / sort key (l, p) with odd before even: sevenn/nn/cue_helper.py:41-45
  * feature layout at the reference boundary is `mul_ir`; the engine's internal
    layout is `ir_mul` (component-major: [2l+1][mul]) so that a wavefront's 64
    lanes read 64 consecutive channels.
"""
from __future__ import annotations

import math
import re
from functools import lru_cache
from typing import List, Tuple

import numpy as np


class Irreps:
    """List of (mul, l, p).  p = +1 (even) / -1 (odd)."""

    def __init__(self, spec=()):
        if isinstance(spec, Irreps):
            self.blocks = list(spec.blocks)
        elif isinstance(spec, str):
            self.blocks = []
            for tok in filter(None, (t.strip() for t in spec.split('+'))):
                m = re.fullmatch(r'(?:(\d+)x)?(\d+)([eo])', tok)
                if m is None:
                    raise ValueError(f'cannot parse irreps token {tok!r}')
                self.blocks.append((int(m.group(1) or 1), int(m.group(2)), 1 if m.group(3) == 'e' else -1))
        else:
            self.blocks = []
            for b in spec:
                if len(b) == 2:  # (mul, (l, p))
                    self.blocks.append((int(b[0]), int(b[1][0]), int(b[1][1])))
                else:
                    self.blocks.append((int(b[0]), int(b[1]), int(b[2])))

    def __iter__(self):
        return iter(self.blocks)

    def __len__(self):
        return len(self.blocks)

    def __getitem__(self, i):
        return self.blocks[i]

    def __eq__(self, other):
        return self.blocks == Irreps(other).blocks

    def __repr__(self):
        return '+'.join(f"{m}x{l}{'e' if p > 0 else 'o'}" for m, l, p in self.blocks)

    @property
    def dim(self):
        return sum(m * (2 * l + 1) for m, l, _ in self.blocks)

    @property
    def lmax(self):
        return max(l for _, l, _ in self.blocks)

    def offsets(self) -> List[int]:
        o, out = 0, []
        for m, l, _ in self.blocks:
            out.append(o)
            o += m * (2 * l + 1)
        return out

    def has(self, l, p) -> bool:
        return any(bl == l and bp == p for _, bl, bp in self.blocks)

    def sorted(self):
        """Stable sort by (l, p); returns (Irreps, perm) with perm[old] = new."""
        order = sorted(range(len(self.blocks)), key=lambda i: (self.blocks[i][1], self.blocks[i][2]))
        perm = [0] * len(order)
        for new, old in enumerate(order):
            perm[old] = new
        return Irreps([self.blocks[i] for i in order]), perm

    def simplified(self):
        out = []
        for m, l, p in self.blocks:
            if m == 0:
                continue
            if out and out[-1][1] == l and out[-1][2] == p:
                out[-1] = (out[-1][0] + m, l, p)
            else:
                out.append((m, l, p))
        return Irreps(out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, l, p ** l) for l in range(lmax + 1)])


def infer_irreps_out(x: Irreps, operand: Irreps, drop_l=False, parity_mode='full', fix_multiplicity=False):
    """Output irreps of one interaction (rule of sevenn/util.py:199-221)."""
    prod = []
    for m1, l1, p1 in x:
        for m2, l2, p2 in operand:
            for l in range(abs(l1 - l2), l1 + l2 + 1):
                prod.append((m1 * m2, l, p1 * p2))
    full = Irreps(prod).sorted()[0].simplified()
    out = []
    for m, l, p in full:
        if drop_l is not False and l > drop_l:
            continue
        if parity_mode == 'even' and p == -1:
            continue
        if parity_mode == 'sph' and p != (-1) ** l:
            continue
        out.append((fix_multiplicity if fix_multiplicity else m, l, p))
    return Irreps(out)


# --------------------------------------------------------------------------- #
# layout conversion  mul_ir (reference boundary)  <->  ir_mul (engine)
# --------------------------------------------------------------------------- #
def mulir_to_irmul_index(irreps: Irreps) -> np.ndarray:
    """idx such that x_irmul = x_mulir[:, idx]."""
    idx, o = [], 0
    for m, l, _ in irreps:
        d = 2 * l + 1
        blk = np.arange(m * d).reshape(m, d).T.reshape(-1) + o
        idx.append(blk)
        o += m * d
    return np.concatenate(idx) if idx else np.zeros(0, np.int64)


def irmul_to_mulir_index(irreps: Irreps) -> np.ndarray:
    f = mulir_to_irmul_index(irreps)
    inv = np.empty_like(f)
    inv[f] = np.arange(f.size)
    return inv


# --------------------------------------------------------------------------- #
# real Wigner-3j (float64), e3nn basis: l=1 components are (x, y, z) with y polar
# --------------------------------------------------------------------------- #
def _cg_complex(j1, j2, j3):
    """<j1 m1 j2 m2|j3 m3> for integer j, log-factorial evaluation."""
    lf = [0.0]
    for n in range(1, 4 * (j1 + j2 + j3) + 8):
        lf.append(lf[-1] + math.log(n))
    out = np.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1))
    if not abs(j1 - j2) <= j3 <= j1 + j2:
        return out
    tri = lf[j1 + j2 - j3] + lf[j1 - j2 + j3] + lf[-j1 + j2 + j3] - lf[j1 + j2 + j3 + 1]
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            m3 = m1 + m2
            if abs(m3) > j3:
                continue
            pre = 0.5 * (math.log(2 * j3 + 1) + tri + lf[j3 + m3] + lf[j3 - m3] + lf[j1 - m1]
                         + lf[j1 + m1] + lf[j2 - m2] + lf[j2 + m2])
            s = 0.0
            for k in range(0, j1 + j2 - j3 + 1):
                den = (k, j1 + j2 - j3 - k, j1 - m1 - k, j2 + m2 - k, j3 - j2 + m1 + k, j3 - j1 - m2 + k)
                if min(den) < 0:
                    continue
                s += (-1) ** k * math.exp(pre - sum(lf[d] for d in den))
            out[j1 + m1, j2 + m2, j3 + m3] = s
    return out


def _basis_change(l):
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=complex)
    r = math.sqrt(0.5)
    for m in range(1, l + 1):
        q[l - m, l + m] = r
        q[l - m, l - m] = -1j * r
        q[l + m, l + m] = (-1) ** m * r
        q[l + m, l - m] = 1j * (-1) ** m * r
    q[l, l] = 1.0
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def real_wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    c = _cg_complex(l1, l2, l3).astype(complex)
    t = np.einsum('ia,kb,nc,ikn->abc', _basis_change(l1), _basis_change(l2),
                  np.conj(_basis_change(l3)), c)
    assert np.abs(t.imag).max() < 1e-10
    t = t.real.copy()
    t[np.abs(t) < 1e-13] = 0.0
    n = np.linalg.norm(t)
    return t / n if n > 0 else t


def cg_nonzeros(l1, l2, l3) -> List[Tuple[int, int, int, float]]:
    """Sparse (m1, m2, m3, value) list of the real 3j tensor."""
    t = real_wigner_3j(l1, l2, l3)
    return [(int(a), int(b), int(c), float(t[a, b, c])) for a, b, c in zip(*np.nonzero(t))]


# --------------------------------------------------------------------------- #
# spherical-harmonic polynomials (component normalisation) as sparse monomials
# --------------------------------------------------------------------------- #
@lru_cache(maxsize=None)
def sh_polynomials(lmax: int):
    """For each l a list (one entry per m) of {(ax, ay, az): coeff} giving the
    homogeneous degree-l polynomial Y_lm(x, y, z) with ||Y_l(unit)||^2 = 2l+1
    (e3nn normalization='component').  Built by the recursion
    Y_{l+1} ~ C(l,1,l+1) Y_l r, fixed positive at the polar axis."""
    def pmul(p, mono, c):
        out = {}
        for k, v in p.items():
            kk = (k[0] + mono[0], k[1] + mono[1], k[2] + mono[2])
            out[kk] = out.get(kk, 0.0) + v * c
        return out

    def padd(a, b):
        out = dict(a)
        for k, v in b.items():
            out[k] = out.get(k, 0.0) + v
        return out

    def peval(p, v):
        return sum(c * v[0] ** k[0] * v[1] ** k[1] * v[2] ** k[2] for k, c in p.items())

    unit = [[{(0, 0, 0): 1.0}]]
    if lmax >= 1:
        unit.append([{(1, 0, 0): 1.0}, {(0, 1, 0): 1.0}, {(0, 0, 1): 1.0}])
    pole = (0.0, 1.0, 0.0)
    monos = [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
    for l in range(1, lmax):
        w = real_wigner_3j(l, 1, l + 1)
        nxt = []
        for k in range(2 * l + 3):
            acc = {}
            for a in range(2 * l + 1):
                for j in range(3):
                    if w[a, j, k] != 0.0:
                        acc = padd(acc, pmul(unit[l][a], monos[j], w[a, j, k]))
            nxt.append(acc)
        nrm = math.sqrt(sum(peval(p, pole) ** 2 for p in nxt))
        nxt = [{k: v / nrm for k, v in p.items() if abs(v) > 1e-14} for p in nxt]
        unit.append(nxt)
    return [[{k: v * math.sqrt(2 * l + 1) for k, v in p.items()} for p in unit[l]] for l in range(lmax + 1)]
