"""TEST INFRASTRUCTURE (imported by tests/ only): fp64 restatement of the reference's DFT-D3 dispersion term
(sevenn/pair_e3gnn/pair_d3_for_ase.cu, the CUDA library behind sevenn.calculator.D3Calculator, calculator.py:387-618).

Energy as the reference's kernels define it, derivatives by torch autograd (forces = -dE/dx, stress = dE/d(strain) / V):
  coordination number   CN_i = sum_{j,tau}' 1 / (1 + exp(-K1 ((rcov_i + rcov_j) / r - 1)))     r^2 <= cn_cutoff   (:1004-1057)
  C6_ij(CN_i, CN_j)     Gaussian-weighted average of the reference C6 values, K3 = -4                              (:765-845)
  two-body energy       E = -1/2 sum_{i,j,tau}' C6_ij [s6 f6(r) / r^6 + 3 s8 r2r4_i r2r4_j f8(r) / r^8], r^2 <= vdw_cutoff
      damp_bj:  f_n / r^n -> 1 / (r^n + R0^n),  R0 = a1 sqrt(3 r2r4_i r2r4_j) + a2                                (:1534-1694)
      damp_zero: f_n = 1 / (1 + 6 (rs_n r0ab / r)^alp_n)                                                           (:1263-1496)
  lattice translations  every tau = n1 a1 + n2 a2 + n3 a3 with |n_k| <= int(r_cut / height_k) + 1 (0 if not periodic),
                        atoms wrapped into the cell first                                            (:979-1001,1170-1219)
  units                 lengths in bohr (0.52917726 A), energies in hartree (27.21138505 eV); cutoffs in bohr^2.
Pinned by the reference's own known answers (tests/unit_tests/test_calculator.py:192-236): tests/test_d3_cpu.py."""
import os

import numpy as np
import torch

AU_TO_ANG = 0.52917726
AU_TO_EV = 27.21138505
K1, K3 = 16.0, -4.0
_BLOB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sevennet_amd', 'data', 'd3_params.npz')


class D3Params:
    def __init__(self, path=_BLOB):
        z = np.load(path)
        self.r0ab, self.r2r4, self.rcov = z['r0ab'], z['r2r4'], z['rcov']
        self.func = {d: dict(zip(z[d + '_names'].tolist(), z[d + '_params'])) for d in ('damp_zero', 'damp_bj', 'damp_zerom', 'damp_bjm')}
        # reference C6 grid per element pair: c6ref[Zi, Zj] -> [5, 5, 3] (c6, cn_i, cn_j); mxc[Z] = number of references
        t = z['c6ab']
        self.c6ref = np.zeros((95, 95, 5, 5, 3))
        self.mxc = np.zeros(95, np.int64)
        a1, a2 = t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
        zi, ri = (a1 - 1) % 100 + 1, (a1 - 1) // 100      # pair_d3_for_ase.cu:340-346
        zj, rj = (a2 - 1) % 100 + 1, (a2 - 1) // 100
        for k in range(len(t)):
            self.c6ref[zi[k], zj[k], ri[k], rj[k]] = (t[k, 0], t[k, 3], t[k, 4])
            self.c6ref[zj[k], zi[k], rj[k], ri[k]] = (t[k, 0], t[k, 4], t[k, 3])
            self.mxc[zi[k]] = max(self.mxc[zi[k]], ri[k] + 1)
            self.mxc[zj[k]] = max(self.mxc[zj[k]], rj[k] + 1)

    def functional(self, damping, name):
        s6, rs6, s18, rs18, alp = self.func[damping][name]
        return dict(s6=s6, a1=rs6, s8=s18, a2=rs18, alp6=alp, alp8=alp + 2.0)   # setfuncpar, :608-628


def translations(cell_au, pbc, r2_cut):
    """integer repetitions (set_lattice_repetition_criteria) and the translation vectors [T,3] in bohr"""
    a = np.asarray(cell_au, np.float64)
    rc = np.sqrt(r2_cut)
    reps = []
    for k in range(3):
        cp = np.cross(a[(k + 1) % 3], a[(k + 2) % 3])
        h = abs(np.dot(cp, a[k]) / np.linalg.norm(cp))
        reps.append(int(abs(rc / h)) + 1 if pbc[k] else 0)
    g = np.stack(np.meshgrid(*[np.arange(-r, r + 1) for r in reps], indexing='ij'), -1).reshape(-1, 3)
    return g @ a, g


def d3(numbers, positions, cell, pbc, damping='damp_bj', functional='pbe', vdw_cutoff=9000.0, cn_cutoff=1600.0, params=None):
    """-> dict(energy eV, forces [N,3] eV/A, stress [3,3] eV/A^3 (dE/d strain / volume), cn [N], c6 [N,N] hartree bohr^6)"""
    P = params or D3Params()
    fp = P.functional(damping, functional)
    Z = np.asarray(numbers, np.int64)
    n = len(Z)
    cell_au = np.asarray(cell, np.float64).reshape(3, 3) / AU_TO_ANG
    frac = (np.asarray(positions, np.float64) / AU_TO_ANG) @ np.linalg.inv(cell_au)
    frac -= np.floor(frac)                                           # load_atom_info: wrap into the cell
    x0 = torch.tensor(frac @ cell_au, dtype=torch.float64, requires_grad=True)
    eps = torch.zeros(3, 3, dtype=torch.float64, requires_grad=True)
    strain = torch.eye(3, dtype=torch.float64) + 0.5 * (eps + eps.T)
    x = x0 @ strain
    cellt = torch.tensor(cell_au) @ strain
    tv, gv = translations(cell_au, pbc, vdw_cutoff)
    tc, gc = translations(cell_au, pbc, cn_cutoff)

    def pair_r2(g):
        tau = torch.tensor(g, dtype=torch.float64) @ cellt                                    # [T,3]
        d = x[None, :, None, :] - x[:, None, None, :] + tau[None, None, :, :]                    # [i,j,T,3] = x_j - x_i + tau
        r2 = (d * d).sum(-1)
        self_img = torch.eye(n, dtype=torch.bool)[:, :, None] & torch.tensor((g == 0).all(1))[None, None, :]
        return r2, ~self_img

    rcov = torch.tensor(P.rcov[Z - 1])
    r2, ok = pair_r2(gc)
    m = ok & (r2.detach() <= cn_cutoff)
    r = torch.sqrt(torch.where(m, r2, torch.ones_like(r2)))
    cnt = 1.0 / (1.0 + torch.exp(-K1 * ((rcov[:, None, None] + rcov[None, :, None]) / r - 1.0)))
    cn = torch.where(m, cnt, torch.zeros_like(cnt)).sum((1, 2))                                  # [N]

    c6 = torch.zeros(n, n, dtype=torch.float64)
    rows = []
    for i in range(n):
        row = []
        for j in range(n):
            ref = torch.tensor(P.c6ref[Z[i], Z[j], :P.mxc[Z[i]], :P.mxc[Z[j]]])                  # [a,b,3]
            use = ref[..., 0] > 0
            rr = (ref[..., 1] - cn[i]) ** 2 + (ref[..., 2] - cn[j]) ** 2
            w = torch.where(use, torch.exp(K3 * rr), torch.zeros_like(rr))
            den = w.sum()
            if float(den.detach()) > 1e-99:
                row.append((w * ref[..., 0]).sum() / den)
            else:                                                                                 # :833-837: nearest reference
                row.append(ref[..., 0][use][torch.argmin(rr[use])])
        rows.append(torch.stack(row))
    c6 = torch.stack(rows)

    r2, ok = pair_r2(gv)
    m = ok & (r2.detach() <= vdw_cutoff)
    r2s = torch.where(m, r2, torch.ones_like(r2))
    r2r4 = torch.tensor(P.r2r4[Z - 1])
    r42 = (r2r4[:, None] * r2r4[None, :])[:, :, None]
    if damping in ('damp_bj', 'damp_bjm'):
        R0 = fp['a1'] * torch.sqrt(3.0 * r42) + fp['a2']
        phi = fp['s6'] / (r2s ** 3 + R0 ** 6) + fp['s8'] * 3.0 * r42 / (r2s ** 4 + R0 ** 8)
    else:
        r = torch.sqrt(r2s)
        r0 = torch.tensor(P.r0ab[Z - 1][:, Z - 1] / AU_TO_ANG)[:, :, None]
        f6 = 1.0 / (1.0 + 6.0 * (fp['a1'] * r0 / r) ** fp['alp6'])
        f8 = 1.0 / (1.0 + 6.0 * (fp['a2'] * r0 / r) ** fp['alp8'])
        phi = fp['s6'] * f6 / r2s ** 3 + fp['s8'] * 3.0 * r42 * f8 / r2s ** 4
    e = -0.5 * (c6[:, :, None] * torch.where(m, phi, torch.zeros_like(phi))).sum()
    gx, ge = torch.autograd.grad(e, (x0, eps))
    vol = abs(np.linalg.det(cell_au))
    return dict(energy=float(e.detach()) * AU_TO_EV, forces=-gx.numpy() * AU_TO_EV / AU_TO_ANG,
                stress=ge.numpy() * AU_TO_EV / (vol * AU_TO_ANG ** 3), cn=cn.detach().numpy(), c6=c6.detach().numpy())
