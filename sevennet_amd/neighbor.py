"""Host-side periodic neighbor list (graph build that precedes the hot path).

Conventions follow the reference's graph build, sevenn/train/dataload.py:32-79,
102-129 (ASE/matscipy 'ijDS'): every ordered pair within the cutoff, no
self-edge, `edge_index[0]` = center atom i, `edge_index[1]` = neighbor j,
`edge_vec = r_j - r_i + S @ cell`.

Implementation: explicit image atoms inside a one-cutoff skin + a KD-tree
(scipy, C speed) -- valid for triclinic cells and cells smaller than the cutoff.
The edge list is returned sorted by center atom (CSR order), which is the layout
the HIP convolution kernels consume.
"""
from __future__ import annotations

import itertools

import numpy as np
from scipy.spatial import cKDTree


def neighbor_list(pos, cell, pbc, cutoff: float):
    """Returns (edge_index[2,E] int64, edge_vec[E,3] float64, shifts[E,3] int64)."""
    pos = np.asarray(pos, dtype=np.float64)
    n = pos.shape[0]
    pbc = np.asarray(pbc, dtype=bool).reshape(3)
    cell = np.array(cell, dtype=np.float64).reshape(3, 3)
    if not pbc.all():
        # pad non-periodic axes like dataload.py:37-48 so that inverse exists
        ext = (np.abs(pos).max() if n else 0.0) + 1.0
        for k in range(3):
            if not pbc[k] and np.linalg.norm(cell[k]) < 1e-12:
                cell[k] = 0.0
                cell[k, k] = ext * 5 * cutoff
    if n == 0:
        return np.zeros((2, 0), np.int64), np.zeros((0, 3)), np.zeros((0, 3), np.int64)

    inv = np.linalg.inv(cell)
    frac = pos @ inv
    wrap = np.zeros_like(frac)
    wrap[:, pbc] = np.floor(frac[:, pbc])
    fw = frac - wrap  # in [0,1) along periodic axes
    pw = fw @ cell

    vol = abs(np.linalg.det(cell))
    # heights between opposite cell faces
    h = np.array([vol / np.linalg.norm(np.cross(cell[(k + 1) % 3], cell[(k + 2) % 3])) for k in range(3)])
    skin = cutoff / h  # fractional skin width
    reps = [int(np.ceil(skin[k])) if pbc[k] else 0 for k in range(3)]

    img_pos, img_idx, img_shift = [], [], []
    for s in itertools.product(*[range(-r, r + 1) for r in reps]):
        s = np.array(s)
        f = fw + s
        ok = np.ones(n, dtype=bool)
        for k in range(3):
            if pbc[k]:
                ok &= (f[:, k] >= -skin[k] - 1e-9) & (f[:, k] <= 1.0 + skin[k] + 1e-9)
        ids = np.nonzero(ok)[0]
        if ids.size == 0:
            continue
        img_pos.append(pw[ids] + s @ cell)
        img_idx.append(ids)
        img_shift.append(np.broadcast_to(s, (ids.size, 3)))
    img_pos = np.concatenate(img_pos)
    img_idx = np.concatenate(img_idx)
    img_shift = np.concatenate(img_shift)

    tree_c = cKDTree(pw)
    tree_n = cKDTree(img_pos)
    coo = tree_c.sparse_distance_matrix(tree_n, cutoff, output_type='coo_matrix')
    i = coo.row.astype(np.int64)
    jj = coo.col.astype(np.int64)
    j = img_idx[jj]
    S = img_shift[jj]
    vec = img_pos[jj] - pw[i]
    d2 = np.einsum('ij,ij->i', vec, vec)
    keep = (d2 < cutoff * cutoff) & ~((i == j) & (S == 0).all(axis=1))
    i, j, S, vec = i[keep], j[keep], S[keep], vec[keep]
    # express shifts relative to the caller's (unwrapped) positions
    S = S + (wrap[i] - wrap[j]).astype(np.int64)
    order = np.lexsort((j, i))
    i, j, S, vec = i[order], j[order], S[order], vec[order]
    return np.stack([i, j]), vec, S.astype(np.int64)


def diamond_cubic(a: float, reps, sigma: float = 0.0, seed: int = 0):
    """Synthetic periodic diamond-structure cell (SURVEY.md §8d workloads):
    conventional 8-atom cell of lattice constant `a` replicated `reps` times,
    Gaussian rattle `sigma` (Angstrom)."""
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    reps = np.asarray(reps, dtype=np.int64).reshape(3)
    g = np.stack(np.meshgrid(*[np.arange(r) for r in reps], indexing='ij'), -1).reshape(-1, 3)
    pos = ((g[:, None, :] + basis[None, :, :]).reshape(-1, 3)) * a
    cell = np.diag(reps * a).astype(np.float64)
    if sigma > 0:
        rng = np.random.default_rng(seed)
        pos = pos + rng.normal(0.0, sigma, pos.shape)
    return pos, cell


def amorphous_cell(a: float, reps, sigma: float, seed: int, min_dist: float = 1.8, max_rounds: int = 400):
    """diamond sites with a melt-like Gaussian disorder `sigma` (0.35 A) and a
    minimum-distance reject -- every atom closer than `min_dist` to another one gets a fresh displacement from
    its lattice site until no such pair is left (periodic KD-tree search)."""
    from scipy.spatial import cKDTree
    site, cell = diamond_cubic(a, reps, 0.0, 0)
    box = np.diag(cell).copy()
    rng = np.random.default_rng(seed)
    pos = site + rng.normal(0.0, sigma, site.shape)
    for k in range(max_rounds):
        w = np.mod(pos, box)
        pairs = cKDTree(w, boxsize=box).query_pairs(min_dist, output_type='ndarray')
        if len(pairs) == 0:
            return pos, cell
        bad = np.unique(pairs[:, 1] if k < 60 else pairs)   # one partner of every close pair; both once the rest is stubborn
        # the last few stubborn atoms (all neighbors displaced towards them) are re-drawn from a narrower Gaussian
        s_k = sigma * (0.8 ** max(0, (k - 60) // 20))
        pos[bad] = site[bad] + rng.normal(0.0, s_k, (len(bad), 3))
    raise RuntimeError('amorphous_cell: could not satisfy the minimum distance')
