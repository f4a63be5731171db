"""Device-side graph build: positions -> CSR edge list ready for HipForceEngine.compute.

GPU counterpart of `sevennet_amd.neighbor.neighbor_list` + `engine.build_graph`: any mix of periodic and open axes,
cells thinner than the cutoff (they meet themselves through several images), molecules without a cell.
Reference conventions: sevenn/train/dataload.py:32-129 (every ordered pair, no self edge,
edge_vec = r_j - r_i + S.cell computed in fp64, stored fp32; open axes: the cell row is irrelevant -- the reference pads
it, :37-48 -- atoms are neither wrapped nor imaged along it).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .engine import Graph


def _domain(cell, pbc, cutoff: float, pos):
    """(cell[3,3] with degenerate open-axis rows padded like the host list, pbc int32[3], fractional range [6] of the atoms)"""
    cell = np.array(cell, np.float64).reshape(3, 3)
    pbc = np.asarray(pbc, bool).reshape(3)
    if not pbc.all():
        pmax = float(pos.abs().max()) if isinstance(pos, torch.Tensor) else (float(np.abs(pos).max()) if len(pos) else 0.0)
        for k in range(3):
            if not pbc[k] and np.linalg.norm(cell[k]) < 1e-12:   # dataload.py:37-48
                cell[k] = 0.0
                cell[k, k] = (pmax + 1.0) * 5 * cutoff
    rng = np.zeros(6, np.float64)
    rng[3:] = 1.0
    if not pbc.all() and abs(np.linalg.det(cell)) > 1e-12 and len(pos):
        inv = np.linalg.inv(cell)
        if isinstance(pos, torch.Tensor):
            f = pos.to(torch.float64) @ torch.as_tensor(inv, device=pos.device)
            lo, hi = f.min(0).values.cpu().numpy(), f.max(0).values.cpu().numpy()
        else:
            f = np.asarray(pos, np.float64) @ inv
            lo, hi = f.min(0), f.max(0)
        rng[:3], rng[3:] = lo, hi
    return np.ascontiguousarray(cell), np.ascontiguousarray(pbc.astype(np.int32)), rng


def gpu_neighbor_supported(cell, pbc, cutoff: float, pos=None) -> bool:
    """True when the device cell list handles this domain: every non-singular cell (after the reference's padding of
    degenerate open axes) whose periodic heights are at least 1/64 of the cutoff"""
    cell_np, pbc_np, rng = _domain(cell, pbc, cutoff, pos if pos is not None else np.zeros((0, 3)))
    if abs(np.linalg.det(cell_np)) < 1e-12:
        return False
    nb = (C.c_int32 * 3)()
    return _lib.load().snet_nl_grid(cell_np.ctypes.data_as(C.POINTER(C.c_double)), float(cutoff),
                                    pbc_np.ctypes.data_as(C.POINTER(C.c_int32)), rng.ctypes.data_as(C.POINTER(C.c_double)), nb) == 0


def build_graph_gpu(types, pos, cell, cutoff: float, device='cuda:0', num_species: int = 0,
                    with_shifts: bool = False, share_pairs: bool = True, pbc=(True, True, True)) -> Graph:
    """All edges with |r_j - r_i + S.cell| < cutoff (S over the periodic axes only), as a device Graph."""
    lib = _lib.load()
    dev = torch.device(device)
    cell_np, pbc_np, rng = _domain(cell, pbc, cutoff, pos)
    cp = cell_np.ctypes.data_as(C.POINTER(C.c_double))
    pp = pbc_np.ctypes.data_as(C.POINTER(C.c_int32))
    rp = rng.ctypes.data_as(C.POINTER(C.c_double))
    nb = (C.c_int32 * 3)()
    _lib.check(lib.snet_nl_grid(cp, float(cutoff), pp, rp, nb), 'snet_nl_grid')
    nbins = int(nb[0]) * int(nb[1]) * int(nb[2])
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = torch.as_tensor(np.asarray(pos, np.float64)).to(dev) if not isinstance(pos, torch.Tensor) \
            else pos.to(dev, torch.float64)
        p = p.contiguous()
        n = int(p.shape[0])
        ty = torch.as_tensor(types).to(dev, torch.int32)
        wpos = torch.empty_like(p)
        wrap = torch.empty(n, 3, dtype=torch.int32, device=dev)
        cid = torch.empty(n, dtype=torch.int32, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        _lib.check(lib.snet_nl_bin(cp, float(cutoff), pp, rp, P(p), n, P(wpos), P(wrap), P(cid), st), 'snet_nl_bin')
        order = torch.sort(cid.long(), stable=True).indices.to(torch.int32)
        bin_start = torch.zeros(nbins + 1, dtype=torch.int64, device=dev)
        bin_start[1:] = torch.cumsum(torch.bincount(cid.long(), minlength=nbins), 0)
        bin_start = bin_start.to(torch.int32)
        count = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(lib.snet_nl_count(cp, float(cutoff), pp, rp, P(wpos), P(cid), P(order), P(bin_start), n, P(count), st),
                   'snet_nl_count')
        row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        row_ptr[1:] = torch.cumsum(count.long(), 0)
        E = int(row_ptr[-1].item())
        if E >= 2 ** 31:
            raise ValueError('more than 2^31 edges')
        row_ptr = row_ptr.to(torch.int32)
        src = torch.empty(E, dtype=torch.int32, device=dev)
        center = torch.empty(E, dtype=torch.int32, device=dev)
        ev = torch.empty(E, 3, dtype=torch.float32, device=dev)
        shifts: Optional[torch.Tensor] = torch.empty(E, 3, dtype=torch.int32, device=dev) if with_shifts else None
        _lib.check(lib.snet_nl_fill(cp, float(cutoff), pp, rp, P(wpos), P(wrap), P(cid), P(order), P(bin_start), n, P(row_ptr),
                                    P(src), P(center), P(ev), None if shifts is None else P(shifts), st), 'snet_nl_fill')
        col_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if E:
            col_ptr[1:] = torch.cumsum(torch.bincount(src.long(), minlength=n), 0)
            eperm = torch.sort(src.long(), stable=True).indices.to(torch.int32)
        else:
            eperm = torch.zeros(0, dtype=torch.int32, device=dev)
        rows = None
        if num_species:
            from .engine import species_row_lists
            rows = species_row_lists(ty, num_species)
        g = Graph(n, n, E, ty, center, src, row_ptr, col_ptr.to(torch.int32), eperm, ev, None, rows)
        g.shifts = shifts
        return g.share_pairs() if share_pairs else g
