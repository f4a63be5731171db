"""Spatial decomposition + ghost-atom halo exchange.

Scheme of the reference's multi-GPU inference path
(sevenn/pair_e3gnn/pair_e3gnn_parallel.cpp:194-528, comm_brick.cpp:1057-1123):
atoms are partitioned into bricks, one rank per GPU; every rank owns its local
atoms, sees ghosts within ONE cutoff, owns the edges whose center is local, and
exchanges ghost node features once per interaction layer t >= 1 (forward) and
ghost gradients once per layer in the reverse pass (accumulating into owners),
plus one force fold.

MI355X-native differences: instead of <= 6 sequential blocking MPI swaps with
ghost-of-ghost forwarding, every rank talks to each peer directly in ONE grouped
point-to-point exchange per layer (`all_to_all_single` on the RCCL backend =
one ncclGroup of send/recv pairs, every peer on its own xGMI link), packing and
unpacking with HIP kernels on the compute stream.  Ghost rows are laid out
contiguously per peer so the receive lands in place.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from .neighbor import neighbor_list


# --------------------------------------------------------------------------- #
# partitioning (host side; in LAMMPS this information comes from the domain decomposition)
# --------------------------------------------------------------------------- #
def processor_grid(world: int) -> tuple:
    """Near-cubic factorisation px*py*pz = world (2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2)."""
    best = (world, 1, 1)
    for px in range(1, world + 1):
        if world % px:
            continue
        for py in range(1, world // px + 1):
            if (world // px) % py:
                continue
            pz = world // px // py
            cand = tuple(sorted((px, py, pz), reverse=True))
            if max(cand) - min(cand) < max(best) - min(best):
                best = cand
    return best


def assign_owners(pos, cell, grid) -> np.ndarray:
    frac = np.asarray(pos, np.float64) @ np.linalg.inv(np.asarray(cell, np.float64))
    frac -= np.floor(frac)
    idx = [np.minimum((frac[:, k] * grid[k]).astype(np.int64), grid[k] - 1) for k in range(3)]
    return (idx[0] * grid[1] + idx[1]) * grid[2] + idx[2]


@dataclass
class BrickGraph:
    rank: int
    world: int
    n_local: int
    types: np.ndarray         # [n_local + n_ghost] species of local then ghost atoms
    global_ids: np.ndarray    # [n_local + n_ghost]
    edge_index: np.ndarray    # [2,E] local indices; row 0 (center) < n_local
    edge_vec: np.ndarray      # [E,3]
    send_lists: List[np.ndarray]   # per peer: local row ids this rank sends (peer's ghost order)
    recv_counts: List[int]         # per peer: ghost rows received (ghost rows are grouped by peer)
    # local atoms are numbered INTERIOR FIRST: atoms [0, n_interior) have no ghost among their sources, so their convolution
    # does not wait for the forward exchange (and their reverse tiles feed no ghost row): the hosts run them while the
    # exchange of the layer is in flight (engine.compute, snet_model_eval)
    n_interior: int = 0


def build_brick_graph(pos, cell, types, cutoff: float, world: int, rank: int, grid=None, pbc=(True, True, True),
                      neighbors=None) -> BrickGraph:
    """Brick `rank` of a `world`-way spatial decomposition of one periodic cell.

    Ghost identity is the global atom (features are translation invariant; the
    periodic image lives in edge_vec), so an owned atom reached through an image
    is simply a local source -- the same aliasing the serial pair style does with
    its tag map (pair_e3gnn.cpp:102,147-149)."""
    types = np.asarray(types)
    grid = grid or processor_grid(world)
    owner = assign_owners(pos, cell, grid)
    if neighbors is None:
        ei, ev, _ = neighbor_list(pos, cell, pbc, cutoff)
    else:
        ei, ev = neighbors
    ci, sj = ei[0], ei[1]
    oc, os_ = owner[ci], owner[sj]
    n = len(types)
    # (needing rank q, owning rank r, global atom j) for every cross-brick source, unique
    cross = oc != os_
    key = np.unique(oc[cross] * n + sj[cross])
    need_q, need_j = key // n, key % n
    need_r = owner[need_j]

    mine = np.nonzero(owner == rank)[0]           # sorted global ids
    n_local = len(mine)
    # interior first: owned atoms without a cross-brick source, then the boundary atoms (both in ascending global id)
    is_boundary = np.zeros(n, bool)
    is_boundary[ci[cross & (oc == rank)]] = True
    mine = np.concatenate([mine[~is_boundary[mine]], mine[is_boundary[mine]]])
    n_interior = int((~is_boundary[mine]).sum())
    if n_local == 0:
        raise RuntimeError(f'rank {rank}: empty sub-domain is not supported (as in the reference, '
                           'docs/source/user_guide/lammps_torch.md:111-113)')
    # ghosts of this rank: sorted by (owner, global id) -> contiguous per peer
    sel = need_q == rank
    gj, gr = need_j[sel], need_r[sel]
    order = np.lexsort((gj, gr))
    gj, gr = gj[order], gr[order]
    recv_counts = [int((gr == p).sum()) for p in range(world)]
    # what each peer q needs from this rank, in q's ghost order (sorted by global id)
    local_of = np.full(n, -1, np.int64)
    local_of[mine] = np.arange(n_local)
    send_lists = []
    for q in range(world):
        s = (need_q == q) & (need_r == rank)
        send_lists.append(local_of[np.sort(need_j[s])])   # (q's ghost order = ascending GLOBAL id: independent of this rank's local numbering)
    ghost_of = np.full(n, -1, np.int64)
    ghost_of[gj] = n_local + np.arange(len(gj))
    e_sel = oc == rank
    c_loc = local_of[ci[e_sel]]
    s_glob = sj[e_sel]
    s_loc = np.where(owner[s_glob] == rank, local_of[s_glob], ghost_of[s_glob])
    assert (c_loc >= 0).all() and (s_loc >= 0).all()
    gids = np.concatenate([mine, gj])
    return BrickGraph(rank, world, n_local, types[gids], gids, np.stack([c_loc, s_loc]), ev[e_sel],
                      send_lists, recv_counts, n_interior)


# --------------------------------------------------------------------------- #
# row pack / unpack: HIP kernels (device tensors only -- there is no host implementation in the
# product; the gloo tests of the exchange plan subclass HaloExchange with torch index ops)
# --------------------------------------------------------------------------- #
def _gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError('halo pack needs ROCm tensors (libsnet_hip.so kernels); no host path exists')
    out = torch.empty(idx.numel(), x.shape[1], dtype=x.dtype, device=x.device)
    if idx.numel() == 0:
        return out
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.snet_gather_rows(C.c_void_p(x.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(out.data_ptr()),
                                    idx.numel(), x.shape[1], C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               'snet_gather_rows')
    return out


def _scatter_add_rows(y: torch.Tensor, idx: torch.Tensor, x: torch.Tensor):
    """y[idx[i]] += x[i]; idx unique within one call (one peer's list)."""
    if not y.is_cuda:
        raise RuntimeError('halo unpack needs ROCm tensors (libsnet_hip.so kernels); no host path exists')
    if idx.numel() == 0:
        return
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.snet_scatter_add_rows(C.c_void_p(x.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(y.data_ptr()),
                                         idx.numel(), y.shape[1], C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               'snet_scatter_add_rows')


class HaloExchange:
    """Per-layer ghost exchange over torch.distributed (backend 'nccl' = RCCL over xGMI)."""

    def __init__(self, send_lists: Sequence[np.ndarray], recv_counts: Sequence[int], device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.dev = torch.device(device)
        self.send_counts = [int(len(s)) for s in send_lists]
        self.recv_counts = [int(c) for c in recv_counts]
        cat = np.concatenate([np.asarray(s, np.int64) for s in send_lists]) if send_lists else np.zeros(0, np.int64)
        self.send_idx = torch.as_tensor(cat, dtype=torch.int32, device=self.dev)
        self.peer_idx = [torch.as_tensor(np.asarray(s), dtype=torch.int32, device=self.dev) for s in send_lists]
        self.n_ghost = sum(self.recv_counts)
        # reverse unpack in two launches for any number of peers: received rows grouped by the local row they
        # add into (stable: contributions of one row keep their peer order, so the sum order is fixed)
        order = np.argsort(cat, kind='stable')
        rows, counts = np.unique(cat, return_counts=True)
        self.red_rows = torch.as_tensor(rows, dtype=torch.int32, device=self.dev)
        self.red_perm = torch.as_tensor(order, dtype=torch.int32, device=self.dev)
        self.red_ptr = torch.as_tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=self.dev)

    # pack / unpack primitives (overridable: the CPU tests of the exchange plan use host tensors)
    def _pack(self, x, idx):
        return _gather_rows(x, idx)

    def _unpack_add(self, y, idx, rows):
        _scatter_add_rows(y, idx, rows)

    def _reduce_add(self, y, recv):
        """y[red_rows[s]] += sum_k recv[red_perm[k]], k in segment s (one segmented sum + one scatter-add)."""
        n_seg = self.red_rows.numel()
        if n_seg == 0:
            return
        from . import _lib
        lib = _lib.load()
        tmp = torch.empty(n_seg, y.shape[1], dtype=y.dtype, device=y.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.snet_segment_sum_rows(C.c_void_p(recv.data_ptr()), C.c_void_p(self.red_ptr.data_ptr()),
                                             C.c_void_p(self.red_perm.data_ptr()), n_seg, y.shape[1],
                                             C.c_void_p(tmp.data_ptr()), st), 'snet_segment_sum_rows')
        self._unpack_add(y, self.red_rows, tmp)

    # The exchange is split into start / finish so the host can put independent kernels between them:
    # the collective runs on the backend's own stream (RCCL) while e.g. the radial MLP of the same
    # layer runs on the compute stream; finish() makes the compute stream wait for the transfer.
    def forward_start(self, x: torch.Tensor, n_local: int):
        """Begin filling ghost rows x[n_local:] with the owners' rows; returns a handle for forward_finish."""
        assert x.shape[0] == n_local + self.n_ghost and x.is_contiguous()
        send = self._pack(x, self.send_idx)
        work = self.dist.all_to_all_single(x[n_local:], send, self.recv_counts, self.send_counts, group=self.group,
                                           async_op=True)
        return work, send  # `send` stays referenced until the transfer has been waited for

    def forward_finish(self, handle):
        handle[0].wait()

    def forward(self, x: torch.Tensor, n_local: int):
        """Fill ghost rows x[n_local:] with the owners' rows."""
        self.forward_finish(self.forward_start(x, n_local))

    def reverse_start(self, gx: torch.Tensor, n_local: int):
        """Begin sending ghost-row gradients gx[n_local:] to their owners; returns a handle for reverse_finish."""
        assert gx.shape[0] == n_local + self.n_ghost and gx.is_contiguous()
        recv = torch.empty(sum(self.send_counts), gx.shape[1], dtype=gx.dtype, device=gx.device)
        out = gx[n_local:].contiguous()
        work = self.dist.all_to_all_single(recv, out, self.send_counts, self.recv_counts, group=self.group, async_op=True)
        return work, recv, out

    def reverse_finish(self, handle, gx: torch.Tensor):
        """Accumulate the received ghost-row gradients into their owners' rows
        (unpack_reverse semantics of pair_e3gnn_parallel.cpp:886-911)."""
        work, recv, _ = handle
        work.wait()
        self._reduce_add(gx, recv)

    def reverse(self, gx: torch.Tensor, n_local: int):
        self.reverse_finish(self.reverse_start(gx, n_local), gx)


class RcclComm:
    """One RCCL communicator created by libsnet_hip.so itself (snet_rccl_comm_create), so that the native
    halo below -- and any C++ host -- talks to RCCL without PyTorch in between.  The 128-byte unique id is made
    on rank 0 and handed to `bcast` (a callable bytes -> bytes that returns rank 0's value on every rank; by
    default torch.distributed.broadcast_object_list on the default process group)."""

    def __init__(self, world: int, rank: int, bcast=None):
        from . import _lib
        self.lib = _lib.load()
        ident = (C.c_char * 128)()
        if rank == 0:
            _lib.check(self.lib.snet_rccl_unique_id(C.cast(ident, C.c_void_p)), 'snet_rccl_unique_id')
        raw = bytes(ident.raw)
        if bcast is not None:
            raw = bcast(raw)
        else:
            import torch.distributed as dist
            # a world-1 communicator inside a LARGER process group keeps its own id (every process would otherwise
            # create a 1-rank communicator from rank 0's id); the broadcast runs only when the group IS this communicator's world
            if world > 1 or (dist.is_available() and dist.is_initialized() and dist.get_world_size() == world):
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
        buf = (C.c_char * 128).from_buffer_copy(raw)
        self.handle = C.c_void_p()
        _lib.check(self.lib.snet_rccl_comm_create(C.cast(buf, C.c_void_p), world, rank, C.byref(self.handle)),
                   'snet_rccl_comm_create')
        self.world, self.rank = world, rank

    @staticmethod
    def available() -> bool:
        """librccl.so can be bound in this process: a LOCAL probe (no collective), for agreeing on the transport before any rank
        enters the collective constructor above"""
        from . import _lib
        return bool(_lib.load().snet_rccl_available())

    def info(self):
        """(world, rank) as RCCL itself reports them (ncclCommCount / ncclCommUserRank)"""
        from . import _lib
        w, r = C.c_int32(), C.c_int32()
        _lib.check(self.lib.snet_rccl_comm_info(self.handle, C.byref(w), C.byref(r)), 'snet_rccl_comm_info')
        return int(w.value), int(r.value)

    def all_reduce_f64(self, t: torch.Tensor):
        """in-place sum of a float64 device tensor over all ranks (total energy, virial)"""
        from . import _lib
        assert t.dtype == torch.float64 and t.is_contiguous()
        _lib.check(self.lib.snet_rccl_allreduce_sum_f64(self.handle, C.c_void_p(t.data_ptr()), t.numel(),
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'snet_rccl_allreduce_sum_f64')
        return t

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.snet_rccl_comm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class LoopbackHub:
    """TEST transport for one GPU (csrc/snet_halo.cpp): W host threads play W ranks of the native halo exchange;
    sends and receives are device-to-device copies with RCCL's group semantics.  `comm(rank)` gives what NativeHalo
    takes in place of an RcclComm."""

    def __init__(self, world: int):
        from . import _lib
        self.lib, self.world = _lib.load(), world
        self.handle = C.c_void_p()
        _lib.check(self.lib.snet_loopback_hub_create(world, C.byref(self.handle)), 'snet_loopback_hub_create')
        self.comms = []

    def comm(self, rank: int):
        from . import _lib
        c = _LoopbackComm()
        c.lib, c.world, c.rank, c.hub = self.lib, self.world, rank, self
        c.handle = C.c_void_p()
        _lib.check(self.lib.snet_loopback_comm_create(self.handle, rank, C.byref(c.handle)), 'snet_loopback_comm_create')
        return c

    def abort(self):
        self.lib.snet_loopback_hub_abort(self.handle)


class _LoopbackComm:
    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.snet_rccl_comm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class NativeHalo:
    """The ghost exchange implemented INSIDE libsnet_hip.so (csrc/snet_halo.cpp: one ncclGroup of send / recv
    pairs per call on the compute stream, pack / reverse-accumulate kernels of the library) -- what replaces
    CommBrick::forward_comm / reverse_comm + the pack / unpack hooks of pair_e3gnn_parallel.cpp:747-911 for
    C++ hosts.  Same plan arguments and the same forward / reverse interface as HaloExchange, so both hosts
    (HipForceEngine, NativeModel) accept it; NativeModel installs it with snet_model_set_rccl_halo, i.e. the
    evaluation then runs without a single Python callback."""

    def __init__(self, comm: RcclComm, send_lists: Sequence[np.ndarray], recv_counts: Sequence[int], recv_perm=None,
                 overlap: bool = True):
        from . import _lib
        self.lib, self.comm = _lib.load(), comm
        # overlap: forward_start runs the exchange on a second HIP stream so that the ghost rows travel while the
        # caller's stream goes on with work that does not read them (HipForceEngine: self-connection, hidden radial
        # layers); forward_finish makes the caller's stream wait for it
        self.overlap, self._side = overlap, None
        world = comm.world
        assert len(send_lists) == world and len(recv_counts) == world
        sc = np.asarray([len(s) for s in send_lists], np.int32)
        rc = np.asarray(list(recv_counts), np.int32)
        idx = np.ascontiguousarray(np.concatenate([np.asarray(s, np.int64) for s in send_lists]) if world else
                                   np.zeros(0, np.int64), dtype=np.int32)
        self.n_ghost = int(rc.sum())
        # recv_perm (optional): the k-th received row (peer order) is ghost row recv_perm[k] of the host's numbering
        perm = None if recv_perm is None else np.ascontiguousarray(recv_perm, dtype=np.int32)
        self.handle = C.c_void_p()
        _lib.check(self.lib.snet_halo_create(comm.handle, world, comm.rank, C.c_void_p(sc.ctypes.data),
                                             C.c_void_p(idx.ctypes.data) if idx.size else None,
                                             C.c_void_p(rc.ctypes.data),
                                             None if perm is None else C.c_void_p(perm.ctypes.data),
                                             C.byref(self.handle)), 'snet_halo_create')

    def forward(self, x: torch.Tensor, n_local: int):
        from . import _lib
        assert x.is_contiguous() and x.dtype == torch.float32
        _lib.check(self.lib.snet_halo_forward(self.handle, C.c_void_p(x.data_ptr()), x.shape[0], n_local, x.shape[1],
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'snet_halo_forward')

    def reverse(self, gx: torch.Tensor, n_local: int):
        from . import _lib
        assert gx.is_contiguous() and gx.dtype == torch.float32
        _lib.check(self.lib.snet_halo_reverse(self.handle, C.c_void_p(gx.data_ptr()), gx.shape[0], n_local, gx.shape[1],
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'snet_halo_reverse')

    def forward_start(self, x: torch.Tensor, n_local: int):
        """Begin filling the ghost rows (same split interface as HaloExchange); returns a handle for forward_finish."""
        if not self.overlap:
            self.forward(x, n_local)
            return None
        from . import _lib
        assert x.is_contiguous() and x.dtype == torch.float32
        cur = torch.cuda.current_stream(x.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=x.device)
        self._side.wait_stream(cur)          # the rows to send are complete
        _lib.check(self.lib.snet_halo_forward(self.handle, C.c_void_p(x.data_ptr()), x.shape[0], n_local, x.shape[1],
                                              C.c_void_p(self._side.cuda_stream)), 'snet_halo_forward')
        done = torch.cuda.Event()
        done.record(self._side)
        return done, x                       # x stays referenced until the caller's stream has waited

    def forward_finish(self, handle):
        if handle is not None:
            torch.cuda.current_stream(handle[1].device).wait_event(handle[0])

    def reverse_start(self, gx: torch.Tensor, n_local: int):
        """Begin the reverse exchange on the halo's second stream: the ghost rows gx[n_local:] travel to their owners and what
        the peers return is staged inside the halo (snet_halo_reverse_exchange).  Only the GHOST rows are read, so the caller's
        stream may go on WRITING the local rows until reverse_finish (HipForceEngine: the interior tiles of the reverse
        convolution, the local rows' segment sum, the self-connection's transposed linear).  Reference: the reverse_comm inside
        the layer loop, pair_e3gnn_parallel.cpp:430-440."""
        from . import _lib
        assert gx.is_contiguous() and gx.dtype == torch.float32
        cur = torch.cuda.current_stream(gx.device)
        st = cur
        if self.overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=gx.device)
            self._side.wait_stream(cur)          # the ghost rows' gradients are complete
            st = self._side
        _lib.check(self.lib.snet_halo_reverse_exchange(self.handle, C.c_void_p(gx.data_ptr()), gx.shape[0], n_local, gx.shape[1],
                                                       C.c_void_p(st.cuda_stream)), 'snet_halo_reverse_exchange')
        done = None
        if self.overlap:
            done = torch.cuda.Event()
            done.record(self._side)
        return done, gx

    def reverse_finish(self, handle, gx: torch.Tensor = None):
        """the caller's stream waits for the exchange, then adds the staged rows into the owners' rows gx[:n_local]"""
        from . import _lib
        done, gx0 = handle
        gx = gx0 if gx is None else gx
        cur = torch.cuda.current_stream(gx.device)
        if done is not None:
            cur.wait_event(done)
        _lib.check(self.lib.snet_halo_reverse_accumulate(self.handle, C.c_void_p(gx.data_ptr()), gx.shape[1],
                                                         C.c_void_p(cur.cuda_stream)), 'snet_halo_reverse_accumulate')

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.snet_halo_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# --------------------------------------------------------------------------- #
# all bricks inside ONE process (one GPU): same exchange semantics through shared memory.
# Used to validate the N-brick == 1-brick property on a single MI355X.
# --------------------------------------------------------------------------- #
class InProcessHaloGroup:
    def __init__(self, bricks: Sequence[BrickGraph], device):
        self.world = len(bricks)
        self.dev = torch.device(device)
        self.barrier = threading.Barrier(self.world)
        self.slots: List[Optional[torch.Tensor]] = [None] * self.world
        self.members = [_InProcessHalo(self, b) for b in bricks]


class _InProcessHalo:
    def __init__(self, grp: InProcessHaloGroup, b: BrickGraph):
        self.g, self.rank = grp, b.rank
        self.send = [torch.as_tensor(np.asarray(s), dtype=torch.int32, device=grp.dev) for s in b.send_lists]
        self.recv_counts = list(b.recv_counts)

    def _sync(self):
        if self.g.dev.type == 'cuda':
            torch.cuda.synchronize(self.g.dev)
        self.g.barrier.wait()

    def forward(self, x, n_local):
        g = self.g
        g.slots[self.rank] = x
        self._sync()
        o = n_local
        for p in range(g.world):
            c = self.recv_counts[p]
            if c:
                x[o:o + c] = _gather_rows(g.slots[p], g.members[p].send[self.rank])
            o += c
        self._sync()

    def reverse(self, gx, n_local):
        self.reverse_finish(self.reverse_start(gx, n_local), gx)

    # the same split protocol as the RCCL halo (start reads the ghost rows only, finish adds into the local rows), so the
    # interior / boundary split of the hosts runs through the N-bricks-on-one-GPU tests
    def forward_start(self, x, n_local):
        self.forward(x, n_local)
        return (x,)

    def forward_finish(self, handle):
        pass

    def reverse_start(self, gx, n_local):
        g = self.g
        g.slots[self.rank] = gx[n_local:].clone()     # a snapshot of the ghost rows: the local rows may still change
        self._sync()
        return (n_local,)

    def reverse_finish(self, handle, gx):
        g = self.g
        for p in range(g.world):  # peer p holds my rows as ghosts at a fixed offset of its ghost block
            mem = g.members[p]
            c = mem.recv_counts[self.rank]
            if c:
                off = sum(mem.recv_counts[:self.rank])
                _scatter_add_rows(gx, self.send[p], g.slots[p][off:off + c].contiguous())
        self._sync()
