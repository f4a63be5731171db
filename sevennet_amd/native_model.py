"""Python handle on the native whole-model sequencer (`snet_model_*`, csrc/snet_model.cpp).

This is the path a C++ host (the LAMMPS pair styles) drives; the class exists so the sequencer is
tested against the same oracle and goldens as `HipForceEngine`, and as a lower-overhead host for
small systems (one C call per evaluation instead of ~150 ctypes calls).
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .engine import Graph
from .model_file import write_model_file


class _DevRows:
    """Zero-copy torch view of a device pointer handed to a halo callback."""

    def __init__(self, ptr: int, rows: int, dim: int):
        self.__cuda_array_interface__ = {'shape': (rows, dim), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


class NativeModel:
    def __init__(self, model, state_dict: Optional[Dict[str, np.ndarray]] = None, device='cuda:0', modal=None):
        """model: path of a `.snet` file, or a reference config dict (then `state_dict` is required and
        the file is written to a temporary location first)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('NativeModel needs a ROCm GPU (no CPU fallback exists)')
        self.dev = torch.device(device)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.dev):
            if isinstance(model, dict):
                if state_dict is None:
                    raise ValueError('state_dict is required with a config dict')
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, 'model.snet')
                    write_model_file(path, model, state_dict, modal=modal)
                    _lib.check(self.lib.snet_model_load(path.encode(), C.byref(self.handle)), 'snet_model_load')
            else:
                _lib.check(self.lib.snet_model_load(os.fspath(model).encode(), C.byref(self.handle)), 'snet_model_load')
        cut, ns, nl = C.c_float(), C.c_int32(), C.c_int32()
        comm = (C.c_int32 * 64)()
        _lib.check(self.lib.snet_model_info(self.handle, C.byref(cut), C.byref(ns), C.byref(nl), comm, 64),
                   'snet_model_info')
        self.cutoff, self.num_species, self.n_layers = float(cut.value), int(ns.value), int(nl.value)
        self.comm_dims = [int(comm[i]) for i in range(self.n_layers)]
        self._cb = None
        # topology-only work (tile list, source grouping, species row lists) is kept per Graph object: a Graph's index arrays
        # are immutable device tensors, so "same object" = "same topology"
        _lib.check(self.lib.snet_model_set_topology_cache(self.handle, 1), 'snet_model_set_topology_cache')
        self._last_graph = None

    def eval_syncs(self) -> int:
        """stream synchronisations snet_model_eval has issued so far (0 per step once the topology is cached)"""
        return int(self.lib.snet_model_eval_syncs(self.handle))

    def meta(self, key: str) -> str:
        buf = C.create_string_buffer(4096)
        _lib.check(self.lib.snet_model_meta(self.handle, key.encode(), buf, 4096), 'snet_model_meta')
        return buf.value.decode()

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.snet_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_halo(self, halo, fold_forces: bool = True) -> None:
        """halo: object with forward(x, n_local) / reverse(gx, n_local) on device tensors
        (sevennet_amd.parallel.HaloExchange), or None."""
        if halo is None:
            self._cb = None
            _lib.check(self.lib.snet_model_set_halo(self.handle, None, None, None, 1), 'snet_model_set_halo')
            return
        from .parallel import NativeHalo
        if isinstance(halo, NativeHalo):  # the library's own RCCL exchange: no Python callback in the evaluation
            self._cb = halo
            _lib.check(self.lib.snet_model_set_rccl_halo(self.handle, halo.handle, int(fold_forces)), 'snet_model_set_rccl_halo')
            return

        def wrap(fn):
            def cb(_user, ptr, n_total, n_local, dim, _stream):
                try:
                    x = torch.as_tensor(_DevRows(ptr, n_total, dim), device=self.dev)
                    fn(x, n_local)
                    return 0
                except Exception as exc:  # never unwind through C
                    self._cb_error = exc
                    return 1
            return _lib.HALO_FN(cb)

        self._cb = (wrap(halo.forward), wrap(halo.reverse))
        _lib.check(self.lib.snet_model_set_halo(self.handle, C.cast(self._cb[0], C.c_void_p),
                                                C.cast(self._cb[1], C.c_void_p), None, int(fold_forces)),
                   'snet_model_set_halo')

    def compute(self, g: Graph, want_atomic_virial: bool = False):
        """Same results dict as HipForceEngine.compute."""
        with torch.cuda.device(self.dev):
            N, NT, E = g.n_local, g.n_total, g.n_edges
            f32 = dict(dtype=torch.float32, device=self.dev)
            energy = torch.empty(1, dtype=torch.float64, device=self.dev)
            virial = torch.empty(6, dtype=torch.float64, device=self.dev)
            e_atom, g_vec, forces = torch.empty(N, **f32), torch.empty(E, 3, **f32), torch.empty(NT, 3, **f32)
            vir_atom = torch.empty(NT, 6, **f32) if want_atomic_virial else None
            types_host = getattr(g, '_types_host', None)  # host copy made once per graph (a D2H copy syncs)
            if types_host is None:
                types_host = np.ascontiguousarray(g.types[:N].cpu().numpy(), dtype=np.int32)
                g._types_host = types_host
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
            self._cb_error = None
            if self._last_graph is not g:   # (holding the reference also keeps the old arrays' addresses from being reused)
                _lib.check(self.lib.snet_model_topology_changed(self.handle), 'snet_model_topology_changed')
                # bricks number their interior rows first: the sequencer overlaps them with the library halo's exchange
                _lib.check(self.lib.snet_model_set_interior(self.handle, int(getattr(g, 'n_interior', 0) or 0)), 'snet_model_set_interior')
                self._last_graph = g
            rc = self.lib.snet_model_eval(self.handle, NT, N, E, p(g.types), C.c_void_p(types_host.ctypes.data),
                                          p(g.row_ptr), p(g.src), p(g.col_ptr), p(g.eperm), p(g.edge_vec), p(g.w_row), p(g.pair_edge),
                                          g.n_pairs if g.w_row is not None else 0, p(energy),
                                          p(e_atom), p(g_vec), p(forces), p(virial), p(vir_atom),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc and self._cb_error is not None:
                raise self._cb_error
            _lib.check(rc, 'snet_model_eval')
            if g.order is not None:
                tmp = torch.empty_like(g_vec)
                tmp[g.order] = g_vec
                g_vec = tmp
            out = dict(energy=energy, atomic_energy=e_atom, dE_dr=g_vec, forces=forces[:N], virial=virial)
            if vir_atom is not None:
                out['atomic_virial'] = vir_atom[:N]
            return out
