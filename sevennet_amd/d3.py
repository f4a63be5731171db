"""DFT-D3 dispersion for the ASE-level surface.

Mirrors sevenn.calculator.D3Calculator / SevenNetD3Calculator (sevenn/calculator.py:236-314, 387-618): same
constructor arguments (damping_type 'damp_bj' | 'damp_zero', functional_name, vdw_cutoff / cn_cutoff in bohr^2), same
result keys, units and signs -- over libsnet_hip.so's own HIP kernels (csrc/snet_d3.hip, C-ABI snet_d3_*).  The
published D3 tables travel as a data blob (sevennet_amd/data/d3_params.npz, written by oracle/tools/make_d3_params.py).
There is no CPU path: the reference needs CUDA for this term, this one needs a ROCm GPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict

import numpy as np

from . import _lib

AU_TO_ANG = 0.52917726
_BLOB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'd3_params.npz')
_DAMPING = {'damp_zero': 0, 'damp_bj': 1}


def _dp(a):
    return C.c_void_p(a.ctypes.data)


class D3Engine:
    """thin handle over snet_d3_*: compute(numbers, positions, cell, pbc) -> energy, forces, stress (3x3, dE/d strain / V)"""

    def __init__(self, damping_type: str = 'damp_bj', functional_name: str = 'pbe', vdw_cutoff: float = 9000.0,
                 cn_cutoff: float = 1600.0, blob: str = _BLOB):
        self.damp_name, self.func_name = damping_type.lower(), functional_name.lower()
        if self.damp_name not in _DAMPING:
            raise ValueError('Error: Invalid damping type.')      # sevenn/calculator.py:424-425
        import torch
        if not torch.cuda.is_available():
            raise NotImplementedError('CPU + D3 is not implemented yet')     # :415-416
        z = np.load(blob)
        names = z[self.damp_name + '_names'].tolist()
        if self.func_name not in names:
            raise ValueError(f'Functional name unknown: {functional_name!r} for {self.damp_name} (known: {names})')
        func = np.ascontiguousarray(z[self.damp_name + '_params'][names.index(self.func_name)], np.float64)
        self.rthr, self.cnthr = float(vdw_cutoff), float(cn_cutoff)
        self.lib = _lib.load()
        self.handle = C.c_void_p()
        _lib.check(self.lib.snet_d3_create(C.byref(self.handle)), 'snet_d3_create')
        tabs = [np.ascontiguousarray(z[k], np.float64) for k in ('r0ab', 'c6ab', 'r2r4', 'rcov')]
        _lib.check(self.lib.snet_d3_set_tables(self.handle, _dp(tabs[0]), _dp(tabs[1]), tabs[1].shape[0], _dp(tabs[2]), _dp(tabs[3])),
                   'snet_d3_set_tables')
        _lib.check(self.lib.snet_d3_settings(self.handle, self.rthr, self.cnthr, _DAMPING[self.damp_name], _dp(func)), 'snet_d3_settings')

    def compute(self, numbers, positions, cell, pbc) -> Dict[str, Any]:
        import torch
        numbers = np.ascontiguousarray(numbers, np.int32)
        positions = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
        cell = np.array(cell, np.float64).reshape(3, 3)
        pbc = np.asarray(pbc, bool).reshape(3)
        if cell.sum() == 0:
            # sevenn/calculator.py:533-548: molecules get an orthogonal periodic box larger than the longest cutoff
            max_cutoff = np.sqrt(max(self.rthr, self.cnthr)) * AU_TO_ANG
            cell = np.eye(3) * (positions.max(0) - positions.min(0) + max_cutoff + 1.0)
            pbc = np.array([True, True, True])
        n = len(numbers)
        _lib.check(self.lib.snet_d3_set_atoms(self.handle, n, _dp(numbers), _dp(positions)), 'snet_d3_set_atoms')
        cell_c = np.ascontiguousarray(cell)
        pbc_c = np.ascontiguousarray(pbc.astype(np.int32))
        _lib.check(self.lib.snet_d3_set_cell(self.handle, _dp(cell_c), _dp(pbc_c)), 'snet_d3_set_cell')
        _lib.check(self.lib.snet_d3_compute(self.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'snet_d3_compute')
        f = np.ctypeslib.as_array(self.lib.snet_d3_forces(self.handle), shape=(n, 3)).copy()
        s = np.ctypeslib.as_array(self.lib.snet_d3_stress(self.handle), shape=(3, 3)).copy()
        cn = np.ctypeslib.as_array(self.lib.snet_d3_coordination_numbers(self.handle), shape=(n,)).copy()
        return dict(energy=float(self.lib.snet_d3_energy(self.handle)), forces=f, stress=s, cn=cn)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.snet_d3_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


try:  # ASE is optional (absent in the offline image): the classes below then expose compute() only
    from ase.calculators.calculator import Calculator, all_changes
    _HAVE_ASE = True
except Exception:  # noqa: BLE001
    _HAVE_ASE = False
    all_changes = None

    class Calculator:  # minimal stand-in with the attributes the calculators use
        def __init__(self, **kwargs):
            self.results = {}

        def calculate(self, atoms=None, properties=None, system_changes=None):
            self.atoms = atoms


class D3Calculator(Calculator):
    """ASE calculator for the D3 van der Waals correction (sevenn/calculator.py:387-618).
    implemented_properties and result conventions as the reference: free_energy = energy (eV), forces (eV/A),
    stress in ASE Voigt order xx, yy, zz, yz, xz, xy (eV/A^3)."""

    implemented_properties = ['free_energy', 'energy', 'forces', 'stress']

    def __init__(self, damping_type: str = 'damp_bj', functional_name: str = 'pbe', vdw_cutoff: float = 9000,
                 cn_cutoff: float = 1600, **kwargs) -> None:
        super().__init__(**kwargs)
        self.engine = D3Engine(damping_type, functional_name, vdw_cutoff, cn_cutoff)
        self.rthr, self.cnthr = self.engine.rthr, self.engine.cnthr
        self.damp_name, self.func_name = self.engine.damp_name, self.engine.func_name

    def compute(self, numbers, positions, cell, pbc) -> Dict[str, Any]:
        r = self.engine.compute(numbers, positions, cell, pbc)
        s = r['stress']
        return {'free_energy': r['energy'], 'energy': r['energy'], 'forces': r['forces'],
                'stress': np.array([s[0, 0], s[1, 1], s[2, 2], s[1, 2], s[0, 2], s[0, 1]])}

    def calculate(self, atoms=None, properties=None, system_changes=all_changes):
        Calculator.calculate(self, atoms, properties, system_changes)
        if atoms is None:
            raise ValueError('No atoms to evaluate')
        if atoms.get_cell().sum() == 0:
            print('Warning: D3Calculator requires a cell.\nWarning: An orthogonal cell large enough is generated.')
        self.results = self.compute(atoms.get_atomic_numbers(), atoms.get_positions(), np.array(atoms.get_cell()), atoms.get_pbc())


def _d3_pair(model, file_type, device, modal, enable_cueq, enable_flash, enable_oeq, sevennet_config, damping_type,
             functional_name, vdw_cutoff, cn_cutoff, kwargs):
    import warnings
    from .calculator import SevenNetCalculator
    if kwargs.get('compute_atomic_virial', False):
        warnings.warn('D3Calculator does not support per-atom stress. Atomic stress from SevenNetD3Calculator will not '
                      'include D3 contributions.')
    d3_kwargs = {k: v for k, v in kwargs.items() if k != 'compute_atomic_virial'}
    d3_calc = D3Calculator(damping_type=damping_type, functional_name=functional_name, vdw_cutoff=vdw_cutoff,
                           cn_cutoff=cn_cutoff, **d3_kwargs)
    sevennet_calc = SevenNetCalculator(model=model, file_type=file_type, device=device, modal=modal, enable_cueq=enable_cueq,
                                       enable_flash=enable_flash, enable_oeq=enable_oeq, sevennet_config=sevennet_config, **kwargs)
    return sevennet_calc, d3_calc


if _HAVE_ASE:
    from ase.calculators.mixing import SumCalculator as _SumBase
else:
    _SumBase = object


class SevenNetD3Calculator(_SumBase):
    """SevenNet + D3 (sevenn/calculator.py:236-314: a SumCalculator subclass holding the two).  With ASE present this is an
    ase.calculators.mixing.SumCalculator subclass like the reference's; without it, `compute(numbers, positions, cell, pbc)`
    returns the summed results."""

    def __init__(self, model='7net-0', file_type: str = 'checkpoint', device='auto', modal=None, enable_cueq=False,
                 enable_flash=False, enable_oeq=False, sevennet_config=None, damping_type: str = 'damp_bj',
                 functional_name: str = 'pbe', vdw_cutoff: float = 9000, cn_cutoff: float = 1600, **kwargs):
        pair = _d3_pair(model, file_type, device, modal, enable_cueq, enable_flash, enable_oeq, sevennet_config, damping_type,
                        functional_name, vdw_cutoff, cn_cutoff, kwargs)
        if _HAVE_ASE:
            super().__init__(list(pair))
        else:
            self.calcs = list(pair)

    def compute(self, numbers, positions, cell, pbc) -> Dict[str, Any]:
        a, b = (c.compute(numbers, positions, cell, pbc) for c in self.calcs)
        out = dict(a)
        for k in ('free_energy', 'energy', 'forces', 'stress'):
            out[k] = a[k] + b[k]
        return out
