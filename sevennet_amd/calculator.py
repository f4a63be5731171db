"""ASE-facing host of the HIP force engine: drop-in for `sevenn.calculator.SevenNetCalculator`.

Keeps the reference's calculator surface (sevenn/calculator.py:20-233):
same constructor keywords, `implemented_properties`, result keys, units and sign
conventions
  energy / free_energy  eV
  energies              per-atom eV
  forces                eV/A  [N,3]
  stress                eV/A^3, Voigt (xx,yy,zz,yz,xz,xy) = -model_stress[[0,1,2,4,5,3]]  (:198-203)
  stresses              per-atom virial, model order xx,yy,zz,xy,yz,zx                  (:212-216)
  num_edges
and the same errors (`ValueError` on bad file_type, unknown atomic number, modal misuse).
ASE itself is optional: without it the class is a plain object whose
`compute(numbers, positions, cell, pbc)` returns the same results dict.
"""
from __future__ import annotations

import os
import pathlib
import warnings
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from .engine import HipForceEngine, build_graph
from .neighbor import neighbor_list
from .neighbor_gpu import build_graph_gpu, gpu_neighbor_supported

try:  # ASE is not a hard dependency of the engine
    from ase.calculators.calculator import Calculator, all_changes
    _HAVE_ASE = True
except ImportError:  # pragma: no cover - depends on the environment
    _HAVE_ASE = False
    all_changes = ['positions', 'numbers', 'cell', 'pbc']

    class Calculator:  # minimal stand-in so the class below is importable without ASE
        def __init__(self, **kwargs):
            self.results: Dict[str, Any] = {}
            self.atoms = None

        def calculate(self, atoms=None, properties=None, system_changes=None):
            self.atoms = atoms


_OLD_MODULE_NAMES = {
    'EdgeEmbedding': 'edge_embedding',
    'reducing nn input to hidden': 'reduce_input_to_hidden',
    'reducing nn hidden to energy': 'reduce_hidden_to_energy',
    'rescale atomic energy': 'rescale_atomic_energy',
}
for _i in range(10):
    for _old, _new in (('self connection intro', 'self_connection_intro'), ('convolution', 'convolution'),
                       ('self interaction 2', 'self_interaction_2'), ('equivariant gate', 'equivariant_gate')):
        _OLD_MODULE_NAMES[f'{_i} {_old}'] = f'{_i}_{_new}'


def map_old_state_dict(state_dict):
    """Module names of checkpoints written before the 2024-04 renaming -> current names, and the
    'denumerator' spelling -> 'denominator' (map_old_model, scripts/backward_compatibility.py:44-76)."""
    out = {}
    for k, v in state_dict.items():
        head, _, follower = k.partition('.')
        follower = follower.replace('denumerator', 'denominator')
        head = _OLD_MODULE_NAMES.get(head, head)
        out[head + ('.' + follower if follower else '')] = v
    return out


def patch_old_config(cfg: dict) -> dict:
    """Defaults that configs of version <= 0.9 lack (patch_old_config, backward_compatibility.py:18-41)."""
    version = cfg.get('version')
    if not version:
        raise ValueError('No version found in config')
    major, minor = (int(t) for t in str(version).split('.')[:2])
    if major == 0 and minor <= 9:
        cf = cfg.get('cutoff_function')
        if isinstance(cf, dict) and cf.get('cutoff_function_name') == 'XPLOR':
            cf.pop('poly_cut_p_value', None)
        if 'train_denominator' not in cfg:
            cfg['train_denominator'] = cfg.pop('train_avg_num_neigh', False)
        if cfg.pop('optimize_by_reduce', None) is False:
            raise ValueError('This checkpoint(optimize_by_reduce: False) is no longer supported')
        cfg.setdefault('conv_denominator', 0.0)
        cfg.setdefault('_normalize_sph', False)
    return cfg


W3J_KEY = '{t}_convolution.convolution._compiled_main_left_right._w3j_{l1}_{l2}_{l3}'


def fix_old_convolution_signs(cfg: dict, sd: dict, strict_reference: bool = False) -> dict:
    """Second half of the reference's `sort_old_convolution` (scripts/backward_compatibility.py:119-137).

    Checkpoints with the pre-0.11 convolution (version < 0.11.0 or 0.11.0.dev0) carry the real Wigner-3j
    tensors their e3nn build compiled into the tensor product, as buffers
    `{t}_convolution.convolution._compiled_main_left_right._w3j_{l1}_{l2}_{l3}`.  Some e3nn builds stored a
    tensor with the opposite global sign of today's `o3.wigner_3j`.  The engine always contracts with
    today's tensor (its generator is pinned to it, tests/golden/w3j_cp0.npz), so for every path with
    l1, l2, l3 > 0 whose stored buffer is the negative of the generator's, the path's columns of the
    radial MLP's last layer are negated (w * (-C) == (-w) * C) -- the weight-column *permutation* half of
    `sort_old_convolution` is not needed: the column offsets follow the checkpoint's instruction order
    (model_spec.build_model_spec, `old_convolution_order`).  A buffer that is neither +C nor -C raises.

    Deliberate difference from the reference: a buffer is shared by all paths of one (l1, l2, l3)
    (O(3) models have e.g. 1o x 1o -> 1e and 1e x 1o -> 1o on `_w3j_1_1_1`); the reference negates the
    buffer IN PLACE after fixing the first such path (`stct[conv_w3j_key] *= -1` aliases `w3j_old`),
    so later paths of the same key keep their weights while their tensor has changed sign.  Here every
    path that reads a flipped buffer is fixed, which preserves the function the checkpoint was trained as.
    All released checkpoints (SO(3)-only, one path per key) are unaffected by the difference.
    When a flipped buffer IS shared by several paths a warning says so, and `strict_reference=True`
    (or SNET_STRICT_REFERENCE_W3J=1) reproduces the upstream behaviour instead: only the first path of a key is
    negated, so that this engine's numbers equal upstream's for such a checkpoint.
    Returns a state dict without the `_w3j_*` buffers."""
    from .irreps import real_wigner_3j
    from .model_spec import build_model_spec, old_convolution_order
    out = {k: v for k, v in sd.items() if '._w3j_' not in k}
    if not old_convolution_order(cfg.get('version', '0.0.0')) or len(out) == len(sd):
        return out   # current instruction order, or a checkpoint stripped of its buffers: nothing to compare
    spec = build_model_spec(cfg)   # paths in the checkpoint's own (unsorted) instruction order
    strict_reference = strict_reference or os.environ.get('SNET_STRICT_REFERENCE_W3J') == '1'
    for ls in spec.layers:
        flipped = {}
        ww_key = f'{ls.t}_convolution.weight_nn.layer{len(ls.mlp_dims) - 2}.weight'
        ww = None
        for p in ls.conv_full.paths:   # the checkpoint's column layout (the engine may have pruned unread paths)
            if not (p.l1 > 0 and p.l2 > 0 and p.l3 > 0):
                continue
            key = W3J_KEY.format(t=ls.t, l1=p.l1, l2=p.l2, l3=p.l3)
            if key not in sd:
                continue   # checkpoint stripped of buffers: nothing to compare with
            old = np.asarray(sd[key], dtype=np.float64)
            now = real_wigner_3j(p.l1, p.l2, p.l3)
            if old.shape != now.shape:
                raise ValueError(f'{key}: shape {old.shape} != {now.shape}')
            if np.allclose(old, now, rtol=1e-5, atol=1e-6):
                continue
            if not np.allclose(old, -now, rtol=1e-5, atol=1e-6):
                raise ValueError(f'{key} is neither +wigner_3j nor -wigner_3j: the checkpoint was written with an '
                                 'incompatible Clebsch-Gordan convention (e3nn < 0.4?)')
            flipped[key] = flipped.get(key, 0) + 1
            if strict_reference and flipped[key] > 1:
                continue   # upstream flipped the shared buffer in place after the first path: later paths keep their weights
            if ww is None:
                ww = np.array(out[ww_key], copy=True)
            ww[:, p.w_off:p.w_off + p.mul] *= -1.0
        if ww is not None:
            out[ww_key] = ww
        shared = sorted(k for k, c in flipped.items() if c > 1)
        if shared:
            warnings.warn(f'{shared}: a sign-flipped Wigner-3j buffer is shared by several tensor-product paths; '
                          + ('only the first path of each was negated, as sevenn/scripts/backward_compatibility.py does '
                             '(strict_reference)' if strict_reference else
                             'every such path was negated (the function the checkpoint was trained as); upstream negates only '
                             'the first -- pass strict_reference=True / SNET_STRICT_REFERENCE_W3J=1 to reproduce its numbers'))
    return out


def load_reference_checkpoint(path: str, strict_reference: bool = False):
    """(config, state_dict) from a reference checkpoint file (sevenn/checkpoint.py:286-308:
    a torch pickle holding 'config' and 'model_state_dict'), with the reference's own
    backward-compatibility steps (patch_state_dict_if_old, scripts/backward_compatibility.py:165-184):
    old config defaults, old module names, and `sort_old_convolution` -- its weight-column permutation is
    absorbed by deriving the column offsets from the checkpoint's instruction order
    (model_spec.old_convolution_order), its Wigner-3j sign fix is `fix_old_convolution_signs`."""
    cp = torch.load(path, map_location='cpu', weights_only=False)
    if not isinstance(cp, dict) or 'config' not in cp or 'model_state_dict' not in cp:
        raise ValueError(f'{path} is not a SevenNet checkpoint (config + model_state_dict expected)')
    cfg = patch_old_config(dict(cp['config']))
    sd = {k: v.detach().cpu().numpy() for k, v in map_old_state_dict(cp['model_state_dict']).items()
          if hasattr(v, 'detach')}
    return cfg, fix_old_convolution_signs(cfg, sd, strict_reference)


class SevenNetCalculator(Calculator):
    """Supporting properties: 'free_energy', 'energy', 'forces', 'stress', 'stresses', 'energies'."""

    implemented_properties = ['free_energy', 'energy', 'forces', 'stress', 'stresses', 'energies']

    def __init__(
        self,
        model: Union[str, pathlib.PurePath, tuple] = '7net-0',
        file_type: str = 'checkpoint',
        device: Union[torch.device, str] = 'auto',
        modal: Optional[str] = None,
        enable_cueq: Optional[bool] = False,   # accepted for signature compatibility, ignored:
        enable_flash: Optional[bool] = False,  # the tensor product always runs in libsnet_hip.so
        enable_oeq: Optional[bool] = False,
        compute_atomic_virial: bool = False,
        sevennet_config: Optional[Dict] = None,
        **kwargs,
    ) -> None:
        super().__init__(**kwargs)
        self.compute_atomic_virial = compute_atomic_virial
        if isinstance(model, pathlib.PurePath):
            model = str(model)
        allowed_file_types = ['checkpoint', 'model_instance']
        file_type = file_type.lower()
        if file_type not in allowed_file_types:
            if file_type == 'torchscript':
                raise ValueError('torchscript file_type is no longer supported. '
                                 'Use checkpoint or model_instance instead.')
            raise ValueError(f'file_type not in {allowed_file_types}')
        if isinstance(device, str):
            device = 'cuda:0' if device in ('auto', 'cuda') else device
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('the HIP force engine runs on a ROCm GPU only (device must be cuda:N)')

        if file_type == 'checkpoint' and isinstance(model, str):
            if not os.path.isfile(model):
                raise ValueError(f'checkpoint file {model!r} not found (pretrained model names need the '
                                 'reference package to download/resolve them)')
            cfg, sd = load_reference_checkpoint(model)
        elif file_type == 'model_instance' and isinstance(model, tuple) and len(model) == 2:
            cfg, sd = model  # (config dict, state_dict of arrays)
        else:
            raise ValueError('Unexpected input combinations')
        self.sevennet_config = cfg if cfg is not None else sevennet_config
        tm = cfg.get('_type_map')
        if not tm:
            raise ValueError('Model must have the type_map to be used with calculator')
        self.type_map = {int(k): int(v) for k, v in tm.items()}
        self.cutoff = float(cfg['cutoff'])
        self.modal = None
        modal_map = cfg.get('_modal_map') if cfg.get('use_modality') else None
        if modal_map:  # sevenn/calculator.py:159-170
            modal_ava = list(modal_map.keys())
            if not modal:
                raise ValueError(f'modal argument missing (avail: {modal_ava})')
            if modal not in modal_ava:
                raise ValueError(f'unknown modal {modal} (not in {modal_ava})')
            self.modal = modal
        elif modal:
            warnings.warn(f'modal={modal} is ignored as model has no modal_map')
        self.model = HipForceEngine(cfg, sd, device=str(self.device), modal=self.modal)
        self._z2type = np.full(120, -1, np.int64)  # sequential.py:80-83
        for z, t in self.type_map.items():
            self._z2type[z] = t

    def set_atoms(self, atoms) -> None:
        for z in set(atoms.get_atomic_numbers()):
            if z not in self.type_map:
                raise ValueError(f'Model do not know atomic number: {z}, (knows: {list(self.type_map.keys())})')

    def compute(self, numbers, positions, cell, pbc) -> Dict[str, Any]:
        numbers = np.asarray(numbers, np.int64)
        types = self._z2type[numbers]
        if (types < 0).any():
            bad = sorted(set(numbers[types < 0].tolist()))
            raise ValueError(f'Model do not know atomic number: {bad[0]}, (knows: {list(self.type_map.keys())})')
        cell = np.asarray(cell, np.float64).reshape(3, 3)
        # per-species row lists only where a per-species (FCTP) self-connection reads them
        ns = self.model.spec.num_species if self.model.needs_species_rows else 0
        if len(numbers) and gpu_neighbor_supported(cell, pbc, self.cutoff, np.asarray(positions, np.float64)):
            # cell list on the GPU: bulk cells, slabs / wires / molecules (open axes), cells thinner than the cutoff
            g = build_graph_gpu(types, positions, cell, self.cutoff, device=str(self.device), num_species=ns, pbc=pbc)
            n_edges = g.n_edges
        else:  # singular cells only: host list
            ei, ev, _ = neighbor_list(positions, cell, pbc, self.cutoff)
            g = build_graph(types, ei, ev, device=str(self.device), num_species=ns)
            n_edges = int(ei.shape[1])
        out = self.model.compute(g, want_atomic_virial=self.compute_atomic_virial)
        energy = float(out['energy'].cpu())
        vol = abs(float(np.linalg.det(cell)))
        vir = out['virial'].cpu().numpy()  # = -sum(r (x) g): model stress * volume, order xx,yy,zz,xy,yz,zx
        stress = -(vir / vol)[[0, 1, 2, 4, 5, 3]] if vol > 0 else np.full(6, np.nan)
        res: Dict[str, Any] = {
            'free_energy': energy, 'energy': energy,
            'energies': out['atomic_energy'].cpu().numpy().astype(np.float64),
            'forces': out['forces'].cpu().numpy().astype(np.float64),
            'stress': stress, 'num_edges': n_edges,
        }
        if self.compute_atomic_virial:
            res['stresses'] = out['atomic_virial'].cpu().numpy()
        return res

    def calculate(self, atoms=None, properties=None, system_changes=all_changes):
        Calculator.calculate(self, atoms, properties, system_changes)
        if atoms is None:
            raise ValueError('No atoms to evaluate')
        self.results = self.compute(atoms.get_atomic_numbers(), atoms.get_positions(),
                                    np.array(atoms.get_cell()), atoms.get_pbc())
