#!/usr/bin/env python3
"""Golden vectors from the reference's own PURE-TORCH modules (run in the BUILD container only).

`import sevenn` needs e3nn / ase / torch_geometric, none of which exist offline, but several modules
on the hot path are plain torch code whose only e3nn dependency is the `@compile_mode('script')`
decorator of their file.  This script executes those class definitions *from the files where they
lie under /root/reference* (ast-extracted, nothing is copied into the repository), with
  KEY                = sevenn/_keys.py loaded by path (plain string constants),
  broadcast          = sevenn/nn/util.py loaded by path (plain torch),
  compile_mode(...)  = identity decorator,
and records inputs + outputs of their `forward` as data:

  tests/golden/ref_torch_modules.npz
    BesselBasis.forward            sevenn/nn/edge_embedding.py:81-103
    PolynomialCutoff.forward       sevenn/nn/edge_embedding.py:106-132
    XPLORCutoff.forward            sevenn/nn/edge_embedding.py:135-160
    ForceStressOutputFromEdge      sevenn/nn/force_output.py:140-230   (forces, stress, atomic virial)
    Rescale / SpeciesWiseRescale / ModalWiseRescale.forward   sevenn/nn/scale.py:22-56,59-162,165-363
    AtomReduce.forward             sevenn/nn/linear.py:105-141
    OnehotEmbedding.forward        sevenn/nn/node_embedding.py:14-53

These pin the oracle branches the TorchScript fixtures do not reach: the XPLOR cutoff (SevenNet-0,
MF-ompa), the edge-based force / virial / atomic-virial reduction and its sign / ordering, species-wise
and modal-wise rescale.  Usage:  python oracle/tools/make_golden_torch_modules.py
"""
import ast
import importlib.util
import math
import os
import re
import sys
from typing import Any, Dict, List, Optional, Union  # noqa: F401 (names used by the executed class bodies)

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/sevenn'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_classes(rel_path: str, names: List[str], extra: Dict[str, Any]) -> Dict[str, Any]:
    """exec the ClassDef / FunctionDef nodes `names` of a reference source file, in place"""
    path = os.path.join(REF, rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert {n.name for n in keep} == set(names), (rel_path, names)
    ns = dict(extra)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, 'exec'), ns)
    return ns


def main():
    KEY = _load_by_path('ref_keys', os.path.join(REF, '_keys.py'))
    util = _load_by_path('ref_nn_util', os.path.join(REF, 'nn', 'util.py'))
    n_univ = int(re.search(r'^NUM_UNIV_ELEMENT\s*=\s*(\d+)', open(os.path.join(REF, '_const.py')).read(), re.M).group(1))
    base = dict(torch=torch, nn=nn, math=math, KEY=KEY, broadcast=util.broadcast, AtomGraphDataType=Dict[str, torch.Tensor],
                NUM_UNIV_ELEMENT=n_univ, compile_mode=lambda *_a, **_k: (lambda cls: cls),
                Any=Any, Dict=Dict, List=List, Optional=Optional, Union=Union)
    out: Dict[str, np.ndarray] = {}
    g = torch.Generator().manual_seed(20260926)

    # ---- radial basis and cutoffs (fp32 like the model, and fp64 for the oracle's own precision)
    ee = reference_classes('nn/edge_embedding.py', ['BesselBasis', 'PolynomialCutoff', 'XPLORCutoff'], base)
    for tag, rc, r_on, p in (('rc5', 5.0, 4.5, 6), ('rc6', 6.0, 5.5, 6), ('rc4', 4.0, 3.0, 5)):
        r = torch.cat([torch.rand(200, generator=g, dtype=torch.float64) * rc,
                       torch.tensor([1e-3, 0.5, r_on - 1e-6, r_on, r_on + 1e-6, rc - 1e-6, rc], dtype=torch.float64)])
        out[f'{tag}_r'] = r.numpy()
        out[f'{tag}_params'] = np.array([rc, r_on, p], np.float64)
        for dt, sfx in ((torch.float64, 'f64'), (torch.float32, 'f32')):
            rr = r.to(dt)
            bb = ee['BesselBasis'](rc, 8, trainable_coeff=True)
            coeffs = bb.coeffs.detach().to(dt)
            bb.coeffs = nn.Parameter(coeffs)
            out[f'{tag}_bessel_{sfx}'] = bb(rr).detach().numpy()
            out[f'{tag}_coeffs_{sfx}'] = coeffs.numpy()
            out[f'{tag}_poly_{sfx}'] = ee['PolynomialCutoff'](rc, p)(rr).detach().numpy()
            out[f'{tag}_xplor_{sfx}'] = ee['XPLORCutoff'](rc, r_on)(rr).detach().numpy()

    # ---- forces / stress / atomic virial from edge gradients
    fo = reference_classes('nn/force_output.py', ['ForceStressOutputFromEdge'], base)
    n_atoms, n_edges = 23, 160
    idx = torch.randint(0, n_atoms, (2, n_edges), generator=g)
    for dt, sfx in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        rij = (torch.randn(n_edges, 3, generator=torch.Generator().manual_seed(7), dtype=torch.float64) * 1.7).to(dt)
        rij.requires_grad_(True)
        Q = torch.randn(3, 3, generator=torch.Generator().manual_seed(8), dtype=torch.float64).to(dt)
        wts = torch.randn(n_edges, generator=torch.Generator().manual_seed(9), dtype=torch.float64).to(dt)
        # any smooth scalar of the edge vectors will do: the module only sees dE/d(rij)
        energy = (wts * torch.sin((rij @ Q * rij).sum(-1))).sum() + (rij.norm(dim=-1) ** 3).sum() * 0.01
        mod = fo['ForceStressOutputFromEdge'](use_atomic_virial=True)
        mod._is_batch_data = False
        mod.eval()
        vol = torch.tensor([97.5], dtype=dt)
        data = {KEY.NUM_ATOMS: torch.tensor([n_atoms]), KEY.EDGE_VEC: rij, KEY.EDGE_IDX: idx,
                KEY.PRED_TOTAL_ENERGY: energy, KEY.CELL_VOLUME: vol}
        (gij,) = torch.autograd.grad(energy, [rij], retain_graph=True)
        data = mod(data)
        out[f'fs_rij_{sfx}'] = rij.detach().numpy()
        out[f'fs_gij_{sfx}'] = gij.numpy()
        out[f'fs_force_{sfx}'] = data[KEY.PRED_FORCE].detach().numpy()
        out[f'fs_stress_{sfx}'] = data[KEY.PRED_STRESS].detach().numpy().reshape(-1)
        out[f'fs_atomic_virial_{sfx}'] = data[KEY.PRED_ATOMIC_VIRIAL].detach().numpy()
    out['fs_edge_index'] = idx.numpy()
    out['fs_volume'] = np.array([97.5])

    # ---- rescale variants
    sc = reference_classes('nn/scale.py', ['_as_univ', 'Rescale', 'SpeciesWiseRescale', 'ModalWiseRescale'], base)
    n, ns, nm = 41, 5, 3
    e_in = torch.randn(n, 1, generator=g)
    types = torch.randint(0, ns, (n,), generator=g)
    shift_s = [float(v) for v in torch.randn(ns, generator=g)]
    scale_s = [float(v) for v in torch.rand(ns, generator=g) + 0.5]
    shift_m = [[float(v) for v in torch.randn(ns, generator=g)] for _ in range(nm)]
    scale_m = [[float(v) for v in torch.rand(ns, generator=g) + 0.5] for _ in range(nm)]
    out['rs_in'], out['rs_types'] = e_in.numpy(), types.numpy()
    d = sc['Rescale'](-1.25, 0.75)({KEY.SCALED_ATOMIC_ENERGY: e_in})
    out['rs_global'] = d[KEY.ATOMIC_ENERGY].detach().numpy()
    out['rs_global_params'] = np.array([-1.25, 0.75])
    d = sc['SpeciesWiseRescale'](shift_s, scale_s)({KEY.SCALED_ATOMIC_ENERGY: e_in, KEY.ATOM_TYPE: types})
    out['rs_species'] = d[KEY.ATOMIC_ENERGY].detach().numpy()
    out['rs_species_shift'], out['rs_species_scale'] = np.array(shift_s, np.float32), np.array(scale_s, np.float32)
    for use_shift, use_scale, tag in ((True, True, 'mm'), (True, False, 'ms'), (False, True, 'sm')):
        mw = sc['ModalWiseRescale'](shift_m if use_shift else shift_s, scale_m if use_scale else scale_s,
                                    use_modal_wise_shift=use_shift, use_modal_wise_scale=use_scale)
        mw._is_batch_data = False
        for modal in range(nm):
            d = mw({KEY.SCALED_ATOMIC_ENERGY: e_in, KEY.ATOM_TYPE: types,
                    KEY.MODAL_TYPE: torch.full((n,), modal, dtype=torch.int64)})
            out[f'rs_modal_{tag}_{modal}'] = d[KEY.ATOMIC_ENERGY].detach().numpy()
    out['rs_modal_shift'], out['rs_modal_scale'] = np.array(shift_m, np.float32), np.array(scale_m, np.float32)

    # ---- energy sum and one-hot embedding
    li = reference_classes('nn/linear.py', ['AtomReduce'], base)
    ar = li['AtomReduce'](KEY.ATOMIC_ENERGY, KEY.PRED_TOTAL_ENERGY)
    ar._is_batch_data = False
    d = ar({KEY.ATOMIC_ENERGY: torch.tensor(out['rs_species'])})
    out['reduce_total'] = d[KEY.PRED_TOTAL_ENERGY].detach().numpy().reshape(-1)
    ne = reference_classes('nn/node_embedding.py', ['OnehotEmbedding'], base)
    oh = ne['OnehotEmbedding'](num_classes=ns)
    d = oh({KEY.NODE_FEATURE: types})
    out['onehot'] = d[KEY.NODE_FEATURE].detach().numpy()

    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'ref_torch_modules.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes')


if __name__ == '__main__':
    main()
