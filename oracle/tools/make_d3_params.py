#!/usr/bin/env python3
"""DFT-D3 parameter blob: the published constants of Grimme's dftd3 (J. Chem. Phys. 132, 154104 (2010); BJ damping:
J. Comput. Chem. 32, 1456 (2011)) as the reference carries them, extracted as DATA into one .npz.

    python oracle/tools/make_d3_params.py [/root/reference] [sevennet_amd/data/d3_params.npz]

Sources read in place (nothing is copied as source text; only numbers and functional names leave):
  sevenn/pair_e3gnn/pair_d3_pars.h        R0AB_TABLE [94,94] (cut-off radii, Angstrom), C6AB_TABLE [32385,5]
                                          (C6 reference value, Z_i + 100 (ref_i - 1), Z_j + 100 (ref_j - 1), CN_i, CN_j)
  sevenn/pair_e3gnn/pair_d3_for_ase.cu    r2r4_ref / rcov_ref [94] (:660-722), functional parameters of the four damping
                                          variants (setfuncpar_zero / _bj / _zerom / _bjm, :394-606)
The blob holds: r0ab, c6ab (the table, float64), r2r4, rcov, and per damping variant the functional names with
(s6, rs6, s18, rs18, alp) as the reference's switch statements assign them (defaults s6 = 1, alp = 14, rs18 = 1 for zero
damping; a statement placed after `break` is dead code there too and is ignored here)."""
import os
import re
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                                         'sevennet_amd', 'data', 'd3_params.npz')
pars = open(os.path.join(ref, 'sevenn/pair_e3gnn/pair_d3_pars.h')).read()
cu = open(os.path.join(ref, 'sevenn/pair_e3gnn/pair_d3_for_ase.cu')).read()

NUM = r'[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?'


def macro(name):
    m = re.search(r'#define\s+' + name + r'\s+\{(.*?)\n\s*\}\s*\n', pars + '\n', re.S)
    return np.array([float(v) for v in re.findall(NUM, m.group(1).replace('\\', ' '))])


r0ab = macro('R0AB_TABLE').reshape(94, 94)
c6ab = macro('C6AB_TABLE').reshape(32385, 5)


def array(name):
    m = re.search(r'double\s+' + name + r'\[94\]\s*=\s*\{(.*?)\};', cu, re.S)
    return np.array([float(v) for v in re.findall(NUM, m.group(1))])


r2r4, rcov = array('r2r4_ref'), array('rcov_ref')
assert r2r4.shape == rcov.shape == (94,)


def functionals(func, defaults):
    body = cu[cu.index(f'void PairD3::{func}()'):]
    body = body[:body.index('\n}\n')]
    names = dict((k, int(v)) for k, v in re.findall(r'\{\s*"([^"]+)"\s*,\s*(\d+)\s*\}', body))
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    body = re.sub(r'//[^\n]*', '', body)
    rows = {}
    for code, stmts in re.findall(r'case\s+(\d+)\s*:(.*?)break\s*;', body, re.S):
        p = dict(defaults)
        for k, v in re.findall(r'(rs6|s18|rs18|s6|alp)\s*=\s*(' + NUM + r')\s*;', stmts):
            p[k] = float(v)
        rows[int(code)] = p
    table = {n: rows[c] for n, c in names.items() if c in rows}
    keys = sorted(table)
    return np.array(keys), np.array([[table[k][q] for q in ('s6', 'rs6', 's18', 'rs18', 'alp')] for k in keys])


blob = dict(r0ab=r0ab, c6ab=c6ab, r2r4=r2r4, rcov=rcov)
for damp, func, dflt in (('damp_zero', 'setfuncpar_zero', dict(s6=1.0, alp=14.0, rs18=1.0)),
                         ('damp_bj', 'setfuncpar_bj', dict(s6=1.0, alp=14.0)),
                         ('damp_zerom', 'setfuncpar_zerom', dict(s6=1.0, alp=14.0)),
                         ('damp_bjm', 'setfuncpar_bjm', dict(s6=1.0, alp=14.0))):
    names, vals = functionals(func, dflt)
    blob[damp + '_names'] = names
    blob[damp + '_params'] = vals
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, **blob)
print(out, os.path.getsize(out), 'bytes;', {k: v.shape for k, v in blob.items()})
i = list(blob['damp_bj_names']).index('pbe')
print('pbe / damp_bj (s6, rs6, s18, rs18, alp):', blob['damp_bj_params'][i])
