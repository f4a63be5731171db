"""CPU: invariants of the fused tensor-product kernel generator (sevennet_amd/codegen_fused.py) that the kernels
and the host-side fragment packer (csrc/snet_mlp.hip: pack_fused_slabs) both rely on."""
import re

import pytest

from sevennet_amd import codegen_fused
from sevennet_amd.shapes import aot_conv_specs

SPECS = aot_conv_specs()
FUSABLE = {tag: sp for tag, sp in SPECS.items() if codegen_fused.fusable(sp)}


def test_released_shapes_are_fusable():
    """every layer of the three benchmark models gets the fused kernels (channel multiplicities % 16 == 0)"""
    from sevennet_amd.model_spec import build_model_spec
    from sevennet_amd.shapes import sevennet_0_config, sevennet_l3i5_config, sevennet_mf_ompa_config
    for cfg in (sevennet_0_config(), sevennet_l3i5_config(), sevennet_mf_ompa_config()):
        for ls in build_model_spec(cfg).layers:
            assert codegen_fused.fusable(ls.conv), ls.conv.tag
    assert any(not codegen_fused.fusable(sp) for sp in SPECS.values())   # the 4-channel unit-test shapes are not


@pytest.mark.parametrize('tag', sorted(FUSABLE))
def test_sub_step_schedule_covers_every_weight_column_once(tag):
    """the weight stream: sub-steps of two 16-column tiles, every column of every path exactly once, tiles of one
    sub-step belong to the same x block and channel tile (they share the staged source-row slice)"""
    sp = FUSABLE[tag]
    cats, pairs_of, cols = codegen_fused.schedule(sp)
    seen = []
    for a, b in cols:
        assert a >= 0 and a % 16 == 0
        seen.extend(range(a, a + 16))
        if b >= 0:
            assert b % 16 == 0
            seen.extend(range(b, b + 16))
    assert sorted(seen) == list(range(sp.weight_numel))
    # stream order = for x block: for channel tile: for path pair
    k = 0
    for cat, pairs in zip(cats, pairs_of):
        paths_of_cat = {pi for pi, _ in cat.paths}
        for ct in range(cat.mul // 16):
            for pa, pb in pairs:
                assert pa in paths_of_cat and (pb is None or pb in paths_of_cat)
                assert cols[k][0] == sp.paths[pa].w_off + 16 * ct
                assert cols[k][1] == (sp.paths[pb].w_off + 16 * ct if pb is not None else -1)
                k += 1
    assert k == len(cols)


@pytest.mark.parametrize('tag', sorted(FUSABLE)[:4] + ['22d6a77ad5ac', '1cad2f51cbd0'])
def test_generated_source_is_consistent(tag):
    sp = FUSABLE[tag]
    src = codegen_fused.gen_conv_fused(sp)
    _, _, cols = codegen_fused.schedule(sp)
    m = re.search(r'constexpr int DX = (\d+), DOUT = (\d+), NSH = (\d+), NSHP = (\d+), WN = (\d+), NS = (\d+);', src)
    dx, dout, nsh, nshp, wn, ns = map(int, m.groups())
    assert (dx, dout, nsh, wn, ns) == (sp.irreps_x.dim, sp.irreps_out.dim, sp.irreps_sh.dim, sp.weight_numel, len(cols))
    # spherical-harmonics staging rows of the forward kernel: padded exactly when four rows span whole bank periods
    assert nshp == (nsh + 1 if (4 * nsh) % 32 == 0 else nsh)
    tab = re.search(r'SUB_COLS\[NS \* 2\] = \{([^}]*)\}', src).group(1)
    assert [int(v) for v in tab.split(',')] == [c for ab in cols for c in ab]
    # one reverse and one forward body per path, both kernels, the registrar
    for pi in range(len(sp.paths)):
        assert f'void bwdf_p{pi}(' in src and f'void fwdf_p{pi}(' in src
    assert f'conv_bwdf_{tag}' in src and f'conv_fwdf_{tag}' in src and 'FusedRegistrar registrar' in src
    # every g_out entry a reverse body reads is listed in its x block's fetch table
    for line in re.findall(r'static const int32_t GOFF\d+_\d+\[16\] = \{([^}]*)\}', src):
        vals = [int(v) for v in line.split(',')]
        assert all(v == -1 or 0 <= v < dout for v in vals)


@pytest.mark.parametrize('tag', sorted(FUSABLE))
def test_reverse_schedule_covers_every_weight_column_once(tag):
    """the reverse kernel's own weight stream (schedule_bwd): blocks of one or two channel tiles of an x block; an x block
    with an odd number of paths pairs the partnerless path's tiles of two consecutive channel tiles in one sub-step, so no
    sub-step of such a block has an empty second tile"""
    sp = FUSABLE[tag]
    blocks, cols = codegen_fused.schedule_bwd(sp)
    seen = sorted(c + i for ab in cols for c in ab if c >= 0 for i in range(16))
    assert seen == list(range(sp.weight_numel))
    _, _, cols_f = codegen_fused.schedule(sp)
    assert len(cols) <= len(cols_f)
    k = 0
    for b in blocks:
        cat, U = b['cat'], b['U']
        assert b['ncb'] * U == cat.mul // 16
        paths = {pi for pi, _ in cat.paths}
        if U == 2:
            assert len(paths) % 2 == 1 and all(tb is not None for _, tb in b['steps'])
            assert b['steps'][-1][0][0] == b['steps'][-1][1][0] and (b['steps'][-1][0][1], b['steps'][-1][1][1]) == (0, 1)
        for cb in range(b['ncb']):
            for ta, tb in b['steps']:
                assert ta[0] in paths and cols[k][0] == sp.paths[ta[0]].w_off + 16 * (U * cb + ta[1])
                assert cols[k][1] == (-1 if tb is None else sp.paths[tb[0]].w_off + 16 * (U * cb + tb[1]))
                k += 1
    assert k == len(cols)
    src = codegen_fused.gen_conv_fused(sp)
    tab = re.search(r'SUB_COLS_B\[\d+\] = \{([^}]*)\}', src).group(1)
    assert [int(v) for v in tab.split(',')] == [c for ab in cols for c in ab]
    assert f'NSUB = {len(cols)};' in src and '(size_t)NSUB * LPS * 64' in src


def test_reverse_schedule_sub_step_counts_of_sevennet_0():
    """SevenNet-0: middle layers 30 sub-steps (34 with one channel tile per block), last layer 7 (14); the first layer
    (a single x block) keeps 16"""
    counts = {tag: len(codegen_fused.schedule_bwd(FUSABLE[tag])[1]) for tag in ('22d6a77ad5ac', '005c575f8ec2', 'ecc5d202727d')}
    assert counts == {'22d6a77ad5ac': 30, '005c575f8ec2': 7, 'ecc5d202727d': 16}


def test_packed_tile_definition_properties():
    """the greedy packed-tile list (tests/helpers.packed_tiles_expected, what snet_edge_tiles_packed is checked against on the GPU):
    every edge in exactly one tile, at most 16 edges and two rows per tile, never more tiles than the per-row list, the 28-neighbour
    crystal loses no lane (7 tiles per 4 rows), and lists of adjacent row ranges chain into one list"""
    import random
    from helpers import packed_tiles_expected
    rnd = random.Random(11)
    for trial in range(60):
        n = rnd.randint(1, 70)
        kind = trial % 4
        deg = [28] * n if kind == 0 else [rnd.choice([0, 0, 1, 2, 3]) for _ in range(n)] if kind == 1 else \
            [rnd.randint(0, 45) for _ in range(n)] if kind == 2 else [rnd.choice([0, 16, 17, 32, 5]) for _ in range(n)]
        rp = [0]
        for d in deg:
            rp.append(rp[-1] + d)
        center = [i for i, d in enumerate(deg) for _ in range(d)]
        cut = rnd.randint(0, n)
        for lo, hi in ((0, n), (0, cut), (cut, n)):
            e0, nodes = packed_tiles_expected(rp, lo, hi)
            nt = len(e0) - 1
            assert len(nodes) == 2 * nt and e0[0] == rp[lo] and e0[-1] == rp[hi]
            for t in range(nt):
                assert 0 < e0[t + 1] - e0[t] <= 16
                assert set(center[e0[t]:e0[t + 1]]) == set(nodes[2 * t:2 * t + 2])
            assert nt <= sum((d + 15) // 16 for d in deg[lo:hi])
        a, na = packed_tiles_expected(rp, 0, cut)
        b, nb = packed_tiles_expected(rp, cut, n)
        assert a[-1] == b[0]                      # the interior list's sentinel is the boundary list's first edge
        if kind == 0 and n % 4 == 0:
            assert len(packed_tiles_expected(rp, 0, n)[0]) - 1 == 7 * n // 4


def test_work_list_format_per_shape():
    """which reverse kernels take packed two-row tiles (DESIGN 4b viii): the two-wave middle-layer class and the narrow lmax-3 shapes;
    not the first / last layers (measured slower there), not the wide lmax-3 shapes (no registers for the second g_out set)"""
    import re
    from sevennet_amd import codegen_fused
    from sevennet_amd.shapes import aot_conv_specs
    specs = aot_conv_specs(())
    mode = {}
    for tag in ('22d6a77ad5ac', 'ecc5d202727d', '005c575f8ec2', '0b99ee053764', '1cad2f51cbd0', '2d3e65aea6f3', '0431c055196e',
                '707b9944ff82', '502c5bba8e99'):
        src = codegen_fused.gen_conv_fused(specs[tag])
        mode[tag] = int(re.search(r'launch_bwd, launch_fwd, (\d)\}', src).group(1))
        packed = 'tile_node[2 * t + 1]' in src
        assert packed == bool(mode[tag])                         # the kernel body and the advertised format agree
        assert ('s_g[NWV][2][NGP * 16 + 16]' in src) == packed   # two g_out sets, 16 floats apart, only in the packed form
    assert mode == {'22d6a77ad5ac': 1, 'ecc5d202727d': 0, '005c575f8ec2': 0, '0b99ee053764': 0, '1cad2f51cbd0': 0, '2d3e65aea6f3': 0,
                    '0431c055196e': 0, '707b9944ff82': 0, '502c5bba8e99': 1}


def test_tuning_options_are_off_by_default_and_generate(monkeypatch):
    """The kernel-tuning generator options that stay in the generator (phase stamps of either kernel: measurement instruments,
    DESIGN 4f) must not leak into the shipped source and must still generate; the one forward-kernel change that shipped (short
    tiles spread over the lane groups, `frow`) must be what the default emits.  The measured-and-dropped variants of rounds 3-5
    (pipe / sgb / stag / bgrp / fpers / fpre / wfirst / gpf / fpf / tpold / nobr) left the generator in round 6
    (tools/experiments/r05_generator_variants.patch restores them): setting their options changes nothing."""
    from sevennet_amd import codegen
    spec = SPECS['22d6a77ad5ac']
    base = codegen_fused.gen_conv_fused(spec)
    fwd = base[base.index('void conv_fwdf_'):]
    assert 'stamp(' not in base and 'snet_stamps' not in base and 'snet_debug_stamps' not in base
    assert 'vb += gridDim.x' not in base and 'pre_rows' not in base
    assert 'rows_t[2]' in fwd and 'edge_of_row(tl, c)' in fwd and 'rows > 3' in base          # frow on
    assert base.count('const int rows)') == len(spec.paths)                                      # every forward body takes the row count
    for opts, must in (({'frow': '0'}, ['return 16 * tl + row; }']),
                       ({'stampf': spec.tag}, ['snet_debug_stamps', 'snet_stamps[n_raw * 16 + i]']),
                       ({'stampl': spec.tag}, ['snet_debug_stamps', 'snet_stamps[t_raw * 16 + i]'])):
        for k, v in opts.items():
            monkeypatch.setitem(codegen.OPTS, k, v)
        src = codegen_fused.gen_conv_fused(spec)
        for m in must:
            assert m in src, (opts, m)
        assert src.count('{') == src.count('}'), opts
        for k in opts:
            monkeypatch.delitem(codegen.OPTS, k)
    for k, v in (('pipe', '2'), ('sgb', '8'), ('stag', '3'), ('bgrp', '3'), ('fpers', '1'), ('fpre', '1'), ('wfirst', '1'), ('gpf', '1'),
                 ('fpf', '2'), ('tpold', '1'), ('nobr', '1')):
        monkeypatch.setitem(codegen.OPTS, k, v)
    assert codegen_fused.gen_conv_fused(spec) == base
    for k in ('pipe', 'sgb', 'stag', 'bgrp', 'fpers', 'fpre', 'wfirst', 'gpf', 'fpf', 'tpold', 'nobr'):
        monkeypatch.delitem(codegen.OPTS, k)
    assert codegen_fused.gen_conv_fused(spec) == base


def test_forward_short_tile_row_order():
    """The forward kernel's row order (`edge_of_row` / `row_edge` in the generated source): edge k of a tile of m edges sits in row
    4 (k // R) + k % R with R = ceil(m / 4).  What the kernel relies on: every edge has exactly one row, no row r >= R of any lane group
    holds an edge (so the bodies may skip it in all 64 lanes), the weight mask `g R + r < m` marks exactly the occupied rows, and a
    full tile keeps the identity order."""
    for m in range(0, 17):
        R = (m + 3) // 4
        rows = {}
        for row in range(16):
            g, r = row >> 2, row & 3
            k = g * R + r                      # edge_of_row(tl, row) - 16 tl
            if r < R and k < m:                # the kernel's mask: r < rows_t && g * rows_t + r < m_t
                assert k not in rows
                rows[k] = row
        assert sorted(rows) == list(range(m))                       # every edge exactly once
        assert all((row & 3) < R for row in rows.values())          # nothing in the skipped rows
        if m == 16:
            assert all(rows[k] == k for k in range(16))
        if m == 12:                                                 # the diamond-cubic second tile: rows {0, 1, 2} of every group
            assert sorted(rows.values()) == [4 * g + r for g in range(4) for r in range(3)]


def test_generated_sources_are_the_committed_ones():
    """Tripwire around generator refactors: the source generated for every ahead-of-time fused shape must be
    byte-identical to the one whose sha1 is committed in tests/golden/fused_kernel_sha1.json.  A deliberate kernel change
    regenerates the table (tools/update_kernel_hashes.py) in the same commit; a restructuring of the generator must not move it."""
    import hashlib
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'fused_kernel_sha1.json')) as f:
        want = json.load(f)
    got = {tag: hashlib.sha1(codegen_fused.gen_conv_fused(sp).encode()).hexdigest() for tag, sp in FUSABLE.items()}
    assert sorted(got) == sorted(want)
    assert [t for t in got if got[t] != want[t]] == []


def test_generator_functions_stay_reviewable():
    """no function of the fused-kernel generator over 300 lines (it was one 1 462-line function through round 5)"""
    import ast
    import inspect
    tree = ast.parse(inspect.getsource(codegen_fused))
    long = [(n.name, n.end_lineno - n.lineno + 1) for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.end_lineno - n.lineno + 1 > 300]
    assert long == []


def test_engine_functions_stay_reviewable():
    """the Python host's evaluation is a sequence of phase methods (compute was one 385-line function through round 6): nothing in
    engine.py over 200 lines"""
    import ast
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sevennet_amd', 'engine.py')
    tree = ast.parse(open(path).read())
    long = [(n.name, n.end_lineno - n.lineno + 1) for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.end_lineno - n.lineno + 1 > 200]
    assert long == []
