"""GPU: HIP force engine (through the C ABI) vs the CPU oracle and the golden fixtures."""
import numpy as np
import pytest
import torch

from helpers import irmul_to_mulir, load_ts_golden, oracle_model, synthetic_system

pytestmark = pytest.mark.gpu

F_TOL = 1e-4  # eV/A, BASELINE.json north_star tolerance (absolute)


def f_tol(scale):
    """the north-star force bar: 1e-4 eV/A ABSOLUTE.  (Rounds 2-4 scaled it up above 30 eV/A of force; no test system is that
    large -- the MD-scale systems pin max|F| = 8 eV/A -- and the knob is gone: a system with larger forces fails here and has
    to state its own tolerance.)"""
    assert float(scale) <= 30.0, f'force scale {float(scale):.3g} eV/A: outside the range the 1e-4 eV/A bar is tested on'
    return F_TOL


def _engine(cfg, sd):
    from sevennet_amd.engine import HipForceEngine
    return HipForceEngine(cfg, sd, device='cuda:0')


def _run(cfg, sd, types, ei, ev, keep=False):
    from sevennet_amd.engine import build_graph
    eng = _engine(cfg, sd)
    g = build_graph(types, ei, ev, device='cuda:0', num_species=eng.spec.num_species)
    out = eng.compute(g, want_atomic_virial=True, keep=keep)
    torch.cuda.synchronize()
    return eng, out


def _close(a, b, rel, floor, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    tol = max(floor, rel * np.abs(b).max()) if b.size else floor
    err = np.abs(a - b).max() if b.size else 0.0
    assert err <= tol, f'{what}: max err {err:.3e} > tol {tol:.3e} (scale {np.abs(b).max() if b.size else 0:.3e})'


def _fp32_class(got, ref, ref32, floor_rel, what):
    """The engine replaces an fp32 PyTorch evaluation: its error against the fp64 oracle must stay within 1.5x the error
    the fp32 PyTorch oracle itself makes on the same inputs (max norm), or below `floor_rel` of the quantity's scale
    (the floor only guards against an accidentally tiny fp32 error)."""
    got, ref, ref32 = (np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float64) for v in (got, ref, ref32))
    got = got.reshape(ref.shape)
    err, err32, scale = np.abs(got - ref).max(), np.abs(ref32.reshape(ref.shape) - ref).max(), np.abs(ref).max()
    assert err <= max(1.5 * err32, floor_rel * scale), f'{what}: engine error {err:.3e} vs fp32 PyTorch error {err32:.3e} (scale {scale:.3e})'
    return err, err32


def _energy_fp32_class(e_per_atom, ea, ref, n_ref, n_unit=None):
    """energies of a system that carries the oracle's per-atom values (ea[replica, n_unit]; the oracle's first n_unit atoms
    are one replica): total within max(1.5x the fp32
    PyTorch error, rtol 1e-5 -- the reference's own LAMMPS-vs-ASE energy bar, tests/lammps_tests/test_lammps.py:201-214);
    atomic energies within max(1.5x the fp32 PyTorch error, 2e-5 of their scale)"""
    r32 = ref['fp32']
    e64, e32 = float(ref['energy']) / n_ref, float(r32['energy']) / n_ref
    assert abs(e_per_atom - e64) <= max(1.5 * abs(e32 - e64), 1e-5 * abs(e64)), (e_per_atom - e64, e32 - e64, e64)
    n_unit = n_ref if n_unit is None else n_unit
    a64 = ref['atomic_energy'].numpy()
    d32 = r32['atomic_energy'].double().numpy() - a64          # the fp32 oracle's errors on all n_ref atoms of its cell
    d = ea - a64[None, :n_unit]                                # the engine's, on every replica
    # thousands of replicas against n_ref samples: compare the rms (sample-size independent); the largest of ~1e5 samples of
    # the same distribution sits at ~4.5 sigma, the largest of 64 at ~2.4 sigma -- hence 6 sigma for the maximum
    rms, rms32 = float(np.sqrt((d ** 2).mean())), float(np.sqrt((d32 ** 2).mean()))
    floor = 2e-5 * np.abs(a64).max()
    assert rms <= max(1.5 * rms32, floor), (rms, rms32)
    assert np.abs(d).max() <= max(1.5 * np.abs(d32).max(), 1.5 * 6.0 * rms32, floor), (np.abs(d).max(), np.abs(d32).max(), rms32)


def _compare(eng, out, ref, n, rel=3e-5, check_inter=True):
    """fp32 engine vs fp64 oracle: errors relative to each quantity's scale, never looser than the north-star 1e-4 eV/A on
    forces.  With ref['fp32'] (the fp32 PyTorch oracle on the same inputs: _md_scale_state) energies, atomic energies and
    forces must be fp32-PyTorch class (_fp32_class); without it the energy bar is SURVEY.md 8d's 1e-6 eV per atom, stated
    for unit rescale."""
    if 'fp32' in ref:
        r32 = ref['fp32']
        _fp32_class(out['energy'], ref['energy'].reshape(1), r32['energy'].reshape(1), 1e-5, 'energy')   # floor: test_lammps.py rtol
        _fp32_class(out['atomic_energy'], ref['atomic_energy'], r32['atomic_energy'], 2e-5, 'atomic_energy')
        _fp32_class(out['forces'], ref['forces'], r32['forces'], 2e-6, 'forces')
    else:
        _close(out['energy'], ref['energy'].reshape(1), 1e-6, 1e-6 * n, 'energy')
        _close(out['atomic_energy'], ref['atomic_energy'], rel, 5e-6, 'atomic_energy')
    _close(out['dE_dr'], ref['dE_dr'], rel, 1e-8, 'dE_dr')
    _close(out['forces'], ref['forces'], rel, 1e-8, 'forces')
    assert np.abs(out['forces'].cpu().numpy() - ref['forces'].numpy()).max() < f_tol(ref['forces'].abs().max().item())
    _close(out['virial'], ref['virial'], rel, 1e-7, 'virial')
    _close(out['atomic_virial'], ref['atomic_virial'], rel, 1e-8, 'atomic_virial')
    if check_inter and 'inter' in out:
        # module-by-module features: 1e-5 of each tensor's scale for the fp32-class kernels (the engine default
        # f16x3 included: tests/unit_tests/test_flash.py:96-125 is the reference's own bar for an accelerated
        # convolution); only an engine explicitly built with bf16x3 / bf16 in-kernel products gets 4e-5 from its
        # first fused convolution on
        loose = [t for t, L in enumerate(eng.layers) if getattr(L, 'fused_fwd', False) and eng.fused_terms < 3]
        first = loose[0] if loose else len(eng.layers)
        for t, L in enumerate(eng.layers):
            ls = L.spec
            for key, irr in ((f'{t}_si1', ls.si1.irreps_out), (f'{t}_conv', ls.conv.irreps_out),
                             (f'{t}_gate_in', ls.gate.irreps_in), (f'{t}_x', ls.gate.irreps_out)):
                downstream = t > first or (t == first and not key.endswith('_si1'))
                got, want = irmul_to_mulir(out['inter'][key], irr), ref['inter'][key]
                if key.endswith('_conv') and ls.w_cols is not None:
                    # the engine does not evaluate paths whose output block the following linear ignores
                    # (model_spec.prune_unread_paths): those columns of its buffer are never written
                    live = torch.zeros(1, irr.dim)
                    for p in ls.conv.paths:
                        for m3 in range(2 * p.l3 + 1):
                            live[0, p.out_off + m3 * p.out_mul + p.out_ch:p.out_off + m3 * p.out_mul + p.out_ch + p.mul] = 1.0
                    live = irmul_to_mulir(live, irr)[0] > 0
                    assert 0 < int(live.sum()) < irr.dim
                    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
                    got, want = got[:, live], want[:, live]
                _close(got, want, 4e-5 if downstream else 1e-5, 1e-9, key)


@pytest.mark.parametrize('name', ['hfo2_12', 'hfo_rs64', 'hfo2_96'])
def test_engine_vs_reference_torchscript_outputs(name):
    """the reference's own deployed model: E / F on the fixtures produced by running it"""
    d, cfg, sd = load_ts_golden(name)
    eng, out = _run(cfg, sd, d['types'], d['edge_index'], d['out_edge_vec'], keep=True)
    n = len(d['types'])
    assert abs(float(out['energy'].cpu()) - float(d['out_energy'])) / n < 1e-5
    assert np.abs(out['forces'].cpu().numpy() - d['out_forces']).max() < F_TOL
    assert np.abs(out['atomic_energy'].cpu().numpy() - d['out_atomic_energy']).max() < 1e-4
    vol = abs(np.linalg.det(d['cell']))
    assert np.abs(out['virial'].cpu().numpy() / vol - d['out_stress']).max() < 1e-5
    assert np.abs(out['inter']['edge_embedding'].cpu().numpy() - d['out_edge_embedding']).max() < 2e-6
    assert np.abs(out['inter']['edge_attr'].cpu().numpy() - d['out_edge_attr']).max() < 2e-6
    assert np.abs(out['dE_dr'].cpu().numpy() - d['par_dE_dr']).max() < F_TOL
    # module-by-module against the fp64 oracle (reference bar: atol 1e-6 features on O(1) values)
    ref = oracle_model(cfg, sd).forward(d['types'], d['edge_index'], d['out_edge_vec'].astype(np.float64), keep=True)
    _compare(eng, out, ref, n)


CASES = {
    'unit_o3_l2': dict(cfg='unit', over={}, cutoff=4.0, nsp=4),
    'unit_o3_l3': dict(cfg='unit', over={'lmax': 3}, cutoff=4.0, nsp=4),
    'unit_so3_l2_linear': dict(cfg='unit', over={'is_parity': False, 'self_connection_type': 'linear'}, cutoff=4.0, nsp=4),
    'mini_7net0': dict(cfg='mini', over={}, cutoff=5.0, nsp=2),
    # model options of the reference path the engine used to refuse (VERDICT r3 missing #4): o3.Linear biases
    # (sevenn/model_build.py:468,518), FCN readout (nn/linear.py:145-180), every activation of sevenn/_const.py:33-47
    'unit_bias_fcn_ssp_abs': dict(cfg='unit', over={'use_bias_in_linear': True, 'readout_as_fcn': True, 'readout_fcn_activation': 'elu',
                                                    'act_radial': 'ssp', 'act_scalar': {'e': 'ssp', 'o': 'abs'},
                                                    'act_gate': {'e': 'ssp', 'o': 'abs'}}, cutoff=4.0, nsp=4),
    'unit_fcn_relu_radial_sigmoid': dict(cfg='unit', over={'readout_as_fcn': True, 'readout_fcn_hidden_neurons': [16],
                                                           'act_radial': 'sigmoid'}, cutoff=4.0, nsp=4),
    'mini_bias_radial_elu': dict(cfg='mini', over={'use_bias_in_linear': True, 'act_radial': 'elu'}, cutoff=5.0, nsp=2),
}


@pytest.mark.parametrize('case', list(CASES))
def test_engine_vs_oracle_synthetic_weights(case):
    from sevennet_amd.shapes import mini_sevennet_0_config, unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    c = CASES[case]
    cfg = unit_test_config(**c['over']) if c['cfg'] == 'unit' else mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=7)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.08, seed=11, cutoff=c['cutoff'], n_species=c['nsp'])
    eng, out = _run(cfg, sd, types, ei, ev, keep=True)
    ref = oracle_model(cfg, sd).forward(types, ei, ev, keep=True)
    _compare(eng, out, ref, len(types))


def test_fused_kernels_with_other_radial_activation_and_biases():
    """SevenNet-0 shape (fused tensor-product kernels) with act_radial = ssp -- the reverse kernel's hidden-layer tail only
    carries silu / tanh, so g_h2 leaves the kernel and the separate hidden-layer reverse runs -- and o3.Linear biases in the
    species tables, the SI1 / SI2 constant rows and the folded readout: vs the fp64 oracle"""
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config(num_species=2)
    cfg.update(act_radial='ssp', use_bias_in_linear=True)
    sd = random_state_dict(cfg, seed=5)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.06, seed=2, cutoff=5.0, n_species=2)
    eng, out = _run(cfg, sd, types, ei, ev, keep=True)
    assert all(L.fused_fwd and L.fused_bwd and not L.mlp_tail for L in eng.layers)
    ref = oracle_model(cfg, sd).forward(types, ei, ev, keep=True)
    _compare(eng, out, ref, len(types), rel=2e-5)


def test_engine_unsorted_edges_and_empty_graph():
    from sevennet_amd.engine import build_graph
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = unit_test_config()
    sd = random_state_dict(cfg, seed=1)
    types, pos, cell, ei, ev = synthetic_system((1, 1, 1), sigma=0.05, seed=3, cutoff=4.0, n_species=4)
    perm = np.random.default_rng(0).permutation(ei.shape[1])
    eng = _engine(cfg, sd)
    a = eng.compute(build_graph(types, ei, ev, device='cuda:0', num_species=4))
    b = eng.compute(build_graph(types, ei[:, perm], ev[perm], device='cuda:0', num_species=4))
    assert abs(float(a['energy'].cpu()) - float(b['energy'].cpu())) < 1e-6 * abs(float(a['energy'].cpu()))
    _close(b['dE_dr'], a['dE_dr'].cpu().numpy()[perm], 1e-6, 1e-9, 'dE_dr (permuted edges)')
    # isolated atom: no edges (reference: convolution.py:265-268)
    iso = eng.compute(build_graph(np.array([2]), np.zeros((2, 0), np.int64), np.zeros((0, 3)), device='cuda:0', num_species=4))
    ref = oracle_model(cfg, sd).forward(np.array([2]), np.zeros((2, 0), np.int64), np.zeros((0, 3)))
    assert abs(float(iso['energy'].cpu()) - float(ref['energy'])) < 1e-4
    assert iso['forces'].abs().max().item() == 0.0


def test_sevennet_0_shape_vs_oracle_small_cell():
    """BASELINE config 1: SevenNet-0 shape on 64-atom Si, synthetic weights; fp32 GPU vs fp64 oracle"""
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, seed=0)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.05, seed=0, cutoff=5.0)
    eng, out = _run(cfg, sd, types, ei, ev, keep=True)
    ref = oracle_model(cfg, sd).forward(types, ei, ev, keep=True)
    # synthetic N(0,1) weights give O(1e2..1e4) energies/forces: compare relative to the force scale
    _compare(eng, out, ref, len(types), rel=2e-5)


@pytest.mark.parametrize('world', [2, 8])
def test_engine_bricks_equal_single_graph_on_one_gpu(world):
    """N bricks (threads sharing cuda:0, in-process halo with the same pack/unpack kernels and
    exchange semantics as the RCCL path) == the un-split evaluation; reference analogue:
    tests/lammps_tests/test_lammps.py:540-578."""
    import threading
    from sevennet_amd.engine import build_graph
    from sevennet_amd.parallel import InProcessHaloGroup, build_brick_graph
    from sevennet_amd.shapes import mini_sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=9)
    types, pos, cell, ei, ev = synthetic_system((4, 4, 4), sigma=0.06, seed=4, cutoff=5.0, n_species=2)
    eng = _engine(cfg, sd)
    ref = eng.compute(build_graph(types, ei, ev, device='cuda:0'))
    torch.cuda.synchronize()
    bricks = [build_brick_graph(pos, cell, types, 5.0, world, r, neighbors=(ei, ev)) for r in range(world)]
    grp = InProcessHaloGroup(bricks, 'cuda:0')
    engines = [_engine(cfg, sd) for _ in range(world)]
    results, errors = [None] * world, []

    def run(r):
        try:
            b = bricks[r]
            g = build_graph(b.types, b.edge_index, b.edge_vec, n_local=b.n_local, n_interior=b.n_interior, device='cuda:0')
            results[r] = engines[r].compute(g, halo=grp.members[r])
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            grp.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errors, errors
    F = np.zeros((len(types), 3), np.float32)
    Ea = np.zeros(len(types), np.float32)
    e_tot = 0.0
    for b, r in zip(bricks, results):
        F[b.global_ids[:b.n_local]] = r['forces'].cpu().numpy()
        Ea[b.global_ids[:b.n_local]] = r['atomic_energy'].cpu().numpy()
        e_tot += float(r['energy'].cpu())
        assert sum(b.recv_counts) > 0
    assert abs(e_tot - float(ref['energy'].cpu())) < 2e-6 * abs(float(ref['energy'].cpu()))
    _close(Ea, ref['atomic_energy'], 1e-5, 1e-7, 'atomic energies (bricks)')
    _close(F, ref['forces'], 2e-5, 1e-8, 'forces (bricks)')


def test_calculator_surface_matches_reference_results():
    """b2 boundary: SevenNetCalculator-compatible results (keys, units, signs, Voigt order) on the
    fixture produced by the reference's own deployed model; reference errors preserved."""
    from sevennet_amd.calculator import SevenNetCalculator
    d, cfg, sd = load_ts_golden('hfo2_12')
    cfg = dict(cfg, _type_map={72: 0, 8: 1})
    with pytest.raises(ValueError):
        SevenNetCalculator((cfg, sd), file_type='torchscript')
    with pytest.raises(ValueError):
        SevenNetCalculator('/nonexistent/checkpoint.pth')
    calc = SevenNetCalculator((cfg, sd), file_type='model_instance', device='cuda:0', compute_atomic_virial=True,
                              enable_flash=True)
    assert calc.implemented_properties == ['free_energy', 'energy', 'forces', 'stress', 'stresses', 'energies']
    numbers = np.where(d['types'] == 0, 72, 8)
    res = calc.compute(numbers, d['pos'], d['cell'], [True, True, True])
    assert res['num_edges'] == d['edge_index'].shape[1]
    assert abs(res['energy'] - float(d['out_energy'])) < 1e-5 * len(numbers)
    assert res['free_energy'] == res['energy']
    assert np.abs(res['forces'] - d['out_forces']).max() < F_TOL
    assert np.abs(res['energies'] - d['out_atomic_energy']).max() < 1e-4
    assert np.abs(res['stress'] - (-d['out_stress'][[0, 1, 2, 4, 5, 3]])).max() < 1e-5
    assert res['stresses'].shape == (len(numbers), 6)
    with pytest.raises(ValueError, match='do not know atomic number'):
        calc.compute(np.array([14]), np.zeros((1, 3)), d['cell'], [True] * 3)


def test_rccl_backend_halo_exchange_world1():
    """The torch.distributed (backend 'nccl' = RCCL) exchange path on device tensors.  One GPU is
    all this box has, so world_size = 1 with a self-exchange: rank 0 'sends' rows to itself, which
    exercises init_process_group, uneven-split all_to_all_single on a tensor view and the HIP
    pack / unpack kernels exactly as the N-GPU bench does."""
    import os
    import torch.distributed as dist
    from sevennet_amd.parallel import HaloExchange
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        n_local, dim = 50, 96
        send = np.array([3, 7, 7 + 1, 20, 49])  # rows this rank exports (here: to itself)
        halo = HaloExchange([send], [len(send)], 'cuda:0')
        g = torch.Generator().manual_seed(0)
        x = torch.randn(n_local + len(send), dim, generator=g).to('cuda:0')
        ref = x.clone()
        halo.forward(x, n_local)
        torch.cuda.synchronize()
        assert torch.equal(x[n_local:], ref[send]) and torch.equal(x[:n_local], ref[:n_local])
        gx = torch.randn(n_local + len(send), dim, generator=g).to('cuda:0')
        gref = gx.clone()
        halo.reverse(gx, n_local)
        torch.cuda.synchronize()
        exp = gref[:n_local].clone()
        exp[send] += gref[n_local:]
        assert torch.allclose(gx[:n_local], exp)
        # split exchange with independent work on the compute stream in between (what the engine does)
        x2, gx2 = ref.clone(), gref.clone()
        hf = halo.forward_start(x2, n_local)
        busy = torch.randn(2048, 2048, device='cuda:0') @ torch.randn(2048, 2048, device='cuda:0')
        halo.forward_finish(hf)
        hr = halo.reverse_start(gx2, n_local)
        busy = busy @ busy
        halo.reverse_finish(hr, gx2)
        torch.cuda.synchronize()
        assert torch.equal(x2, x) and torch.allclose(gx2[:n_local], exp)
    finally:
        dist.destroy_process_group()


def test_reference_checkpoint_weights_cp0():
    """The reference's own test checkpoint (tests/data/checkpoints/cp_0.pth: trained Hf/O model,
    channel 4, lmax 2, O(3) parity, 3 layers) fed unchanged (config + state_dict names) to the
    engine; the oracle runs the same tensors in fp64."""
    import json
    from helpers import GOLDEN
    d = np.load(f'{GOLDEN}/cp0_state.npz')
    cfg = json.loads(str(d['__config__']))
    sd = {k: d[k] for k in d.files if not k.startswith('__')}
    g = np.load(f'{GOLDEN}/ts_oracle_hfo2_96.npz')  # 96-atom HfO2 structure shipped with the reference
    from sevennet_amd.neighbor import neighbor_list
    ei, ev, _ = neighbor_list(g['pos'], g['cell'], [True] * 3, cfg['cutoff'])
    eng, out = _run(cfg, sd, g['types'], ei, ev, keep=True)
    ref = oracle_model(cfg, sd).forward(g['types'], ei, ev, keep=True)
    _compare(eng, out, ref, len(g['types']))
    assert ref['forces'].abs().max().item() > 1e-2  # non-trivial forces


@pytest.mark.parametrize('case', ['si_444', 'si_222', 'si_111', 'hfo2_triclinic'])
def test_gpu_neighbor_list_matches_host(case):
    """f1: device cell-list graph build == host KD-tree build (same edge multiset, incl. the
    multi-image cases nb = 1, 2 bins per axis and a triclinic cell); engine results agree."""
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.neighbor_gpu import build_graph_gpu, gpu_neighbor_supported
    if case.startswith('si_'):
        r = int(case[3])
        pos, cell = diamond_cubic(5.431, (r, r, r), 0.1, 7)
        pos = pos + 3.7  # some atoms outside the cell: wrapping must not change edge vectors
        cutoff, types = 5.0, np.zeros(len(pos), np.int64)
    else:
        g = np.load(f'{__import__("helpers").GOLDEN}/ts_oracle_hfo2_96.npz')
        pos, cell, cutoff, types = g['pos'], g['cell'], 4.0, g['types']
    assert gpu_neighbor_supported(cell, [True] * 3, cutoff)
    ei, ev, S = neighbor_list(pos, cell, [True] * 3, cutoff)
    gg = build_graph_gpu(types, pos, cell, cutoff, with_shifts=True)
    torch.cuda.synchronize()
    assert gg.n_edges == ei.shape[1]
    c, s, v, sh = gg.center.cpu().numpy(), gg.src.cpu().numpy(), gg.edge_vec.cpu().numpy(), gg.shifts.cpu().numpy()
    assert (np.diff(c) >= 0).all() and (np.bincount(c, minlength=len(pos)) == np.diff(gg.row_ptr.cpu().numpy())).all()

    def canon(i, j, S_, vec):
        key = np.lexsort((S_[:, 2], S_[:, 1], S_[:, 0], j, i))
        return i[key], j[key], S_[key], vec[key]
    a = canon(ei[0], ei[1], S, ev)
    b = canon(c.astype(np.int64), s.astype(np.int64), sh.astype(np.int64), v.astype(np.float64))
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
    assert np.abs(a[3] - b[3]).max() < 2e-6


def _same_edges(pos, cell, pbc, cutoff):
    """device cell list == host list as multisets of (center, source, image shift), edge vectors to fp32"""
    from sevennet_amd.neighbor import neighbor_list
    from sevennet_amd.neighbor_gpu import build_graph_gpu, gpu_neighbor_supported
    assert gpu_neighbor_supported(cell, pbc, cutoff, pos)
    ei, ev, S = neighbor_list(pos, cell, pbc, cutoff)
    gg = build_graph_gpu(np.zeros(len(pos), np.int64), pos, cell, cutoff, with_shifts=True, pbc=pbc)
    torch.cuda.synchronize()
    assert gg.n_edges == ei.shape[1], (gg.n_edges, ei.shape[1])
    c, s, v, sh = gg.center.cpu().numpy(), gg.src.cpu().numpy(), gg.edge_vec.cpu().numpy(), gg.shifts.cpu().numpy()
    assert (np.diff(c) >= 0).all() and (np.bincount(c, minlength=len(pos)) == np.diff(gg.row_ptr.cpu().numpy())).all()

    def canon(i, j, S_, vec):
        key = np.lexsort((S_[:, 2], S_[:, 1], S_[:, 0], j, i))
        return i[key], j[key], S_[key], vec[key]
    if gg.n_edges:
        a = canon(ei[0], ei[1], S, ev)
        b = canon(c.astype(np.int64), s.astype(np.int64), sh.astype(np.int64), v.astype(np.float64))
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
        assert np.abs(a[3] - b[3]).max() < 2e-6
    return gg


def test_gpu_neighbor_list_open_axes_and_thin_cells():
    """f1 outside bulk crystals (VERDICT r3 missing #5): slabs, wires and molecules (open axes: neither wrapped nor
    imaged; a missing cell row is padded like sevenn/train/dataload.py:37-48), periodic cells THINNER than the cutoff
    (several images of the cell itself), and the reference's own pinned edge counts (tests/unit_tests/test_data.py:48,
    cutoff 4.0: bulk NaCl 36, H2O 6, H 0, one-atom Cu cell 18)"""
    from sevennet_amd.neighbor import diamond_cubic
    pos, cell = diamond_cubic(5.431, (3, 3, 2), 0.1, 5)
    pos = pos + np.array([2.0, -1.5, 30.0])      # atoms outside the cell; the open axis keeps its raw coordinate
    for pbc in ([True, True, False], [True, False, False], [False, True, True], [False, False, False]):
        _same_edges(pos, cell, pbc, 5.0)
    # triclinic slab with a degenerate (zero) cell row on the open axis
    tri = np.array([[9.0, 0.0, 0.0], [2.5, 8.0, 0.0], [0.0, 0.0, 0.0]])
    rng = np.random.default_rng(2)
    p2 = rng.uniform(0, 1, (60, 3)) @ np.array([[9.0, 0.0, 0.0], [2.5, 8.0, 0.0], [0.0, 0.0, 7.0]])
    _same_edges(p2, tri, [True, True, False], 4.5)
    # periodic cells thinner than the cutoff: 1, 2 and 3 images per side, also triclinic
    for scale, pbc in ((0.9, [True] * 3), (0.45, [True] * 3), (0.3, [True, True, False]), (0.6, [True, False, True])):
        small = np.array([[1.0, 0.1, 0.0], [0.2, 1.1, 0.1], [0.0, 0.3, 0.95]]) * (scale * 5.0)
        p3 = rng.uniform(0, 1, (5, 3)) @ small
        _same_edges(p3, small, pbc, 5.0)
    # the reference's pins
    a = 5.63
    g = _same_edges(np.array([[0, 0, 0], [a / 2] * 3]), np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]]), [True] * 3, 4.0)
    assert g.n_edges == 36
    a = 3.61
    g = _same_edges(np.zeros((1, 3)), np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]]), [True] * 3, 4.0)
    assert g.n_edges == 18
    h2o = np.array([[0, 0, 0.119262], [0, 0.763239, -0.477047], [0, -0.763239, -0.477047]])
    assert _same_edges(h2o, np.zeros((3, 3)), [False] * 3, 4.0).n_edges == 6
    assert _same_edges(np.zeros((1, 3)), np.zeros((3, 3)), [False] * 3, 4.0).n_edges == 0


def test_sevennet_l3i5_shape_vs_oracle_small_cell():
    """BASELINE config 4 shape (lmax 3, 34 paths per middle layer) on a small cell, fp32 GPU vs fp64 oracle"""
    from sevennet_amd.model_spec import sevennet_l3i5_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_l3i5_config()
    sd = random_state_dict(cfg, seed=1)
    types, pos, cell, ei, ev = synthetic_system((1, 1, 2), sigma=0.3, seed=3, cutoff=5.0)
    eng, out = _run(cfg, sd, types, ei, ev, keep=True)
    ref = oracle_model(cfg, sd).forward(types, ei, ev, keep=True)
    _compare(eng, out, ref, len(types), rel=2e-5)


@pytest.mark.parametrize('case', ['bulk', 'tiny_cell', 'brick'])
def test_undirected_pair_map_and_shared_radial_weights(case):
    """snet_edge_pairs: every directed edge finds its reverse (same atoms, opposite vector), pairs are
    numbered densely, ghost-source edges stay singletons; running the radial MLP once per pair leaves
    energies and forces unchanged (the two directions have the same |r|)."""
    from sevennet_amd.engine import build_graph
    from sevennet_amd.parallel import build_brick_graph
    from sevennet_amd.shapes import mini_sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=3)
    reps = (1, 1, 1) if case == 'tiny_cell' else (3, 3, 3)
    types, pos, cell, ei, ev = synthetic_system(reps, sigma=0.06, seed=8, cutoff=5.0, n_species=2)
    n_local = None
    if case == 'brick':
        b = build_brick_graph(pos, cell, types, 5.0, 2, 0, neighbors=(ei, ev))
        types, ei, ev, n_local = b.types, b.edge_index, b.edge_vec, b.n_local
    g = build_graph(types, ei, ev, n_local=n_local, device='cuda:0')
    g0 = build_graph(types, ei, ev, n_local=n_local, device='cuda:0', share_pairs=False)
    assert g0.w_row is None and g.w_row is not None
    E = g.n_edges
    w_row, pe = g.w_row.cpu().numpy(), g.pair_edge.cpu().numpy()
    c, s, v = g.center.cpu().numpy(), g.src.cpu().numpy(), g.edge_vec.cpu().numpy()
    assert len(pe) == g.n_pairs and w_row.min() == 0 and w_row.max() == g.n_pairs - 1
    assert np.array_equal(w_row[pe], np.arange(g.n_pairs))          # pair_edge[p] is a member of pair p
    counts = np.bincount(w_row, minlength=g.n_pairs)
    assert counts.max() <= 2
    for p in np.nonzero(counts == 2)[0][:2000]:
        a, bb = np.nonzero(w_row == p)[0]
        assert c[a] == s[bb] and s[a] == c[bb] and np.abs(v[a] + v[bb]).max() <= 2e-5
    singles = np.nonzero(counts == 1)[0]
    if case == 'brick':
        assert len(singles) > 0 and (s[pe[singles]] >= g.n_local).all()   # reverse edge lives on the peer
    else:
        assert len(singles) == 0 and g.n_pairs * 2 == E
    if case != 'brick':
        eng = _engine(cfg, sd)
        a, bb = eng.compute(g), eng.compute(g0)
        torch.cuda.synchronize()
        assert abs(float(a['energy'].cpu()) - float(bb['energy'].cpu())) <= 1e-6 * abs(float(bb['energy'].cpu()))
        _close(a['forces'], bb['forces'], 2e-6, 1e-8, 'forces (pair-shared radial weights)')
        _close(a['dE_dr'], bb['dE_dr'], 2e-6, 1e-8, 'dE_dr (pair-shared radial weights)')


@pytest.mark.parametrize('modal', ['x1', 'x2'])
def test_multi_modal_model_vs_oracle(modal):
    """multi-fidelity model (all four modal-patched linears, modal-wise shift): Python host and native
    sequencer against the oracle's literal one-hot concatenation, for both channels"""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    from oracle.model import OracleModel
    cfg = unit_test_config(use_modality=True, _number_of_modalities=2, _modal_map={'x1': 0, 'x2': 1},
                           use_modal_node_embedding=True, use_modal_self_inter_intro=True,
                           use_modal_self_inter_outro=True, use_modal_output_block=True,
                           use_modal_wise_shift=True, use_modal_wise_scale=False)
    sd = random_state_dict(cfg, seed=12)
    sd['rescale_atomic_energy.shift'] = np.array([[0.1, -0.2, 0.3, 0.0], [1.5, 2.5, -3.5, 0.5]], np.float32)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.08, seed=5, cutoff=4.0, n_species=4)
    eng = HipForceEngine(cfg, sd, device='cuda:0', modal=modal)
    g = build_graph(types, ei, ev, device='cuda:0', num_species=eng.spec.num_species)
    out = eng.compute(g, want_atomic_virial=True, keep=True)
    nat = NativeModel(cfg, sd, modal=modal).compute(g, want_atomic_virial=True)
    torch.cuda.synchronize()
    ref = OracleModel(cfg, sd, dtype=torch.float64, modal=modal).forward(types, ei, ev, keep=True)
    _compare(eng, out, ref, len(types))
    for k in ('energy', 'atomic_energy', 'dE_dr', 'forces', 'virial'):
        assert torch.equal(out[k], nat[k]), k
    other = OracleModel(cfg, sd, dtype=torch.float64, modal='x2' if modal == 'x1' else 'x1').forward(types, ei, ev)
    assert abs(float(other['energy']) - float(ref['energy'])) > 1e-3   # the channels really differ
    with pytest.raises(ValueError, match='modal'):
        HipForceEngine(cfg, sd, device='cuda:0')


def test_sevennet_mf_ompa_shape_vs_oracle_small_cell():
    """BASELINE config 5 shape: O(3) parity, lmax 3, 119-species `nequip` self-connection, two fidelity
    channels, modal-wise shift, cutoff 6 (XPLOR 5.5); 4-species decoration of a 64-atom cell,
    synthetic weights; fp32 GPU vs fp64 oracle"""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.model_spec import sevennet_mf_ompa_config
    from sevennet_amd.synthetic import random_state_dict
    from oracle.model import OracleModel
    cfg = sevennet_mf_ompa_config()
    sd = random_state_dict(cfg, seed=0)
    sd['rescale_atomic_energy.shift'] = np.random.default_rng(1).standard_normal((2, 119)).astype(np.float32)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.05, seed=0, cutoff=6.0)
    types = np.random.default_rng(4).choice(np.array([3, 8, 14, 22]), size=len(types))
    eng = HipForceEngine(cfg, sd, device='cuda:0', modal='omat24')
    g = build_graph(types, ei, ev, device='cuda:0', num_species=119)
    out = eng.compute(g, want_atomic_virial=True, keep=True)
    torch.cuda.synchronize()
    ref = OracleModel(cfg, sd, dtype=torch.float64, modal='omat24').forward(types, ei, ev, keep=True)
    _compare(eng, out, ref, len(types), rel=2e-5)


def test_bench_multi_rank_path_dry_run():
    """`bench.py --gpus N` end to end (brick decomposition, split halo exchange inside the timed step,
    energy all-reduce) with 2 and 4 ranks sharing this box's one GPU over gloo (SNET_DIST_BACKEND): a
    functional dry run of the path the 8-GPU RCCL bench takes; the total energy must equal the
    single-process one."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ['--reps', '6', '--steps', '1', '--warmup', '1', '--no-cpu-baseline']
    env = dict(os.environ, SNET_DIST_BACKEND='gloo')

    def run(cmd):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
        return json.loads(line)

    one = run([sys.executable, 'bench.py', '--brick-proxy', '2'] + common)
    # the brick-proxy leg: rank 0's brick of a 2-way decomposition timed alone on this GPU, both hosts
    bp = one['brick_proxy']
    assert bp['world'] == 2 and 0 < bp['atoms_local'] < one['config']['atoms'] and bp['ghost_rows'] > 0 and bp['interior_atoms'] >= 0
    assert bp['ms_per_step'] > 0 and bp['ideal_ms'] > 0 and bp['dispatches_per_step'] > 10 and bp['ms_per_step_native_host'] > 0
    for n, port in ((2, 29611), (4, 29612)):
        many = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
                    '--master-addr', '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', str(n)] + common)
        assert many['n_gpus'] == n and many['config']['atoms'] == one['config']['atoms']
        assert many['config']['edges'] == one['config']['edges']
        assert abs(many['config']['energy'] - one['config']['energy']) <= 1e-9 * abs(one['config']['energy'])
        # the start-up self-check of the decomposition travels in the line
        sc = many['config']['decomposition_selfcheck']
        assert sc['ok'] and sum(sc['owned_atoms_by_rank']) == one['config']['atoms'] and len(sc['ghost_rows_by_rank']) == n
        assert sc['ghost_rows_by_rank'][0] == many['config']['ghost_rows_rank0'] and min(sc['peers_by_rank']) >= 1
        # value = atoms / median step; the contract's bracket (mean) rides along
        # value = the contract's bracket (atoms / ms_per_step); the median-based statistic sits beside it
        assert many['ms_per_step_median'] > 0 and abs(many['value'] - many['config']['atoms'] / (many['ms_per_step'] * 1e-3)) < 1e-6 * many['value']
        assert abs(many['value_median'] - many['config']['atoms'] / (many['ms_per_step_median'] * 1e-3)) < 1e-6 * many['value_median']
        assert 'sclk_mhz' in many['config'] and 'socket_power_w' in many['config'] and many['config']['telemetry_samples'] >= 0
    # the world-1 soak of the N > 1 path on REAL RCCL: the communicator reports itself (ncclCommCount)
    soak = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
                '--master-port', '29613', 'bench.py', '--gpus', '1', '--dist-path', '--halo', 'native'] + common)
    assert soak['config']['decomposition_selfcheck']['rccl_comm'] == {'nranks': 1, 'rank0_user_rank': 0}


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: real RCCL at world size 2')
def test_bench_two_ranks_real_rccl():
    """the day a box has two GPUs this runs `bench.py --gpus 2` on REAL RCCL (one rank per GPU, the library's own
    send / recv groups over xGMI, interior / boundary split around every exchange, both hosts): the total energy must equal the
    single-process one and the line must carry the halo fields.  Skipped on the one-GPU boxes of this pool."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ['--reps', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']

    def run(cmd):
        r = subprocess.run(cmd, cwd=root, env=dict(os.environ), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    one = run([sys.executable, 'bench.py'] + common)
    for host in ('python', 'native'):
        two = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                   '--master-port', '29631', 'bench.py', '--gpus', '2', '--host', host] + common)
        assert two['n_gpus'] == 2 and two['config']['atoms'] == one['config']['atoms'] and two['config']['edges'] == one['config']['edges']
        assert abs(two['config']['energy'] - one['config']['energy']) <= 1e-9 * abs(one['config']['energy'])
        assert 'RCCL' in two['config']['halo'] and two['config']['halo_exchanges_per_step'] == 9
        assert two['config']['halo_exposed_ms'] is not None
        assert two['config']['decomposition_selfcheck']['rccl_comm']['nranks'] == 2


MD_FMAX = 8.0  # eV/A: largest force component of the MD-scale parity systems


def _md_scale_state(cfg, sd, types, ei, ev, modal=None):
    """seeded synthetic weights give max|F| ~ 0.03 eV/A at rescale scale = 1, where the north-star bar of 1e-4 eV/A
    absolute is a 0.3 % relative bar.  Returns (state dict, fp64 oracle result) with `rescale_atomic_energy.scale`
    chosen so that the oracle's largest force component is MD_FMAX; ref['fp32'] = the fp32 PyTorch oracle on the same inputs."""
    from oracle.model import OracleModel
    r0 = OracleModel(cfg, sd, dtype=torch.float64, modal=modal).forward(types, ei, ev)
    k = MD_FMAX / float(r0['forces'].abs().max())
    sd = dict(sd)
    sd['rescale_atomic_energy.scale'] = (np.asarray(sd['rescale_atomic_energy.scale'], np.float64) * k).astype(np.float32)
    ref = OracleModel(cfg, sd, dtype=torch.float64, modal=modal).forward(types, ei, ev, keep=True)
    assert abs(float(ref['forces'].abs().max()) - MD_FMAX) < 1e-3 * MD_FMAX
    ref['fp32'] = OracleModel(cfg, sd, dtype=torch.float32, modal=modal).forward(types, ei, ev)   # the arithmetic the engine replaces
    return sd, ref


@pytest.mark.parametrize('model', ['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
def test_md_scale_forces_within_1e4_absolute_small_cell(model):
    """North-star tolerance where it means something: 64-atom cells of the three released shapes with forces of
    MD magnitude (max|F| = 8 eV/A).  Engine default (fused kernels, f16x3 in-kernel products) vs the fp64 oracle:
    |dF| <= 1e-4 eV/A ABSOLUTE and <= 1e-5 of max|F|; every intermediate feature tensor within 1e-5 of its scale
    (reference bar for an accelerated convolution: tests/unit_tests/test_flash.py:96-125)."""
    from bench import model_config
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config(model)
    multi = bool(cfg.get('use_modality'))
    modal = 'mpa' if multi else None
    sd = random_state_dict(cfg, seed=0)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.05 if model == 'sevennet_0' else 0.1, seed=0,
                                                cutoff=cfg['cutoff'])
    if multi:
        types = np.random.default_rng(4).choice(np.array([3, 8, 14, 22]), size=len(types))
    sd, ref = _md_scale_state(cfg, sd, types, ei, ev, modal)
    eng = HipForceEngine(cfg, sd, device='cuda:0', modal=modal)
    assert eng.fused_mode == 'f16x3' and all(L.fused_fwd and L.fused_bwd for L in eng.layers)
    g = build_graph(types, ei, ev, device='cuda:0', num_species=eng.spec.num_species)
    out = eng.compute(g, want_atomic_virial=True, keep=True)
    torch.cuda.synchronize()
    dF = np.abs(out['forces'].cpu().numpy() - ref['forces'].numpy()).max()
    assert dF < 1e-4 and dF < 1e-5 * MD_FMAX, dF
    _compare(eng, out, ref, len(types), rel=1e-5)


def test_bf16_compute_mode_whole_model_mf_ompa_shape():
    """BASELINE config 5's STATED mode -- "bf16 compute with fp32 force accumulation" = `fused_terms=2` (two bf16 terms per operand,
    three matrix-core products; bench.py --terms 2) -- as a whole-model test (VERDICT r4 weak #1a): SevenNet-MF-ompa shape,
    4-species 64-atom cell, forces at MD scale (max|F| = 8 eV/A), against the fp64 oracle.  Tolerance of SURVEY.md 8(d) config 5:
    1e-3 eV/A absolute on forces (measured 4e-5, profiles/r03_terms_accuracy*.txt: the bar is asserted 5x tighter than stated, at
    2e-4); energy rtol 1e-5.  The fp32-class default mode on the same system must stay inside 1e-4 and be the more accurate one."""
    from bench import model_config
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config('sevennet_mf_ompa')
    sd = random_state_dict(cfg, seed=0)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.1, seed=0, cutoff=cfg['cutoff'])
    types = np.random.default_rng(4).choice(np.array([3, 8, 14, 22]), size=len(types))
    sd, ref = _md_scale_state(cfg, sd, types, ei, ev, 'mpa')
    f_ref = ref['forces'].numpy()
    err = {}
    for mode in ('bf16x3', 'f16x3'):
        eng = HipForceEngine(cfg, sd, device='cuda:0', modal='mpa', fused_terms=mode)
        assert eng.fused_mode == mode and eng.fused_terms == {'bf16x3': 2, 'f16x3': 4}[mode]
        assert all(L.fused_fwd and L.fused_bwd for L in eng.layers)
        g = build_graph(types, ei, ev, device='cuda:0', num_species=eng.spec.num_species)
        out = eng.compute(g)
        torch.cuda.synchronize()
        err[mode] = float(np.abs(out['forces'].cpu().numpy() - f_ref).max())
        assert abs(float(out['energy'].cpu()) - float(ref['energy'])) <= 1e-5 * abs(float(ref['energy']))
        assert np.abs(out['forces'].cpu().numpy().astype(np.float64).sum(0)).max() < 1e-3
    assert err['bf16x3'] < 1e-3 and err['bf16x3'] < 2e-4, err      # config 5's bar, and what the mode actually delivers
    assert err['f16x3'] < 1e-4 and err['f16x3'] <= err['bf16x3'], err


@pytest.mark.parametrize('n_tile', [11, 23])
def test_sevennet_0_full_size_equals_tiled_small_cell(n_tile):
    """BASELINE config 2 / 3 sizes (SevenNet-0 shape, 11^3 x 8 = 10 648 and 23^3 x 8 = 97 336 atoms, GPU
    neighbor list) through a
    size-independent property: the big cell is an exact n^3 tiling of a rattled 8-atom cell, so every
    replica must carry the forces / atomic energies of the 2^3 tiling (64 atoms), which the fp64 oracle
    can evaluate; the total force must vanish."""
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import neighbor_list
    from sevennet_amd.neighbor_gpu import build_graph_gpu
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, seed=0)
    a = 5.431
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]]) * a
    unit = basis + np.random.default_rng(7).normal(0.0, 0.05, basis.shape)

    def tile(n):
        g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), -1).reshape(-1, 3) * a
        return (g[:, None, :] + unit[None, :, :]).reshape(-1, 3), np.eye(3) * n * a

    pos_s, cell_s = tile(2)
    ei, ev, _ = neighbor_list(pos_s, cell_s, [True] * 3, cfg['cutoff'])
    sd, ref = _md_scale_state(cfg, sd, np.zeros(len(pos_s), np.int64), ei, ev)   # max|F| = 8 eV/A
    f_unit = ref['forces'].numpy()[:8]          # replica (0,0,0) of the 2^3 tiling
    e_unit = ref['atomic_energy'].numpy()[:8]
    pos, cell = tile(n_tile)
    n_big = len(pos)
    assert n_big == 8 * n_tile ** 3
    eng = HipForceEngine(cfg, sd, device='cuda:0')
    g = build_graph_gpu(np.zeros(len(pos), np.int64), pos, cell, cfg['cutoff'], device='cuda:0')
    assert g.n_edges == ei.shape[1] // 64 * n_big and g.n_pairs * 2 == g.n_edges
    out = eng.compute(g)
    torch.cuda.synchronize()
    F = out['forces'].cpu().numpy().reshape(-1, 8, 3)
    Ea = out['atomic_energy'].cpu().numpy().reshape(-1, 8)
    scale = max(1.0, np.abs(f_unit).max())
    assert np.abs(F - f_unit[None]).max() < 1e-4, np.abs(F - f_unit[None]).max()   # absolute, at MD-scale forces
    _energy_fp32_class(float(out['energy'].cpu()) / n_big, Ea, ref, 64, 8)
    assert np.abs(out['forces'].cpu().numpy().astype(np.float64).sum(0)).max() < 1e-3 * scale


@pytest.mark.parametrize('model', ['sevennet_0', 'sevennet_l3i5'])
def test_fused_engine_equals_separate_kernels(model):
    """whole step with the radial MLP's last layer inside the tensor-product kernels (no w / g_w in memory)
    == the separate-kernel path on the same graph: energy, forces, dE/dr, virial"""
    from bench import model_config
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config(model)
    sd = random_state_dict(cfg, 3)
    pos, cell = diamond_cubic(5.431, (3, 3, 3), 0.08, 5)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    g = build_graph(np.zeros(len(pos), np.int64), ei, ev, device='cuda:0')
    a = HipForceEngine(cfg, sd, device='cuda:0', fused=True)
    assert all(L.fused_fwd and L.fused_bwd for L in a.layers)
    b = HipForceEngine(cfg, sd, device='cuda:0', fused=False)
    ra, rb = a.compute(g), b.compute(g)
    torch.cuda.synchronize()
    n = len(pos)
    assert abs(ra['energy'].item() - rb['energy'].item()) / n < 2e-6
    fs = max(1.0, rb['forces'].abs().max().item())
    assert (ra['forces'] - rb['forces']).abs().max().item() < 2e-5 * fs
    assert (ra['dE_dr'] - rb['dE_dr']).abs().max().item() < 2e-5 * max(1.0, rb['dE_dr'].abs().max().item())
    assert (ra['virial'] - rb['virial']).abs().max().item() < 2e-5 * max(1.0, rb['virial'].abs().max().item())
    # mixed modes agree too
    for mode in ('fwd', 'bwd'):
        rc = HipForceEngine(cfg, sd, device='cuda:0', fused=mode).compute(g)
        assert (rc['forces'] - rb['forces']).abs().max().item() < 2e-5 * fs


@pytest.mark.parametrize('model', ['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
def test_transposed_last_layer_equals_per_edge_rows(model):
    """last interaction layer (scalar outputs only): source-row gradient as a forward convolution of the transposed
    product over the source-grouped edges == per-edge g_xe rows + segment sum; with and without the undirected-pair map"""
    from bench import model_config
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config(model)
    sd = random_state_dict(cfg, 4)
    modal = 'mpa' if cfg.get('use_modality') else None
    pos, cell = diamond_cubic(5.431, (3, 3, 2), 0.08, 6)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    a = HipForceEngine(cfg, sd, device='cuda:0', modal=modal)
    b = HipForceEngine(cfg, sd, device='cuda:0', modal=modal, transposed_conv=False)
    assert [L.tplan is not None for L in a.layers] == [False] * (len(a.layers) - 1) + [True]
    assert all(L.tplan is None for L in b.layers)
    for pair_map in (True, False):
        g = build_graph(np.zeros(len(pos), np.int64), ei, ev, device='cuda:0', share_pairs=pair_map,
                        num_species=a.spec.num_species)
        ra, rb = a.compute(g), b.compute(g)
        torch.cuda.synchronize()
        assert (g.w_row is not None) == pair_map
        assert ra['energy'].item() == rb['energy'].item()
        fs = max(1.0, rb['forces'].abs().max().item())
        assert (ra['forces'] - rb['forces']).abs().max().item() < 1e-5 * fs
        assert (ra['virial'] - rb['virial']).abs().max().item() < 1e-5 * max(1.0, rb['virial'].abs().max().item())


def test_native_rccl_halo_self_exchange():
    """csrc/snet_halo.cpp on a world-1 RCCL communicator (one GPU): the rank sends rows to itself.  forward fills
    the ghost rows with the owners' rows; reverse adds ghost rows into their owners, duplicates summed in fixed
    order -- the semantics of pair_e3gnn_parallel.cpp:747-911 (pack / unpack hooks), checked against index ops.
    The N-rank plan itself is the one HaloExchange uses (tests/test_parallel_cpu.py, world 2 / 4 over gloo)."""
    from sevennet_amd.parallel import NativeHalo, RcclComm
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(5)
    n_local, dim = 50, 96
    send = torch.randint(0, n_local, (37,), generator=g).numpy()     # with repeats: several ghosts of one owner row
    comm = RcclComm(1, 0)
    halo = NativeHalo(comm, [send], [len(send)])
    assert halo.n_ghost == len(send)
    x = torch.randn(n_local + len(send), dim, generator=g).to(dev)
    x0 = x.clone()
    halo.forward(x, n_local)
    torch.cuda.synchronize()
    assert torch.equal(x[:n_local], x0[:n_local])
    assert torch.equal(x[n_local:], x0[torch.as_tensor(send, device=dev).long()])
    gx = torch.randn(n_local + len(send), dim, generator=g).to(dev)
    ref = gx.clone().double()
    ref[:n_local].index_add_(0, torch.as_tensor(send, device=dev).long(), ref[n_local:].clone())
    halo.reverse(gx, n_local)
    torch.cuda.synchronize()
    assert (gx[:n_local].double() - ref[:n_local]).abs().max() < 1e-5
    assert torch.equal(gx[n_local:].double(), ref[n_local:])            # ghost rows themselves are left alone
    # a second call out of the same buffers, other width
    y = torch.randn(n_local + len(send), 3, generator=g).to(dev)
    y0 = y.clone()
    halo.forward(y, n_local)
    torch.cuda.synchronize()
    assert torch.equal(y[n_local:], y0[torch.as_tensor(send, device=dev).long()])
    # split exchange: RCCL on the halo's second stream, the caller's stream waits in forward_finish
    z = torch.randn(n_local + len(send), dim, generator=g).to(dev)
    z0 = z.clone()
    pending = halo.forward_start(z, n_local)
    busy = torch.randn(512, 512, device=dev) @ torch.randn(512, 512, device=dev)   # work beside the transfer
    halo.forward_finish(pending)
    torch.cuda.synchronize()
    assert pending is not None and torch.equal(z[n_local:], z0[torch.as_tensor(send, device=dev).long()]) and busy.isfinite().all()
    e = torch.tensor([1.5, -2.0], dtype=torch.float64, device=dev)
    comm.all_reduce_f64(e)
    torch.cuda.synchronize()
    assert e.tolist() == [1.5, -2.0]
    with pytest.raises(RuntimeError, match='ghost row count'):
        halo.forward(torch.zeros(n_local + 1, 4, device=dev), n_local)
    # hosts that number their ghost rows themselves: k-th stream row -> ghost row perm[k]
    perm = torch.randperm(len(send), generator=g).numpy()
    hp = NativeHalo(comm, [send], [len(send)], recv_perm=perm)
    x = x0.clone()
    hp.forward(x, n_local)
    torch.cuda.synchronize()
    want = torch.empty_like(x0[n_local:])
    want[torch.as_tensor(perm, device=dev).long()] = x0[torch.as_tensor(send, device=dev).long()]
    assert torch.equal(x[n_local:], want)
    gx = torch.randn(n_local + len(send), dim, generator=g).to(dev)
    ref = gx.clone().double()
    ref[:n_local].index_add_(0, torch.as_tensor(send, device=dev).long(), ref[n_local:][torch.as_tensor(perm, device=dev).long()])
    hp.reverse(gx, n_local)
    torch.cuda.synchronize()
    assert (gx[:n_local].double() - ref[:n_local]).abs().max() < 1e-5
    with pytest.raises(RuntimeError, match='permutation'):
        NativeHalo(comm, [send], [len(send)], recv_perm=np.zeros(len(send), np.int32))


def _run_ranks(world, fn, hub):
    """fn(rank) on one host thread per rank (each with its own HIP stream); a failing rank aborts the hub"""
    import threading
    errors, results = [], [None] * world

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device='cuda:0')):
                results[r] = fn(r)
                torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((r, e))
            hub.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errors, errors
    return results


@pytest.mark.parametrize('world', [2, 3, 5])
def test_native_halo_multi_rank_plan_on_the_loopback_transport(world):
    """csrc/snet_halo.cpp with SEVERAL peers on one GPU: `world` host threads, one rank each, exchange through the
    library's in-process transport (device-to-device copies with ncclGroup semantics) -- the pack / land / permute /
    reverse-accumulate plan and its per-peer offsets are the code the RCCL path runs.  Random send lists with repeats,
    self-sends and empty pairs; one rank numbers its ghost rows in its own order (recv_perm)."""
    from sevennet_amd.parallel import LoopbackHub, NativeHalo
    dev = 'cuda:0'
    rng = np.random.default_rng(100 + world)
    n_local = [int(rng.integers(20, 60)) for _ in range(world)]
    send = [[rng.integers(0, n_local[r], int(rng.integers(0, 25))) if rng.random() > 0.2 else np.zeros(0, np.int64)
             for _ in range(world)] for r in range(world)]          # send[r][p]: rows of r that p holds as ghosts
    recv_counts = [[len(send[p][r]) for p in range(world)] for r in range(world)]
    n_ghost = [sum(c) for c in recv_counts]
    perm_rank = world - 1
    perm = rng.permutation(n_ghost[perm_rank])
    hub = LoopbackHub(world)
    halos = [NativeHalo(hub.comm(r), send[r], recv_counts[r], recv_perm=perm if r == perm_rank else None) for r in range(world)]
    for dim in (96, 5):
        g = torch.Generator().manual_seed(dim)
        x0 = [torch.randn(n_local[r] + n_ghost[r], dim, generator=g) for r in range(world)]
        gx0 = [torch.randn(n_local[r] + n_ghost[r], dim, generator=g) for r in range(world)]

        def fn(r):
            x, gx = x0[r].to(dev), gx0[r].to(dev)
            halos[r].forward(x, n_local[r])
            halos[r].reverse(gx, n_local[r])
            halos[r].forward(gx, n_local[r])      # a second forward on the reverse's result: buffers are reused in order
            return x.cpu(), gx.cpu()
        out = _run_ranks(world, fn, hub)
        gx_sum = []
        for r in range(world):
            stream = torch.cat([x0[p][:n_local[p]][torch.as_tensor(send[p][r]).long()] for p in range(world)]) \
                if n_ghost[r] else torch.zeros(0, dim)
            want = stream
            if r == perm_rank and n_ghost[r]:
                want = torch.empty_like(stream)
                want[torch.as_tensor(perm).long()] = stream
            assert torch.equal(out[r][0][:n_local[r]], x0[r][:n_local[r]])
            assert torch.equal(out[r][0][n_local[r]:], want), (world, dim, r)
            # reverse: my local rows += every ghost copy of them held by any peer (peer order, fp32 sums)
            acc = gx0[r][:n_local[r]].double().clone()
            for p in range(world):
                gp = gx0[p][n_local[p]:]
                if p == perm_rank and n_ghost[p]:
                    gp = gp[torch.as_tensor(perm).long()]       # back to peer-stream order
                off = sum(recv_counts[p][:r])
                rows = gp[off:off + len(send[r][p])]
                if len(rows):
                    acc.index_add_(0, torch.as_tensor(send[r][p]).long(), rows.double())
            gx_sum.append(acc)
        for r in range(world):
            assert (out[r][1][:n_local[r]].double() - gx_sum[r]).abs().max() < 1e-5, (world, dim, r)
            stream = torch.cat([gx_sum[p][torch.as_tensor(send[p][r]).long()] for p in range(world)]) if n_ghost[r] \
                else torch.zeros(0, dim, dtype=torch.float64)
            want = stream
            if r == perm_rank and n_ghost[r]:
                want = torch.empty_like(stream)
                want[torch.as_tensor(perm).long()] = stream
            assert (out[r][1][n_local[r]:].double() - want).abs().max() < 1e-5, (world, dim, r)


@pytest.mark.parametrize('world,host', [(2, 'native'), (4, 'native'), (8, 'native'), (4, 'python'), (4, 'python-inline')])
def test_bricks_through_the_native_halo_equal_single_graph(world, host):
    """The whole multi-rank evaluation path of `bench.py --gpus N` / a LAMMPS e3gnn/parallel run on one GPU: `world`
    bricks of one cell, one host thread each, ghost features and gradients exchanged by csrc/snet_halo.cpp (loopback
    transport) INSIDE snet_model_eval (native host, no Python callback) or from HipForceEngine.compute == the un-split
    evaluation (reference analogue: tests/lammps_tests/test_lammps.py:540-578, serial vs parallel pair style)."""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.parallel import LoopbackHub, NativeHalo, build_brick_graph
    from sevennet_amd.shapes import mini_sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=9)
    types, pos, cell, ei, ev = synthetic_system((4, 4, 4), sigma=0.06, seed=4, cutoff=5.0, n_species=2)
    ref = HipForceEngine(cfg, sd, device='cuda:0').compute(build_graph(types, ei, ev, device='cuda:0'))
    torch.cuda.synchronize()
    bricks = [build_brick_graph(pos, cell, types, 5.0, world, r, neighbors=(ei, ev)) for r in range(world)]
    hub = LoopbackHub(world)
    # 'python': the forward exchange runs on each rank's second stream beside the self-connection (forward_start /
    # forward_finish); 'python-inline': on the compute stream
    halos = [NativeHalo(hub.comm(r), bricks[r].send_lists, bricks[r].recv_counts, overlap=host != 'python-inline')
             for r in range(world)]
    models = [NativeModel(cfg, sd) if host == 'native' else HipForceEngine(cfg, sd, device='cuda:0') for _ in range(world)]

    def fn(r):
        b = bricks[r]
        g = build_graph(b.types, b.edge_index, b.edge_vec, n_local=b.n_local, n_interior=b.n_interior, device='cuda:0')
        if host == 'native':
            models[r].set_halo(halos[r])
            out = models[r].compute(g)
        else:
            out = models[r].compute(g, halo=halos[r])
        return float(out['energy'].cpu()), out['forces'].cpu().numpy()
    res = _run_ranks(world, fn, hub)
    F = np.zeros((len(types), 3), np.float32)
    for b, (_, f) in zip(bricks, res):
        F[b.global_ids[:b.n_local]] = f[:b.n_local]
    e_tot = sum(e for e, _ in res)
    e_ref = float(ref['energy'].cpu())
    assert abs(e_tot - e_ref) < 2e-6 * abs(e_ref)
    fr = ref['forces'].cpu().numpy()
    assert np.abs(F - fr).max() <= max(1e-8, 2e-5 * np.abs(fr).max())


@pytest.mark.parametrize('world', [2, 4])
def test_interior_boundary_split_of_the_fused_convolutions(world):
    """Halo overlap: bricks number their interior atoms first; the fused forward kernel runs the interior rows
    while the forward exchange is in flight and the boundary rows after it; in reverse the boundary tiles come first, the ghost
    rows of g_h travel while the interior tiles, the local rows and sc^T run (last layer: the transposed convolution's ghost rows
    first).  SevenNet-0 shape (fused kernels) through the library's halo over the in-process transport: must equal the
    un-split engine on the whole cell, and be BIT-IDENTICAL to the same bricks evaluated without the split."""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.parallel import LoopbackHub, NativeHalo, build_brick_graph
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config(num_species=2)
    sd = random_state_dict(cfg, seed=4)
    types, pos, cell, ei, ev = synthetic_system((4, 4, 3), sigma=0.06, seed=8, cutoff=5.0, n_species=2)
    ref = HipForceEngine(cfg, sd, device='cuda:0').compute(build_graph(types, ei, ev, device='cuda:0'))
    torch.cuda.synchronize()
    bricks = [build_brick_graph(pos, cell, types, 5.0, world, r, neighbors=(ei, ev)) for r in range(world)]
    for b in bricks:   # interior rows have no ghost source, every boundary row has one
        c, s_ = b.edge_index
        has_ghost = np.zeros(b.n_local, bool)
        has_ghost[c[s_ >= b.n_local]] = True
        assert not has_ghost[:b.n_interior].any() and has_ghost[b.n_interior:].all()
    assert any(0 < b.n_interior < b.n_local for b in bricks)
    from sevennet_amd.native_model import NativeModel
    out = {}
    class _CallbackHalo:   # the same exchange behind Python callbacks: the sequencer then cannot split (snet_model_set_halo)
        def __init__(self, h):
            self.h = h

        def forward(self, x, n_local):
            return self.h.forward(x, n_local)

        def reverse(self, gx, n_local):
            return self.h.reverse(gx, n_local)

    for split in (True, False, 'native', 'native_late'):
        hub = LoopbackHub(world)
        halos = [NativeHalo(hub.comm(r), bricks[r].send_lists, bricks[r].recv_counts) for r in range(world)]
        models = [NativeModel(cfg, sd) if str(split).startswith('native') else HipForceEngine(cfg, sd, device='cuda:0') for _ in range(world)]

        def fn(r):
            b = bricks[r]
            g = build_graph(b.types, b.edge_index, b.edge_vec, n_local=b.n_local, n_interior=b.n_interior, device='cuda:0')
            if split == 'native_late':
                # the graph is first evaluated UN-split (callback halo), its topology cached; then the library halo is
                # installed and the SAME graph evaluated again -- a cache hit that must not reuse a boundary tile list nobody built
                models[r].set_halo(_CallbackHalo(halos[r]))
                first = models[r].compute(g)
                torch.cuda.synchronize()
                models[r].set_halo(halos[r])
                o = models[r].compute(g)
                assert torch.equal(first['forces'], o['forces']) and torch.equal(first['energy'], o['energy'])
            elif split == 'native':      # the C++ sequencer: same split on its own second stream (snet_model_set_interior)
                models[r].set_halo(halos[r])
                o = models[r].compute(g)
            else:
                models[r].halo_split = split
                assert all(L.fused_fwd and L.fused_bwd for L in models[r].layers)
                o = models[r].compute(g, halo=halos[r])
            return float(o['energy'].cpu()), o['forces'].cpu().numpy(), o['dE_dr'].cpu().numpy()
        out[split] = _run_ranks(world, fn, hub)
    for (e1, f1, d1), (e0, f0, d0), (e2, f2, d2) in zip(out[True], out[False], out['native']):
        assert e1 == e0 and np.array_equal(f1, f0) and np.array_equal(d1, d0)
        assert e1 == e2 and np.array_equal(f1, f2) and np.array_equal(d1, d2)   # both hosts bit for bit
    for (e2, f2, d2), (e3, f3, d3) in zip(out['native'], out['native_late']):
        assert e2 == e3 and np.array_equal(f2, f3) and np.array_equal(d2, d3)
    F = np.zeros((len(types), 3), np.float32)
    for b, (_, f, _) in zip(bricks, out[True]):
        F[b.global_ids[:b.n_local]] = f[:b.n_local]
    e_ref = float(ref['energy'].cpu())
    assert abs(sum(e for e, _, _ in out[True]) - e_ref) < 2e-6 * abs(e_ref)
    fr = ref['forces'].cpu().numpy()
    assert np.abs(F - fr).max() <= max(1e-8, 2e-5 * np.abs(fr).max())


@pytest.mark.parametrize('host', ['native', 'python'])
def test_rank_without_ghosts_still_serves_its_peers(host):
    """A rank whose own graph has NO ghost rows (n_total == n_local) but whose rows a peer needs must still join every
    exchange: the halo is a group of send/recv pairs, and a rank that skips it leaves its peer's receive pending
    forever (round-2 advisor finding on csrc/snet_model.cpp: the hooks ran only when n_total > n_local).  Two bricks
    of one cell with every edge (center on rank 1, source on rank 0) removed: rank 1 then has recv = 0, send > 0.
    Must equal the un-split evaluation of the same directed edge list."""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.parallel import LoopbackHub, NativeHalo, assign_owners, build_brick_graph, processor_grid
    from sevennet_amd.shapes import mini_sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=10)
    types, pos, cell, ei, ev = synthetic_system((4, 3, 3), sigma=0.06, seed=5, cutoff=5.0, n_species=2)
    owner = assign_owners(pos, cell, processor_grid(2))
    keep = ~((owner[ei[0]] == 1) & (owner[ei[1]] == 0))
    ei, ev = ei[:, keep], ev[keep]
    ref = HipForceEngine(cfg, sd, device='cuda:0').compute(build_graph(types, ei, ev, device='cuda:0'))
    torch.cuda.synchronize()
    bricks = [build_brick_graph(pos, cell, types, 5.0, 2, r, neighbors=(ei, ev)) for r in range(2)]
    assert sum(bricks[1].recv_counts) == 0 and len(bricks[1].send_lists[0]) > 0 and sum(bricks[0].recv_counts) > 0
    hub = LoopbackHub(2)
    halos = [NativeHalo(hub.comm(r), bricks[r].send_lists, bricks[r].recv_counts) for r in range(2)]
    models = [NativeModel(cfg, sd) if host == 'native' else HipForceEngine(cfg, sd, device='cuda:0') for _ in range(2)]

    def fn(r):
        b = bricks[r]
        g = build_graph(b.types, b.edge_index, b.edge_vec, n_local=b.n_local, n_interior=b.n_interior, device='cuda:0')
        assert (g.n_total == g.n_local) == (r == 1)
        if host == 'native':
            models[r].set_halo(halos[r])
            out = models[r].compute(g)
        else:
            out = models[r].compute(g, halo=halos[r])
        return float(out['energy'].cpu()), out['forces'].cpu().numpy()
    res = _run_ranks(2, fn, hub)
    F = np.zeros((len(types), 3), np.float32)
    for b, (_, f) in zip(bricks, res):
        F[b.global_ids[:b.n_local]] = f[:b.n_local]
    e_ref = float(ref['energy'].cpu())
    assert abs(sum(e for e, _ in res) - e_ref) < 2e-6 * abs(e_ref)
    fr = ref['forces'].cpu().numpy()
    assert np.abs(F - fr).max() <= max(1e-8, 2e-5 * np.abs(fr).max())


@pytest.mark.parametrize('model,n_tile,mode', [('sevennet_l3i5', 19, 'f16x3'), ('sevennet_mf_ompa', 15, 'f16x3'),
                                               ('sevennet_mf_ompa', 29, 'f16x3'), ('sevennet_mf_ompa', 29, 'bf16x3')])
def test_lmax3_shapes_full_size_equal_tiled_small_cell(model, n_tile, mode):
    """BASELINE config 4 / 5 sizes through the size-independent tiling property: SevenNet-l3i5 shape at
    19^3 x 8 = 54 872 atoms and the SevenNet-MF-ompa shape (119 species, two fidelity channels, cutoff 6) at
    15^3 x 8 = 27 000 atoms and -- config 5's full workload on ONE GPU -- at 29^3 x 8 = 195 112 atoms (9.0 M edges).  The big cell is an exact n^3 tiling of a rattled, 4-species-decorated 8-atom cell, so
    every replica must carry the forces / atomic energies of the 2^3 tiling, which the fp64 oracle evaluates;
    the fused tensor-product kernels (engine default) run every layer of both shapes.  mode 'bf16x3' = config 5's stated
    "bf16 compute, fp32 force accumulation" (bench.py --terms 2) on its full 195 112-atom workload: SURVEY.md 8(d)'s bar for that
    configuration is 1e-3 eV/A (asserted at 2e-4); the default fp32-class mode keeps the 1e-4 eV/A bar."""
    from bench import model_config
    from oracle.model import OracleModel
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.neighbor import neighbor_list
    from sevennet_amd.neighbor_gpu import build_graph_gpu
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config(model)
    sd = random_state_dict(cfg, seed=0)
    multi = bool(cfg.get('use_modality'))
    modal = 'mpa' if multi else None
    a = 5.431
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]]) * a
    unit = basis + np.random.default_rng(11).normal(0.0, 0.08, basis.shape)
    z_unit = np.array([3, 8, 14, 22, 8, 3, 22, 14]) if multi else np.zeros(8, np.int64)

    def tile(n):
        g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), -1).reshape(-1, 3) * a
        return (g[:, None, :] + unit[None, :, :]).reshape(-1, 3), np.eye(3) * n * a, np.tile(z_unit, n ** 3)

    pos_s, cell_s, ty_s = tile(2)
    ei, ev, _ = neighbor_list(pos_s, cell_s, [True] * 3, cfg['cutoff'])
    sd, ref = _md_scale_state(cfg, sd, ty_s, ei, ev, modal)   # max|F| = 8 eV/A
    f_unit, e_unit = ref['forces'].numpy()[:8], ref['atomic_energy'].numpy()[:8]
    pos, cell, ty = tile(n_tile)
    n_big = len(pos)
    eng = HipForceEngine(cfg, sd, device='cuda:0', modal=modal, fused_terms=mode)
    assert eng.fused_mode == mode and all(L.fused_fwd and L.fused_bwd for L in eng.layers)
    g = build_graph_gpu(ty, pos, cell, cfg['cutoff'], device='cuda:0',
                        num_species=eng.spec.num_species if eng.needs_species_rows else 0)
    assert g.n_edges * 64 == ei.shape[1] * n_big
    out = eng.compute(g)
    torch.cuda.synchronize()
    F = out['forces'].cpu().numpy().reshape(-1, 8, 3)
    Ea = out['atomic_energy'].cpu().numpy().reshape(-1, 8)
    scale = max(1.0, np.abs(f_unit).max())
    if mode == 'bf16x3':   # config 5's own tolerance; energies to the reference's LAMMPS-vs-ASE rtol
        assert np.abs(F - f_unit[None]).max() < 2e-4, np.abs(F - f_unit[None]).max()
        assert abs(float(out['energy'].cpu()) / n_big - float(ref['energy']) / 64) <= 1e-5 * abs(float(ref['energy']) / 64)
        return
    assert np.abs(F - f_unit[None]).max() < 1e-4, (np.abs(F - f_unit[None]).max(), scale)   # absolute, at MD-scale forces
    _energy_fp32_class(float(out['energy'].cpu()) / n_big, Ea, ref, 64, 8)
    # net force: every replica repeats the same rounding, so the total grows with the number of replicas -- per replica
    # (8 atoms, max|F| = 8 eV/A) it must stay at the fp32 rounding of the forces
    assert np.abs(out['forces'].cpu().numpy().astype(np.float64).sum(0)).max() / (n_big / 8) < 4e-5


def test_amorphous_supercell_at_config4_size():
    """BASELINE config 4 as written -- SevenNet-l3i5 shape on an AMORPHOUS cell of > 50 000 atoms (ragged degrees at size):
    a 512-atom amorphous cell (sigma = 0.35 A, 1.8 A minimum distance; degrees 21 .. 32) tiled 5 x 5 x 5 = 64 000 atoms.
    Every replica must carry the forces / atomic energies the fp64 oracle finds for the periodic 512-atom cell; forces at
    MD scale (max|F| = 8 eV/A), absolute 1e-4 eV/A bar."""
    from bench import model_config
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.neighbor import amorphous_cell, neighbor_list
    from sevennet_amd.neighbor_gpu import build_graph_gpu
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config('sevennet_l3i5')
    sd = random_state_dict(cfg, seed=0)
    unit, cell_u = amorphous_cell(5.431, (4, 4, 4), 0.35, 3, 1.8)
    n_u = len(unit)
    ei, ev, _ = neighbor_list(unit, cell_u, [True] * 3, cfg['cutoff'])
    deg = np.bincount(ei[0], minlength=n_u)
    assert deg.max() - deg.min() >= 10            # not the uniform degree of a rattled crystal
    sd, ref = _md_scale_state(cfg, sd, np.zeros(n_u, np.int64), ei, ev)
    n_t = 5
    L = float(cell_u[0, 0])
    shifts = np.stack(np.meshgrid(*[np.arange(n_t)] * 3, indexing='ij'), -1).reshape(-1, 3) * L
    pos = (shifts[:, None, :] + np.mod(unit, L)[None, :, :]).reshape(-1, 3)
    n_big = len(pos)
    assert n_big == 64000
    eng = HipForceEngine(cfg, sd, device='cuda:0')
    g = build_graph_gpu(np.zeros(n_big, np.int64), pos, cell_u * n_t, cfg['cutoff'], device='cuda:0')
    assert g.n_edges == ei.shape[1] * n_t ** 3
    out = eng.compute(g)
    torch.cuda.synchronize()
    F = out['forces'].cpu().numpy().reshape(-1, n_u, 3)
    Ea = out['atomic_energy'].cpu().numpy().reshape(-1, n_u)
    f_ref, e_ref = ref['forces'].numpy(), ref['atomic_energy'].numpy()
    assert np.abs(F - f_ref[None]).max() < 1e-4, np.abs(F - f_ref[None]).max()
    _energy_fp32_class(float(out['energy'].cpu()) / n_big, Ea, ref, n_u)


@pytest.mark.gpu
def _benchmark_workload(model):
    """(cfg, pos, cell, types, modal) of bench.py's workload for `model` (SURVEY.md 8(d) configs 3, 4 and config 5's per-GPU share)"""
    from bench import model_config, species_of
    from sevennet_amd.neighbor import amorphous_cell, diamond_cubic
    cfg = model_config(model)
    if model == 'sevennet_l3i5':
        pos, cell = amorphous_cell(5.431, (19,) * 3, 0.35, 3, 1.8)
    elif model == 'sevennet_mf_ompa':
        pos, cell = diamond_cubic(5.431, (15,) * 3, 0.1, 4)
    else:
        pos, cell = diamond_cubic(5.431, (23,) * 3, 0.05, 2)
    assert len(pos) == {'sevennet_0': 97336, 'sevennet_l3i5': 54872, 'sevennet_mf_ompa': 27000}[model]
    return cfg, pos, np.asarray(cell, np.float64), species_of(cfg, len(pos)), ('mpa' if cfg.get('use_modality') else None)


@pytest.mark.parametrize('model', ['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
def test_symmetries_at_the_benchmark_size(model):
    """BASELINE config 3's, config 4's and config 5's (per-GPU share) workloads themselves (SevenNet-0 shape, 97 336 atoms; l3i5 shape
    -- lmax 3 --, 54 872-atom "amorphous" cell; MF-ompa shape -- 119 species, 4 of them present, fidelity channel, cutoff 6 A --, 27 000
    atoms: the cells bench.py times) through the properties an
    E(3)-equivariant, permutation- and translation-invariant potential has at ANY size (what the reference's model guarantees by
    construction, nn/convolution.py:118-141 + force_output.py:171-230): the total force vanishes; a rigid rotation of cell and positions
    leaves the energy, rotates the forces and conjugates the virial; a translation and a relabelling of the atoms change nothing.
    Each transformed system gets its own GPU neighbor list (a rotated cubic cell is a general triclinic one for the list builder) and its
    own edge order, so the comparison also crosses different summation orders: forces within twice the single-evaluation bar (1e-4 eV/A
    at max|F| = 8 eV/A, scaled to this system's max|F|), energy within 5e-7 and virial within 1e-6 of their magnitudes (the fp64 sums of
    fp32 terms move by 1e-8 relative between orders)."""
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.neighbor_gpu import build_graph_gpu
    from sevennet_amd.synthetic import random_state_dict
    cfg, pos, cell, types, modal = _benchmark_workload(model)
    eng = HipForceEngine(cfg, random_state_dict(cfg, seed=0), device='cuda:0', modal=modal)
    n = len(pos)
    ns = eng.spec.num_species if eng.needs_species_rows else 0

    def evaluate(p, c, ty=types):
        g = build_graph_gpu(ty, p, c, cfg['cutoff'], device='cuda:0', num_species=ns)
        out = eng.compute(g)
        torch.cuda.synchronize()
        v = out['virial'].cpu().numpy()          # xx yy zz xy yz zx (include/snet_hip.h)
        V = np.array([[v[0], v[3], v[5]], [v[3], v[1], v[4]], [v[5], v[4], v[2]]])
        return g.n_edges, float(out['energy'].cpu()), out['forces'].cpu().numpy().astype(np.float64), V

    ne0, e0, f0, v0 = evaluate(pos, cell)
    fmax, vmax = np.abs(f0).max(), np.abs(v0).max()
    # measured on an MI355X (SevenNet-0 shape): rotation |dE| 1.2e-5 eV of 1 759 eV, |dF| 3.6e-7 eV/A at max|F| = 0.083, |dV| 9.6e-5 of 4 460 eV;
    # relabelling + translation 4.3e-5 eV, 8.9e-8 eV/A, 1.0e-5 eV
    f_tol, e_tol, v_tol = 2 * 1e-4 * fmax / 8.0, 5e-7 * abs(e0), 1e-6 * vmax
    assert np.abs(f0.sum(0)).max() < 1e-3 * max(1.0, fmax)                      # Newton's third law over 2.7 M edges
    # rigid rotation (proper, random): rows of `cell` are lattice vectors
    q, r = np.linalg.qr(np.random.default_rng(3).normal(size=(3, 3)))
    R = q * np.sign(np.diag(r))
    if np.linalg.det(R) < 0:
        R[:, 0] = -R[:, 0]
    ne1, e1, f1, v1 = evaluate(pos @ R.T, cell @ R.T)
    assert ne1 == ne0
    assert abs(e1 - e0) <= e_tol, (e1 - e0, e_tol)
    assert np.abs(f1 - f0 @ R.T).max() <= f_tol, (np.abs(f1 - f0 @ R.T).max(), f_tol)
    assert np.abs(v1 - R @ v0 @ R.T).max() <= v_tol, (np.abs(v1 - R @ v0 @ R.T).max(), v_tol)
    # translation (atoms leave the cell on one side: the list builder wraps them) and relabelling
    perm = np.random.default_rng(4).permutation(n)
    ne2, e2, f2, v2 = evaluate((pos + np.array([1.234, -7.5, 40.1]))[perm], cell, types[perm])
    assert ne2 == ne0
    assert abs(e2 - e0) <= e_tol, (e2 - e0, e_tol)
    assert np.abs(f2 - f0[perm]).max() <= f_tol, (np.abs(f2 - f0[perm]).max(), f_tol)
    assert np.abs(v2 - v0).max() <= v_tol, (np.abs(v2 - v0).max(), v_tol)
    # (no inversion test: the released SevenNet shapes are built with is_parity = False -- every irrep even, paths such as 1 x 1 -> 1
    # present --, i.e. they are SO(3)- but not O(3)-equivariant by construction; measured here, forces change by 1.5 % under inversion)


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
def test_forces_are_the_gradient_of_the_energy_at_the_benchmark_size(model):
    """the benchmark cells (97 336 / 54 872 atoms) through the defining property of a force (force_output.py:171-230: F = -dE/dr by
    autograd in the reference, a hand-scheduled reverse pass here): a central difference of the total energy along a collective
    displacement d (every atom along its own force, |d_i| <= 1) reproduces -sum_i F_i . d_i.  A wrong sign, a missing path of the
    reverse pass or a mis-folded periodic image shows at the 1e-1 level; truncation (h = 2e-3 A) and fp32 rounding at a few 1e-5."""
    from sevennet_amd.engine import HipForceEngine
    from sevennet_amd.neighbor_gpu import build_graph_gpu
    from sevennet_amd.synthetic import random_state_dict
    cfg, pos, cell, types, modal = _benchmark_workload(model)
    eng = HipForceEngine(cfg, random_state_dict(cfg, seed=0), device='cuda:0', modal=modal)
    ns = eng.spec.num_species if eng.needs_species_rows else 0

    def evaluate(p, c=cell):
        out = eng.compute(build_graph_gpu(types, p, c, cfg['cutoff'], device='cuda:0', num_species=ns))
        torch.cuda.synchronize()
        v = out['virial'].cpu().numpy()          # xx yy zz xy yz zx = -sum r (x) dE/dr (include/snet_hip.h)
        return float(out['energy'].cpu()), out['forces'].cpu().numpy().astype(np.float64), \
            np.array([[v[0], v[3], v[5]], [v[3], v[1], v[4]], [v[5], v[4], v[2]]])

    e0, f0, v0 = evaluate(pos)
    d = f0 / np.abs(f0).max()
    h = 2e-3
    ep = evaluate(pos + h * d)[0]
    em = evaluate(pos - h * d)[0]
    lhs, rhs = (ep - em) / (2 * h), -(f0 * d).sum()
    assert abs(rhs) > 1.0 and abs(ep - em) > 1e3 * 1e-8 * abs(e0)          # the signal is far above the rounding of the (fp64) energy sum
    # measured on an MI355X: relative mismatch 3.6e-5 (SevenNet-0: dE = -3.76 eV of -1 759 eV), 1.3e-5 (l3i5: -3.41 eV of 90 375 eV)
    assert abs(lhs - rhs) <= 2e-4 * abs(rhs), (lhs, rhs, ep - em, e0)
    # ... and the virial is the strain derivative: dE/d(eps) = -virial : eps for a symmetric strain of cell and positions
    # (force_output.py:218-230 forms the stress from it; pair_e3gnn.cpp:254-270 hands it to LAMMPS)
    eps = np.array([[0.3, 0.1, -0.2], [0.1, -0.5, 0.4], [-0.2, 0.4, 0.7]]) * 2e-4
    es = [evaluate(pos @ (np.eye(3) + sg * eps), cell @ (np.eye(3) + sg * eps))[0] for sg in (+1, -1)]
    lhs, rhs = (es[0] - es[1]) / 2, -(v0 * eps).sum()
    assert abs(lhs - rhs) <= 1e-3 * abs(rhs) + 1e-7 * abs(e0), (lhs, rhs, e0)


@pytest.mark.gpu
def test_eight_bricks_of_the_benchmark_cell_equal_the_single_graph():
    """What `bench.py --gpus 8` computes, checked at its own size on one GPU: the 97 336-atom cell of BASELINE config 3 cut into the
    8 bricks of the 2 x 2 x 2 processor grid (12 167 +- a few atoms each, ~5 500 ghost rows, interior / boundary split on), every rank
    on its own host thread and stream, ghost features and gradients exchanged by csrc/snet_halo.cpp's in-process transport inside
    snet_model_eval (the plan, the pack / accumulate kernels and the call sequence are those of the RCCL path) == the un-split
    evaluation of the whole cell: energy to 1e-8 relative, every force to 5e-6 of max|F| (reference analogue:
    tests/lammps_tests/test_lammps.py:540-578, serial against parallel pair style)."""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.parallel import LoopbackHub, NativeHalo, build_brick_graph
    from sevennet_amd.shapes import sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    world = 8
    cfg = sevennet_0_config()
    sd = random_state_dict(cfg, seed=0)
    pos, cell = diamond_cubic(5.431, (23,) * 3, 0.05, 2)
    types = np.zeros(len(pos), np.int64)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    ref = HipForceEngine(cfg, sd, device='cuda:0').compute(build_graph(types, ei, ev, device='cuda:0'))
    torch.cuda.synchronize()
    e_ref, fr = float(ref['energy'].cpu()), ref['forces'].cpu().numpy()
    del ref
    bricks = [build_brick_graph(pos, cell, types, cfg['cutoff'], world, r, neighbors=(ei, ev)) for r in range(world)]
    assert sum(b.n_local for b in bricks) == len(pos) and all(b.n_interior > 0 for b in bricks)
    hub = LoopbackHub(world)
    halos = [NativeHalo(hub.comm(r), bricks[r].send_lists, bricks[r].recv_counts, overlap=True) for r in range(world)]
    models = [NativeModel(cfg, sd) for _ in range(world)]

    def fn(r):
        b = bricks[r]
        g = build_graph(b.types, b.edge_index, b.edge_vec, n_local=b.n_local, n_interior=b.n_interior, device='cuda:0')
        models[r].set_halo(halos[r])
        out = models[r].compute(g)
        return float(out['energy'].cpu()), out['forces'].cpu().numpy()
    res = _run_ranks(world, fn, hub)
    F = np.zeros((len(types), 3), np.float32)
    for b, (_, f) in zip(bricks, res):
        F[b.global_ids[:b.n_local]] = f[:b.n_local]
    e_tot = sum(e for e, _ in res)
    # measured on an MI355X: the energies agree to the last printed digit, forces within 5.2e-8 eV/A at max|F| = 0.083 (6e-7 relative:
    # the ghost gradients are added in another order than the single graph's segment sum)
    assert abs(e_tot - e_ref) < 1e-8 * abs(e_ref), (e_tot, e_ref)
    assert np.abs(F - fr).max() <= 5e-6 * np.abs(fr).max(), (np.abs(F - fr).max(), np.abs(fr).max())


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
def test_gpu_neighbor_list_of_the_benchmark_cells_equals_the_host_list(model):
    """f1 at BASELINE's sizes, index-exact: the device cell list of each benchmark cell (97 336 atoms / 2.73 M edges; the 54 872-atom
    amorphous cell with its ragged degrees; 27 000 atoms at MF-ompa's 6-A cutoff) is the host KD-tree list (train/dataload.py:32-129
    restated) as a multiset of (center, source, periodic image), rows sorted by center, edge vectors to fp32 rounding."""
    cfg, pos, cell, _, _ = _benchmark_workload(model)
    g = _same_edges(pos, cell, [True] * 3, cfg['cutoff'])
    assert g.n_edges > 20 * len(pos)
