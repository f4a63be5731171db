"""GPU: the native whole-model sequencer (snet_model_*, what a C++ host such as the LAMMPS pair
styles calls) against the Python-hosted engine, the reference's TorchScript outputs and the oracle."""
import os
import threading

import numpy as np
import pytest
import torch

from helpers import load_ts_golden, oracle_model, synthetic_system

pytestmark = pytest.mark.gpu

F_TOL = 1e-4  # eV/A, BASELINE.json north_star tolerance


def _both(cfg, sd, types, ei, ev):
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.native_model import NativeModel
    eng = HipForceEngine(cfg, sd, device='cuda:0')
    nat = NativeModel(cfg, sd, device='cuda:0')
    g = build_graph(types, ei, ev, device='cuda:0', num_species=eng.spec.num_species)
    a = eng.compute(g, want_atomic_virial=True)
    b = nat.compute(g, want_atomic_virial=True)
    torch.cuda.synchronize()
    return eng, nat, g, a, b


def _assert_identical(a, b):
    """same kernels in the same order on the same inputs: results must agree bit for bit"""
    for k in ('energy', 'atomic_energy', 'dE_dr', 'forces', 'virial', 'atomic_virial'):
        assert torch.equal(a[k], b[k]), f'{k}: native sequencer differs from the Python-hosted engine'


@pytest.mark.parametrize('name', ['hfo2_12', 'hfo2_96'])
def test_native_model_vs_reference_torchscript_outputs(name):
    d, cfg, sd = load_ts_golden(name)
    eng, nat, g, a, b = _both(cfg, sd, d['types'], d['edge_index'], d['out_edge_vec'])
    _assert_identical(a, b)
    n = len(d['types'])
    assert abs(float(b['energy'].cpu()) - float(d['out_energy'])) / n < 1e-5
    assert np.abs(b['forces'].cpu().numpy() - d['out_forces']).max() < F_TOL
    assert np.abs(b['dE_dr'].cpu().numpy() - d['par_dE_dr']).max() < F_TOL
    assert nat.cutoff == pytest.approx(float(cfg['cutoff'])) and nat.n_layers == len(eng.layers)
    assert nat.comm_dims == [L.spec.si1.dim_out for L in eng.layers]
    assert nat.meta('chemical_symbols_to_index') == 'Hf O' and float(nat.meta('cutoff')) == 4.0
    assert nat.meta('model_type') == 'E3_equivariant_model'
    with pytest.raises(RuntimeError, match='no such key'):
        nat.meta('nope')


CASES = {
    'unit_o3_l2': dict(cfg='unit', over={}, cutoff=4.0, nsp=4),                    # per-species FCTP self-connection
    'unit_so3_l2_linear': dict(cfg='unit', over={'is_parity': False, 'self_connection_type': 'linear'}, cutoff=4.0, nsp=4),
    'mini_7net0': dict(cfg='mini', over={}, cutoff=5.0, nsp=2),
    # o3.Linear biases (folded into the .snet linears' constant rows, the species tables and the readout vector) + other activations
    'unit_bias_ssp_abs': dict(cfg='unit', over={'use_bias_in_linear': True, 'act_radial': 'ssp', 'act_scalar': {'e': 'ssp', 'o': 'abs'},
                                                'act_gate': {'e': 'ssp', 'o': 'abs'}}, cutoff=4.0, nsp=4),
    # `readout_as_fcn` in the .snet file (format v4): the native sequencer runs engine.py's launches in engine.py's order
    'unit_bias_fcn_elu': dict(cfg='unit', over={'use_bias_in_linear': True, 'readout_as_fcn': True, 'readout_fcn_activation': 'elu'},
                              cutoff=4.0, nsp=4),
    'unit_fcn_one_hidden_relu': dict(cfg='unit', over={'readout_as_fcn': True, 'readout_fcn_hidden_neurons': [16]}, cutoff=4.0, nsp=4),
}


@pytest.mark.parametrize('case', list(CASES))
def test_native_model_vs_engine_and_oracle(case):
    from sevennet_amd.shapes import mini_sevennet_0_config, unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    c = CASES[case]
    cfg = unit_test_config(**c['over']) if c['cfg'] == 'unit' else mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=7)
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.08, seed=11, cutoff=c['cutoff'], n_species=c['nsp'])
    eng, nat, g, a, b = _both(cfg, sd, types, ei, ev)
    _assert_identical(a, b)
    ref = oracle_model(cfg, sd).forward(types, ei, ev)
    scale = max(1.0, ref['forces'].abs().max().item())
    assert np.abs(b['forces'].cpu().numpy() - ref['forces'].numpy()).max() < F_TOL * scale
    # second evaluation out of the same arena, smaller then larger system: no stale state
    types2, _, _, ei2, ev2 = synthetic_system((1, 1, 1), sigma=0.05, seed=2, cutoff=c['cutoff'], n_species=c['nsp'])
    from sevennet_amd.engine import build_graph
    g2 = build_graph(types2, ei2, ev2, device='cuda:0', num_species=eng.spec.num_species)
    _assert_identical(eng.compute(g2, want_atomic_virial=True), nat.compute(g2, want_atomic_virial=True))
    _assert_identical(a, nat.compute(g, want_atomic_virial=True))


def test_native_model_isolated_atom_and_errors(tmp_path):
    from sevennet_amd import _lib
    from sevennet_amd.engine import build_graph
    from sevennet_amd.model_file import write_model_file
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = unit_test_config()
    sd = random_state_dict(cfg, seed=1)
    path = tmp_path / 'unit.snet'
    write_model_file(str(path), cfg, sd)
    nat = NativeModel(path)
    iso = nat.compute(build_graph(np.array([2]), np.zeros((2, 0), np.int64), np.zeros((0, 3)), device='cuda:0', num_species=4))
    ref = oracle_model(cfg, sd).forward(np.array([2]), np.zeros((2, 0), np.int64), np.zeros((0, 3)))
    assert abs(float(iso['energy'].cpu()) - float(ref['energy'])) < 1e-4
    assert iso['forces'].abs().max().item() == 0.0
    # truncated file, wrong magic
    blob = path.read_bytes()
    (tmp_path / 'short.snet').write_bytes(blob[:len(blob) // 2])
    with pytest.raises(RuntimeError, match='malformed'):
        NativeModel(tmp_path / 'short.snet')
    (tmp_path / 'bad.snet').write_bytes(b'NOTSNET!' + blob[8:])
    with pytest.raises(RuntimeError, match='not a .snet'):
        NativeModel(tmp_path / 'bad.snet')
    with pytest.raises(RuntimeError, match='cannot open'):
        NativeModel(tmp_path / 'missing.snet')
    # ghost atoms without halo hooks are refused
    g = build_graph(np.array([0, 1, 2]), np.array([[0], [2]]), np.array([[1.0, 0.5, 0.2]]), n_local=2, device='cuda:0',
                    num_species=4)
    with pytest.raises(RuntimeError, match='halo'):
        nat.compute(g)
    assert _lib.load().snet_last_error()


def test_native_model_bricks_equal_single_graph_on_one_gpu():
    """2 bricks, each a native model on its own host thread, ghost exchange through the halo
    callbacks == the un-split evaluation (reference analogue: tests/lammps_tests/test_lammps.py:540-578)"""
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.parallel import InProcessHaloGroup, build_brick_graph
    from sevennet_amd.shapes import mini_sevennet_0_config
    from sevennet_amd.synthetic import random_state_dict
    world = 2
    cfg = mini_sevennet_0_config()
    sd = random_state_dict(cfg, seed=9)
    types, pos, cell, ei, ev = synthetic_system((4, 4, 4), sigma=0.06, seed=4, cutoff=5.0, n_species=2)
    ref = HipForceEngine(cfg, sd, device='cuda:0').compute(build_graph(types, ei, ev, device='cuda:0'))
    torch.cuda.synchronize()
    bricks = [build_brick_graph(pos, cell, types, 5.0, world, r, neighbors=(ei, ev)) for r in range(world)]
    grp = InProcessHaloGroup(bricks, 'cuda:0')
    models = [NativeModel(cfg, sd) for _ in range(world)]
    results, errors = [None] * world, []

    def run(r):
        try:
            b = bricks[r]
            g = build_graph(b.types, b.edge_index, b.edge_vec, n_local=b.n_local, n_interior=b.n_interior, device='cuda:0')
            models[r].set_halo(grp.members[r])
            results[r] = models[r].compute(g)
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
    e_tot = 0.0
    for b, r in zip(bricks, results):
        F[b.global_ids[:b.n_local]] = r['forces'].cpu().numpy()
        e_tot += float(r['energy'].cpu())
    assert abs(e_tot - float(ref['energy'].cpu())) < 2e-6 * abs(float(ref['energy'].cpu()))
    fr = ref['forces'].cpu().numpy()
    assert np.abs(F - fr).max() <= max(1e-8, 2e-5 * np.abs(fr).max())


def test_topology_cache_no_sync_on_repeated_graph():
    """the native sequencer keeps the topology-only work (tile list -- whose construction synchronises the
    stream --, edges grouped by source, species row lists) per graph: evaluations after the first issue NO stream
    synchronisation inside snet_model_eval, give bit-identical results, and a new graph (other index arrays) rebuilds it"""
    from sevennet_amd.engine import build_graph
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.native_model import NativeModel
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config(num_species=2)
    sd = random_state_dict(cfg, seed=2)
    nat = NativeModel(cfg, sd, device='cuda:0')
    types, pos, cell, ei, ev = synthetic_system((2, 2, 2), sigma=0.05, seed=1, cutoff=5.0, n_species=2)
    g = build_graph(types, ei, ev, device='cuda:0', num_species=2)
    s0 = nat.eval_syncs()
    a = nat.compute(g)
    s1 = nat.eval_syncs()
    assert s1 > s0                      # first evaluation: the tile list is built (one readback)
    g.edge_vec.mul_(1.0)                # new positions would only change edge_vec: same topology
    b = nat.compute(g)
    c = nat.compute(g)
    assert nat.eval_syncs() == s1       # no synchronisation inside the next evaluations
    for k in ('energy', 'forces', 'dE_dr'):
        assert torch.equal(a[k], b[k]) and torch.equal(b[k], c[k])
    types2, pos2, cell2, ei2, ev2 = synthetic_system((2, 2, 3), sigma=0.05, seed=2, cutoff=5.0, n_species=2)
    g2 = build_graph(types2, ei2, ev2, device='cuda:0', num_species=2)
    d = nat.compute(g2)
    assert nat.eval_syncs() > s1
    ref = oracle_model(cfg, sd).forward(types2, ei2, ev2)
    assert np.abs(d['forces'].cpu().numpy() - ref['forces'].numpy()).max() < 2e-5 * float(ref['forces'].abs().max())
    e = nat.compute(g)                  # back to the first graph: rebuilt again, same answer
    assert torch.equal(e['forces'], a['forces'])
