"""CPU: the `.snet` model-file writer and the loader's refusal paths (no GPU work)."""
import struct

import numpy as np
import pytest


def _write(tmp_path, cfg, seed=3):
    from sevennet_amd.model_file import write_model_file
    from sevennet_amd.synthetic import random_state_dict
    sd = random_state_dict(cfg, seed=seed)
    p = tmp_path / 'm.snet'
    write_model_file(str(p), cfg, sd)
    return p, sd


def test_model_file_header_and_size(tmp_path):
    from sevennet_amd.model_spec import ACT_CST, build_model_spec, sevennet_0_config
    cfg = sevennet_0_config()
    p, sd = _write(tmp_path, cfg)
    blob = p.read_bytes()
    assert blob[:8] == b'SNETMDL4'
    sp = build_model_spec(cfg)
    hdr = struct.unpack('<10i3f', blob[8:8 + 52])
    assert hdr[0] == sp.num_species and hdr[1] == 5 and hdr[2] == 2 and hdr[4] == 8 and hdr[9] == 128
    assert hdr[10] == pytest.approx(5.0) and hdr[12] == pytest.approx(ACT_CST['silu'])
    # every weight of the state dict is in the file exactly once (plus tables): size lower bound
    n_w = sum(int(np.prod(v.shape)) for k, v in sd.items() if 'weight' in k)
    assert len(blob) > 4 * n_w
    # first payload after the header: Bessel coefficients
    coeffs = np.frombuffer(blob[60:60 + 32], '<f4')
    assert np.allclose(coeffs, np.asarray(sd['edge_embedding.basis_function.coeffs'], np.float32))
    # deterministic
    p2, _ = _write(tmp_path / '..' / tmp_path.name, cfg)
    assert p2.read_bytes() == blob


def test_model_file_missing_weight_is_refused(tmp_path):
    from sevennet_amd.model_file import write_model_file
    from sevennet_amd.shapes import unit_test_config
    from sevennet_amd.synthetic import random_state_dict
    cfg = unit_test_config()
    sd = random_state_dict(cfg, seed=0)
    sd.pop(next(k for k in sd if 'self_interaction_2' in k))
    with pytest.raises(KeyError):
        write_model_file(str(tmp_path / 'x.snet'), cfg, sd)


def test_loader_rejects_non_model_files_without_gpu(tmp_path):
    import ctypes as C
    from sevennet_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    bad = tmp_path / 'bad.snet'
    bad.write_bytes(b'definitely not a model file')
    assert lib.snet_model_load(str(bad).encode(), C.byref(h)) != 0
    assert b'not a .snet' in lib.snet_last_error()
    assert lib.snet_model_load(str(tmp_path / 'nope').encode(), C.byref(h)) != 0
    assert b'cannot open' in lib.snet_last_error()
    assert not h.value


def test_deploy_cli_from_checkpoint(tmp_path):
    """reference-format checkpoint (config + model_state_dict) -> .snet with the deploy metadata"""
    import torch
    import json
    from helpers import GOLDEN
    from sevennet_amd.deploy import main
    d = np.load(f'{GOLDEN}/cp0_state.npz')
    cfg = json.loads(str(d['__config__']))
    sd = {k: d[k] for k in d.files if not k.startswith('__')}
    ck = tmp_path / 'cp.pth'
    torch.save({'config': cfg, 'model_state_dict': {k: torch.as_tensor(v) for k, v in sd.items()}}, ck)
    out = tmp_path / 'm.snet'
    assert main([str(ck), '-o', str(out)]) == 0
    blob = out.read_bytes()
    assert blob[:8] == b'SNETMDL4'
    tail = blob[-400:].decode('latin1')
    assert 'chemical_symbols_to_index=Hf O\n' in tail and 'model_type=E3_equivariant_model' in tail


def test_old_checkpoint_names_and_config_defaults(tmp_path):
    """checkpoints written before the 2024-04 module renaming / version <= 0.9 configs
    (reference: scripts/backward_compatibility.py:18-76)"""
    import torch
    from sevennet_amd.calculator import load_reference_checkpoint
    old = {'EdgeEmbedding.basis_function.coeffs': torch.arange(8.0),
           '0 convolution.denumerator': torch.tensor([28.0]),
           '0 self interaction 2.linear.weight': torch.ones(4),
           '0 equivariant gate.x': torch.zeros(1),
           '1 self connection intro.linear.weight': torch.ones(2),
           'reducing nn input to hidden.linear.weight': torch.ones(3),
           'rescale atomic energy.shift': torch.zeros(1),
           'onehot_to_feature_x.linear.weight': torch.ones(5)}
    cfg = {'version': '0.9.0', 'train_avg_num_neigh': True, 'cutoff': 5.0,
           'cutoff_function': {'cutoff_function_name': 'XPLOR', 'cutoff_on': 4.5, 'poly_cut_p_value': 6}}
    p = tmp_path / 'old.pth'
    torch.save({'config': cfg, 'model_state_dict': old}, p)
    c, sd = load_reference_checkpoint(str(p))
    assert set(sd) == {'edge_embedding.basis_function.coeffs', '0_convolution.denominator',
                       '0_self_interaction_2.linear.weight', '0_equivariant_gate.x',
                       '1_self_connection_intro.linear.weight', 'reduce_input_to_hidden.linear.weight',
                       'rescale_atomic_energy.shift', 'onehot_to_feature_x.linear.weight'}
    assert c['train_denominator'] is True and c['conv_denominator'] == 0.0 and c['_normalize_sph'] is False
    assert 'poly_cut_p_value' not in c['cutoff_function']
    torch.save({'config': {'cutoff': 5.0}, 'model_state_dict': old}, p)
    with pytest.raises(ValueError, match='version'):
        load_reference_checkpoint(str(p))
    torch.save({'config': dict(cfg, optimize_by_reduce=False), 'model_state_dict': old}, p)
    with pytest.raises(ValueError, match='optimize_by_reduce'):
        load_reference_checkpoint(str(p))


def test_old_checkpoint_w3j_sign_fix(tmp_path):
    """sort_old_convolution's Wigner-3j sign fix (scripts/backward_compatibility.py:119-137): a pre-0.11
    checkpoint whose stored `_w3j_l1_l2_l3` buffer is the negative of today's tensor gets the radial-weight
    columns of the paths that read it negated -- so a checkpoint with (buffer, columns) both negated loads to
    the SAME weights as the original; `0.11.0.dev0` counts as old (:176-183); a buffer that is neither
    +C nor -C is refused."""
    import json
    import torch
    from helpers import GOLDEN
    from sevennet_amd.calculator import W3J_KEY, fix_old_convolution_signs, load_reference_checkpoint
    from sevennet_amd.model_spec import build_model_spec, old_convolution_order
    assert old_convolution_order('0.10.0') and old_convolution_order('0.9.3') and old_convolution_order('0.11.0.dev0')
    assert not old_convolution_order('0.11.0') and not old_convolution_order('0.11.0.dev1') and not old_convolution_order('0.12.1')
    d = np.load(f'{GOLDEN}/cp0_state.npz')
    w3 = np.load(f'{GOLDEN}/w3j_cp0.npz')
    cfg = json.loads(str(d['__config__']))
    assert old_convolution_order(cfg['version'])           # cp_0.pth is 0.10.0
    spec = build_model_spec(cfg)
    sd = {k: d[k] for k in d.files if not k.startswith('__')}
    for ls in spec.layers:                                  # the buffers e3nn stored, as in the real file
        for key in w3.files:
            l1, l2, l3 = key.split('_')
            sd[W3J_KEY.format(t=ls.t, l1=l1, l2=l2, l3=l3)] = w3[key]
    base = fix_old_convolution_signs(cfg, sd)
    assert not any('_w3j_' in k for k in base)
    for k in base:                                          # stored tensors == today's: nothing changes
        assert np.array_equal(base[k], sd[k]), k
    # flip one tensor of layer 1 together with the columns of every path that reads it
    flipped = {k: np.array(v, copy=True) for k, v in sd.items()}
    ls = spec.layers[1]
    key = W3J_KEY.format(t=ls.t, l1=1, l2=1, l3=1)
    flipped[key] *= -1
    ww = f'{ls.t}_convolution.weight_nn.layer2.weight'
    hit = [p for p in ls.conv.paths if (p.l1, p.l2, p.l3) == (1, 1, 1)]
    assert len(hit) >= 1                                    # every path that reads the buffer (O(3) models can have several)
    for p in hit:
        flipped[ww][:, p.w_off:p.w_off + p.mul] *= -1
    assert not np.array_equal(flipped[ww], sd[ww])
    fixed = fix_old_convolution_signs(cfg, flipped)
    for k in base:
        assert np.array_equal(fixed[k], base[k]), k         # bit-identical weights -> bit-identical forces
    # through the file loader, and the version gate
    ck = tmp_path / 'flipped.pth'
    torch.save({'config': cfg, 'model_state_dict': {k: torch.as_tensor(v) for k, v in flipped.items()}}, ck)
    _, sd2 = load_reference_checkpoint(str(ck))
    assert np.array_equal(sd2[ww], base[ww])
    torch.save({'config': dict(cfg, version='0.11.0.dev0'), 'model_state_dict': {k: torch.as_tensor(v) for k, v in flipped.items()}}, ck)
    _, sd3 = load_reference_checkpoint(str(ck))
    assert np.array_equal(sd3[ww], base[ww])
    torch.save({'config': dict(cfg, version='0.11.0'), 'model_state_dict': {k: torch.as_tensor(v) for k, v in flipped.items()}}, ck)
    _, sd4 = load_reference_checkpoint(str(ck))
    assert np.array_equal(sd4[ww], flipped[ww])             # current order: buffers are not consulted
    bad = dict(flipped)
    bad[key] = bad[key] * 0.5
    with pytest.raises(ValueError, match='neither'):
        fix_old_convolution_signs(cfg, bad)


def test_species_wise_rescale_checkpoint_with_scalar_config():
    """shift / scale shapes come from the checkpoint tensors: a species-wise-rescale checkpoint whose config
    still holds scalars loads; sizes that match neither 1 nor n_species are refused; unsupported
    radial basis / activation / cutoff names raise explicit errors"""
    from sevennet_amd.model_spec import build_model_spec
    from sevennet_amd.shapes import unit_test_config
    cfg = unit_test_config(shift=-1.0, scale=2.0)
    sp = build_model_spec(cfg)
    ns = sp.num_species
    sd = {'rescale_atomic_energy.scale': np.arange(1, ns + 1, dtype=np.float32), 'rescale_atomic_energy.shift': np.array([0.5], np.float32)}
    sc, sh = sp.rescale_vectors(sd, -1)
    assert sc.tolist() == list(range(1, ns + 1)) and sh.tolist() == [0.5] * ns
    with pytest.raises(ValueError, match='entries'):
        sp.rescale_vectors({'rescale_atomic_energy.scale': np.ones(ns + 1), 'rescale_atomic_energy.shift': np.ones(1)}, -1)
    with pytest.raises(NotImplementedError, match='radial basis'):
        build_model_spec(unit_test_config(radial_basis={'radial_basis_name': 'gaussian'}))
    with pytest.raises(ValueError, match='act_radial'):
        build_model_spec(unit_test_config(act_radial='gelu'))     # not in sevenn/_const.py:33-47 either
    # every activation, o3.Linear biases and the FCN readout of the reference are model options now (not refusals)
    from sevennet_amd.model_spec import ACT_CST, ACT_ID
    from oracle.e3 import normalize2mom_const
    assert sorted(ACT_ID) == sorted(['relu', 'silu', 'tanh', 'abs', 'ssp', 'sigmoid', 'elu'])
    for name, cst in ACT_CST.items():      # e3nn's normalize2mom: seeded 1e6-sample second moment (the oracle re-derives it)
        assert abs({'silu': 1.6791767923989418, 'tanh': 1.5937334472592695}.get(name, normalize2mom_const(name)) - cst) < 1e-12 * cst
    sp = build_model_spec(unit_test_config(use_bias_in_linear=True, readout_as_fcn=True, act_radial='relu'))
    shapes = sp.param_shapes()
    assert shapes['readout_FCN.fcn.layer0.weight'] == (4, 30) and shapes['readout_FCN.fcn.layer2.weight'] == (30, 1)
    assert shapes['onehot_to_feature_x.linear.bias'] == (4,) and 'reduce_input_to_hidden.linear.weight' not in shapes
    assert '0_self_connection_intro.fc_tensor_product.bias' not in shapes     # the self-connection carries no bias
    with pytest.raises(NotImplementedError, match='cutoff function'):
        build_model_spec(unit_test_config(cutoff_function={'cutoff_function_name': 'cosine'}))


def test_paths_nothing_reads_are_not_evaluated():
    """SevenNet-MF-ompa's third interaction layer computes 0o / 1e / 2o / 3e blocks that its SI2 (an o3.Linear: unmatched
    input irreps are ignored) never reads: the engine's shape keeps the 34 paths that are read, the checkpoint layout
    (parameter shapes, old-checkpoint sign fix) keeps all 68"""
    from sevennet_amd.model_spec import (build_model_spec, sevennet_0_config, sevennet_l3i5_config,
                                         sevennet_mf_ompa_config)
    cfg = sevennet_mf_ompa_config()
    sp = build_model_spec(cfg)
    ls = sp.layers[3]
    assert (len(ls.conv_full.paths), len(ls.conv.paths)) == (68, 34)
    assert (ls.conv_full.weight_numel, ls.conv.weight_numel, len(ls.w_cols)) == (3520, 1760, 1760)
    assert sp.param_shapes()['3_convolution.weight_nn.layer2.weight'] == (64, 3520)
    read = {b.in_off for b in ls.si2.blocks}
    offs = ls.si2.irreps_in.offsets()
    blocks = [(o, o + m * (2 * l + 1)) for o, (m, l, _) in zip(offs, ls.si2.irreps_in)]
    block_of = lambda off: next(o for o, e in blocks if o <= off < e)   # noqa: E731
    kept = {(p.i_x, p.i_sh, p.out_off, p.out_ch) for p in ls.conv.paths}
    col = 0
    for p in ls.conv_full.paths:
        live = block_of(p.out_off) in read
        assert ((p.i_x, p.i_sh, p.out_off, p.out_ch) in kept) == live
        if live:   # weight columns of the kept paths, in order
            assert list(ls.w_cols[col:col + p.mul]) == list(range(p.w_off, p.w_off + p.mul))
            col += p.mul
    sd = {f'3_convolution.weight_nn.layer{i}.weight': np.arange(a * b, dtype=np.float64).reshape(a, b)
          for i, (a, b) in enumerate(zip(ls.mlp_dims_full[:-1], ls.mlp_dims_full[1:]))}
    w = ls.radial_weights(sd)
    assert w[2].shape == (64, 1760) and np.array_equal(w[2], sd['3_convolution.weight_nn.layer2.weight'][:, ls.w_cols])
    # every other layer of the three benchmark shapes reads all it computes; the switch restores the reference's list
    for c in (sevennet_0_config(), sevennet_l3i5_config(), cfg):
        for l2 in build_model_spec(c).layers:
            assert (l2.w_cols is None) == (len(l2.conv.paths) == len(l2.conv_full.paths))
            assert l2.w_cols is None or (c is cfg and l2.t == 3)
    full = build_model_spec(dict(cfg, _prune_unread_paths=False)).layers[3]
    assert len(full.conv.paths) == 68 and full.w_cols is None


def test_model_file_carries_the_fcn_readout(tmp_path):
    """format v4: `readout_as_fcn` models are written (readout kind 1: widths, activation, W_i / sqrt(fan_in)); the tail of the file
    before the metadata is exactly that block"""
    from sevennet_amd.model_spec import ACT_CST, ACT_ID
    from sevennet_amd.shapes import unit_test_config
    cfg = unit_test_config(readout_as_fcn=True, readout_fcn_hidden_neurons=[16, 8], readout_fcn_activation='elu')
    p, sd = _write(tmp_path, cfg)
    blob = p.read_bytes()
    assert blob[:8] == b'SNETMDL4'
    meta_len = blob.rfind(b'chemical_symbols_to_index=')
    n_meta = struct.unpack('<i', blob[meta_len - 4:meta_len])[0]
    assert meta_len + n_meta == len(blob)
    dims = [4, 16, 8, 1]
    n_w = sum(dims[i] * dims[i + 1] for i in range(3))
    blk = blob[meta_len - 4 - 4 * n_w - 4 * (1 + 4 + 1) - 4:meta_len - 4]
    head = struct.unpack('<6i', blk[:24])
    assert list(head) == [3, 4, 16, 8, 1, ACT_ID['elu']]
    assert struct.unpack('<f', blk[24:28])[0] == pytest.approx(ACT_CST['elu'])
    w = np.frombuffer(blk[28:], '<f4')
    off = 0
    for i in range(3):
        ref = np.asarray(sd[f'readout_FCN.fcn.layer{i}.weight'], np.float64).reshape(dims[i], dims[i + 1]) / np.sqrt(dims[i])
        assert np.allclose(w[off:off + ref.size], ref.astype(np.float32).ravel())
        off += ref.size
