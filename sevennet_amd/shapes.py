"""Model configurations whose tensor-product shapes are compiled into libsnet_hip.so.

The engine's convolution kernels are generated per shape (codegen.py) and built
ahead of time by `sevennet_amd.build`.  To run a model with other irreps, add
its config here (or call `sevennet_amd.build.build(extra_configs=[...])`).
"""
from __future__ import annotations

from typing import Dict, List

from .model_spec import (ConvSpec, build_model_spec, sevennet_0_config,
                         sevennet_l3i5_config, sevennet_mf_ompa_config, transposed_scalar_conv)

# the reference's deployed example model (sevenn 0.8.6), used by the golden fixtures
TS_EXAMPLE_CONFIG = dict(
    cutoff=4.0, channel=4, lmax=1, is_parity=True, num_convolution_layer=4,
    self_connection_type='nequip',
    cutoff_function={'cutoff_function_name': 'poly_cut', 'poly_cut_p_value': 6},
    radial_basis={'radial_basis_name': 'bessel', 'bessel_basis_num': 8},
    weight_nn_hidden_neurons=[64, 64], act_radial='silu', _normalize_sph=False,
    version='0.8.6', _number_of_species=2, _legacy_v08=True)


def unit_test_config(**over) -> dict:
    """tests/unit_tests/test_model.py:53-86 of the reference (channel 4, lmax 2,
    O(3), 3 layers) -- same shapes as tests/data/checkpoints/cp_0.pth."""
    cfg = dict(cutoff=4.0, channel=4, lmax=2, is_parity=True, num_convolution_layer=3,
               weight_nn_hidden_neurons=[64, 64], conv_denominator=30.0,
               self_connection_type='nequip', shift=-10.0, scale=10.0,
               _normalize_sph=True, _number_of_species=4, version='0.12.0')
    cfg.update(over)
    return cfg


def mini_sevennet_0_config(num_species: int = 2) -> dict:
    """SevenNet-0's structure (5 layers, SO(3), XPLOR, linear self-connection)
    at 1/8 width -- a CPU-oracle-sized stand-in for parity tests."""
    cfg = sevennet_0_config(num_species)
    cfg.update(channel=16, irreps_manual=['16x0e'] + ['16x0e+8x1e+4x2e'] * 4 + ['16x0e'],
               conv_denominator=20.0)
    return cfg


def aot_configs() -> Dict[str, dict]:
    return {
        'ts_example': TS_EXAMPLE_CONFIG,
        'unit_o3_l2': unit_test_config(),
        'unit_o3_l3': unit_test_config(lmax=3),
        # instruction order of checkpoints older than 0.11 (tests/data/checkpoints/cp_0.pth is 0.10.0)
        'unit_o3_l2_v010': unit_test_config(version='0.10.0'),
        'unit_so3_l2_linear': unit_test_config(is_parity=False, self_connection_type='linear'),
        'mini_7net0': mini_sevennet_0_config(),
        'sevennet_0': sevennet_0_config(),
        'sevennet_l3i5': sevennet_l3i5_config(),
        'sevennet_mf_ompa': sevennet_mf_ompa_config(),
    }


def aot_conv_specs(extra_configs: List[dict] = ()) -> Dict[str, ConvSpec]:
    specs: Dict[str, ConvSpec] = {}
    for cfg in list(aot_configs().values()) + list(extra_configs):
        for ls in build_model_spec(cfg).layers:
            specs.setdefault(ls.conv.tag, ls.conv)
            # scalar-output (last) layers: the source-row gradient runs as a forward convolution of the transposed product
            tr = transposed_scalar_conv(ls.conv)
            if tr is not None and all(mul % 16 == 0 for mul, _, _ in ls.conv.irreps_x):
                specs.setdefault(tr[0].tag, tr[0])
    return specs
