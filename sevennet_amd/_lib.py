"""ctypes binding of libsnet_hip.so (the C ABI declared in include/snet_hip.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the
shared object is missing this module raises immediately (build it with
`python -m sevennet_amd.build`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 3   # == SNET_ABI_VERSION of include/snet_hip.h (tests/test_abi_cpu.py keeps the two in step)
LIB_PATH = os.environ.get('SNET_HIP_LIB') or os.path.join(_HERE, 'libsnet_hip.so')  # env: kernel experiments

c_f32p = C.c_void_p   # device float*
c_i32p = C.c_void_p   # device int32*
c_f64p = C.c_void_p
c_stream = C.c_void_p


class EdgeParams(C.Structure):
    _fields_ = [('cutoff', C.c_float), ('n_basis', C.c_int32), ('cutoff_kind', C.c_int32),
                ('poly_p', C.c_int32), ('cutoff_on', C.c_float), ('lmax', C.c_int32),
                ('normalize', C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [('B', C.c_void_p), ('B_split', C.c_void_p), ('a_off', C.c_int64), ('c_off', C.c_int64), ('d', C.c_int32), ('K', C.c_int32),
                ('N', C.c_int32), ('accumulate', C.c_int32)]


class GateSeg(C.Structure):
    _fields_ = [('kind', C.c_int32), ('in_off', C.c_int32), ('out_off', C.c_int32), ('mul', C.c_int32),
                ('l', C.c_int32), ('gate_off', C.c_int32), ('act', C.c_int32), ('cst', C.c_float)]


# name -> (restype, argtypes); mirrors include/snet_hip.h one to one
SIGNATURES = {
    'snet_abi_version': (C.c_int, []),
    'snet_last_error': (C.c_char_p, []),
    'snet_conv_num_shapes': (C.c_int, []),
    'snet_conv_shape_tag': (C.c_char_p, [C.c_int]),
    'snet_edge_embed_fwd': (C.c_int, [C.POINTER(EdgeParams), C.POINTER(C.c_float), c_f32p, C.c_int64, c_f32p,
                                      c_f32p, c_f32p, c_stream]),
    'snet_edge_embed_bwd': (C.c_int, [C.POINTER(EdgeParams), C.POINTER(C.c_float), c_f32p, C.c_int64, c_f32p,
                                      c_f32p, c_f32p, C.c_int32, c_stream]),
    'snet_radial_mlp_plan_create': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float),
                                              C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_float,
                                              C.c_int32, C.POINTER(C.c_void_p)]),
    'snet_radial_mlp_plan_destroy': (None, [C.c_void_p]),
    'snet_radial_mlp_fwd': (C.c_int, [C.c_void_p, c_f32p, C.c_int64, c_f32p, c_stream]),
    'snet_radial_mlp_bwd': (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int64, c_f32p, c_stream]),
    'snet_conv_bwd_edge_vec': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int64,
                                         C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    'snet_gemm': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                            C.c_int64, C.c_int64, C.c_int64, c_i32p, C.c_int32, c_stream]),
    'snet_gemm_grouped': (C.c_int, [C.POINTER(GemmDesc), C.c_int32, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int64,
                                    c_i32p, c_stream]),
    'snet_gemm_split_size': (C.c_int64, [C.c_int32, C.c_int32]),
    'snet_gemm_split_pack': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'snet_act_fwd': (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, c_stream]),
    'snet_act_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, c_stream]),
    'snet_conv_plan_create': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'snet_conv_plan_destroy': (None, [C.c_void_p]),
    'snet_conv_register_library': (C.c_int, [C.c_char_p]),
    'snet_conv_plan_dims': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'snet_conv_fwd': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int64, C.c_float,
                                c_f32p, c_stream]),
    'snet_radial_mlp_hidden_fwd': (C.c_int, [C.c_void_p, c_f32p, C.c_int64, c_f32p, c_stream]),
    'snet_radial_mlp_hidden_fwd_layers': (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, c_f32p, C.c_int64, C.POINTER(C.c_void_p), c_stream]),
    'snet_radial_mlp_hidden_bwd': (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int64, c_f32p, c_stream]),
    'snet_conv_fused_available': (C.c_int, [C.c_void_p]),
    'snet_conv_plan_transposed': (C.c_int, [C.c_void_p, C.c_char_p, c_f32p, c_i32p, C.c_int32, C.POINTER(C.c_int32)]),
    'snet_edges_by_source': (C.c_int, [c_i32p, C.c_int64, c_i32p, c_i32p, C.c_int64, c_i32p, c_i32p, c_stream]),
    'snet_fused_plan_create': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    'snet_fused_plan_destroy': (None, [C.c_void_p]),
    'snet_edge_tiles': (C.c_int, [c_i32p, C.c_int64, c_i32p, c_i32p, C.c_int64, C.POINTER(C.c_int64), c_stream]),
    'snet_conv_fwd_fused': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int64, C.c_float,
                                      c_f32p, c_stream]),
    'snet_conv_bwd_fused': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i32p,
                                      c_i32p, C.c_int64, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                      c_stream]),
    'snet_conv_bwd_fused_sh': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i32p,
                                         c_i32p, C.c_int64, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                         c_stream]),
    'snet_fused_plan_has_mlp_tail': (C.c_int, [C.c_void_p]),
    'snet_fused_plan_gxe_chunks': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]),
    'snet_segment_sum_rows_chunked': (C.c_int, [c_f32p, c_i32p, c_i32p, C.c_int64, C.c_int32, c_i32p, c_f32p, c_stream]),
    'snet_edge_vectors': (C.c_int, [c_f64p, c_i32p, c_i32p, c_f64p, C.c_int64, c_f32p, c_stream]),
    'snet_row_absmax': (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, c_stream]),
    'snet_row_absmax_multi': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.c_int32, c_stream]),
    'snet_row_norm2': (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_float, c_f32p, c_stream]),
    'snet_conv_bwd_edge': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int64, C.c_float,
                                     c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    'snet_segment_sum_rows': (C.c_int, [c_f32p, c_i32p, c_i32p, C.c_int64, C.c_int32, c_f32p, c_stream]),
    'snet_conv_bwd_node': (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_int64, C.c_float,
                                     c_f32p, c_f32p, c_stream]),
    'snet_gate_fwd': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(GateSeg), C.c_int32,
                                c_stream]),
    'snet_gate_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(GateSeg),
                                C.c_int32, c_stream]),
    'snet_gate_bwd_norm': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(GateSeg),
                                     C.c_int32, C.c_float, c_f32p, c_stream]),
    'snet_embed_rows': (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int64, C.c_int32, c_stream]),
    'snet_add_row_bias': (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_stream]),
    'snet_add_inplace': (C.c_int, [c_f32p, c_f32p, C.c_int64, c_stream]),
    'snet_permute_cols': (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int64, C.c_int32, c_stream]),
    'snet_rescale_reduce': (C.c_int, [c_f32p, c_i32p, c_f32p, c_f32p, C.c_int32, C.c_int64, c_f32p, c_f64p,
                                      c_stream]),
    'snet_readout_energy': (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f64p, C.c_double, c_i32p, c_f32p, c_f32p, C.c_int32,
                                      c_f32p, c_f64p, c_stream]),
    'snet_readout_grad': (C.c_int, [c_f64p, C.c_int32, c_i32p, c_f32p, C.c_int32, C.c_int64, c_f32p, c_stream]),
    'snet_edge_force': (C.c_int, [c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int64, C.c_int64, c_f32p, c_f32p,
                                  c_f64p, c_stream]),
    'snet_nl_grid': (C.c_int, [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    'snet_nl_bin': (C.c_int, [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p, C.c_int64,
                              C.c_void_p, c_i32p, c_i32p, c_stream]),
    'snet_nl_count': (C.c_int, [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p, c_i32p,
                                c_i32p, c_i32p, C.c_int64, c_i32p, c_stream]),
    'snet_nl_fill': (C.c_int, [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p, c_i32p,
                               c_i32p, c_i32p, c_i32p, C.c_int64, c_i32p, c_i32p, c_i32p, c_f32p, c_i32p, c_stream]),
    'snet_gather_rows': (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int64, C.c_int32, c_stream]),
    'snet_scatter_add_rows': (C.c_int, [c_f32p, c_i32p, c_f32p, C.c_int64, C.c_int32, c_stream]),
    'snet_model_load': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'snet_model_load_memory': (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    'snet_model_destroy': (None, [C.c_void_p]),
    'snet_model_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.c_int32]),
    'snet_model_meta': (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32]),
    'snet_edge_tiles_packed': (C.c_int, [c_i32p, C.c_int64, C.c_int64, c_i32p, c_i32p, C.c_int64, C.POINTER(C.c_int64), c_stream]),
    'snet_fused_plan_tile_mode': (C.c_int, [C.c_void_p]),
    'snet_i32_shift': (C.c_int, [c_i32p, C.c_int32, c_i32p, C.c_int64, c_stream]),
    'snet_model_set_interior': (C.c_int, [C.c_void_p, C.c_int64]),
    'snet_model_set_topology_cache': (C.c_int, [C.c_void_p, C.c_int32]),
    'snet_model_topology_changed': (C.c_int, [C.c_void_p]),
    'snet_model_eval_syncs': (C.c_int64, [C.c_void_p]),
    'snet_model_set_halo': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    'snet_model_eval': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p,
                                  c_i32p, c_f32p, c_i32p, c_i32p, C.c_int64, C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, c_stream]),
    'snet_rccl_unique_id': (C.c_int, [C.c_void_p]),
    'snet_rccl_comm_create': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'snet_rccl_comm_destroy': (None, [C.c_void_p]),
    'snet_rccl_available': (C.c_int, []),
    'snet_rccl_comm_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'snet_rccl_allreduce_sum_f64': (C.c_int, [C.c_void_p, c_f64p, C.c_int64, c_stream]),
    'snet_loopback_hub_create': (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    'snet_loopback_hub_abort': (None, [C.c_void_p]),
    'snet_loopback_hub_destroy': (None, [C.c_void_p]),
    'snet_loopback_comm_create': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    'snet_halo_create': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    'snet_halo_destroy': (None, [C.c_void_p]),
    'snet_halo_ghost_rows': (C.c_int64, [C.c_void_p]),
    'snet_halo_send_rows': (C.c_int64, [C.c_void_p]),
    'snet_halo_forward': (C.c_int, [C.c_void_p, c_f32p, C.c_int64, C.c_int64, C.c_int32, c_stream]),
    'snet_halo_reverse': (C.c_int, [C.c_void_p, c_f32p, C.c_int64, C.c_int64, C.c_int32, c_stream]),
    'snet_halo_reverse_exchange': (C.c_int, [C.c_void_p, c_f32p, C.c_int64, C.c_int64, C.c_int32, c_stream]),
    'snet_halo_reverse_accumulate': (C.c_int, [C.c_void_p, c_f32p, C.c_int32, c_stream]),
    'snet_model_set_rccl_halo': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'snet_edge_pairs': (C.c_int, [c_i32p, c_i32p, c_f32p, C.c_int64, C.c_int64, c_i32p, c_i32p, C.POINTER(C.c_int64),
                                  c_stream]),
    'snet_d3_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'snet_d3_destroy': (None, [C.c_void_p]),
    'snet_d3_set_tables': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'snet_d3_settings': (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p]),
    'snet_d3_set_atoms': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'snet_d3_set_cell': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'snet_d3_compute': (C.c_int, [C.c_void_p, c_stream]),
    'snet_d3_energy': (C.c_double, [C.c_void_p]),
    'snet_d3_forces': (C.POINTER(C.c_double), [C.c_void_p]),
    'snet_d3_stress': (C.POINTER(C.c_double), [C.c_void_p]),
    'snet_d3_coordination_numbers': (C.POINTER(C.c_double), [C.c_void_p]),
    'snet_md_create': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'snet_md_destroy': (None, [C.c_void_p]),
    'snet_md_nodes': (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]),
    'snet_md_list_unchanged': (C.c_int, [C.c_void_p]),
    'snet_md_compute': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_stream]),
}

HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p)

_lib = None


def load():
    """dlopen libsnet_hip.so and type every entry point.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64 and owns the device context, streams and memory this
    # library works on: it must be loaded first so that libsnet_hip.so binds to the same HIP runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP force engine has not been built '
            '(run `python -m sevennet_amd.build`); there is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.snet_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH}: ABI version {lib.snet_abi_version()}, this package binds version {ABI_VERSION} '
                           '(include/snet_hip.h SNET_ABI_VERSION): rebuild with `python -m sevennet_amd.build`')
    _lib = lib
    return lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = load().snet_last_error()
        raise RuntimeError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')


def compiled_conv_tags():
    lib = load()
    return [lib.snet_conv_shape_tag(i).decode() for i in range(lib.snet_conv_num_shapes())]
