"""ctypes binding of the gfx950 C-ABI library (include/butd_pointnet2.h).

The product path has NO CPU fallback: if ``libbutd_detr_hip.so`` is missing or a symbol the headers
declare is absent, importing/using the ops raises immediately.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: torch ships its own libamdhip64; loading
              # ours first would bring up a second HIP runtime that does not see torch's context.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BUTD_HIP_LIB") or os.path.join(_HERE, "lib", "libbutd_detr_hip.so")  # env: debug hook

_c_int, _c_float, _c_void_p, _c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol the headers under include/ declare.
POINTNET2_SYMBOLS = {
    "butd_pointnet2_abi_version": (_c_int, []),
    "butd_error_string": (ctypes.c_char_p, [_c_int]),
    "butd_opt_n_threads": (_c_int, [_c_int]),
    "butd_furthest_point_sampling": (_c_int, [_c_int] * 3 + [_c_void_p] * 4),
    "butd_fps_prefix_check": (_c_int, [_c_int] * 3 + [_c_void_p] * 3),
    "butd_fps_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "butd_furthest_point_sampling_ws": (_c_int, [_c_int] * 3 + [_c_void_p] * 4 + [_c_size_t, _c_void_p]),
    "butd_gather_points": (_c_int, [_c_int] * 4 + [_c_void_p] * 4),
    "butd_gather_points_grad": (_c_int, [_c_int] * 4 + [_c_void_p] * 4),
    "butd_ball_query": (_c_int, [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 4),
    "butd_ball_query_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "butd_ball_query_ws": (_c_int, [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 4 + [_c_size_t, _c_void_p]),
    "butd_group_points": (_c_int, [_c_int] * 5 + [_c_void_p] * 4),
    "butd_group_points_grad": (_c_int, [_c_int] * 5 + [_c_void_p] * 4),
    "butd_three_nn": (_c_int, [_c_int] * 3 + [_c_void_p] * 5),
    "butd_three_interpolate": (_c_int, [_c_int] * 4 + [_c_void_p] * 5),
    "butd_three_interpolate_grad": (_c_int, [_c_int] * 4 + [_c_void_p] * 5),
}

_c_long, _c_u32 = ctypes.c_long, ctypes.c_uint32


class GemmProblem(ctypes.Structure):
    """ctypes mirror of ``butd_gemm_problem`` (include/butd_attention.h)."""
    _fields_ = [("a", _c_void_p), ("a2", _c_void_p), ("b", _c_void_p), ("bias", _c_void_p),
                ("c", _c_void_p), ("bias_grad", _c_void_p),
                ("M", _c_int), ("N", _c_int), ("K", _c_int),
                ("lda_m", _c_long), ("lda_k", _c_long), ("ldb_n", _c_long), ("ldb_k", _c_long),
                ("ldc", _c_long),
                ("scale", _c_float), ("a2_mode", _c_int), ("a2_scale", _c_float),
                ("relu", _c_int), ("accumulate", _c_int), ("ones_col", _c_int), ("split_k", _c_int),
                ("dropout_p", _c_float), ("dropout_site", _c_u32),
                ("a_chan_scale", _c_void_p), ("a_chan_shift", _c_void_p),
                ("b_chan_scale", _c_void_p), ("b_chan_shift", _c_void_p),
                ("a_drop_p", _c_float), ("a_drop_site", _c_u32),
                ("b_drop_p", _c_float), ("b_drop_site", _c_u32),
                ("col_sum", _c_void_p), ("col_sumsq", _c_void_p),
                ("c_add", _c_int), ("c2", _c_void_p), ("col_slots", _c_int), ("col_slot_stride", _c_long),
                ("compute_bf16", _c_int), ("c_gate", _c_void_p), ("c_gate_scale", _c_float),
                ("a_bn_sum", _c_void_p), ("a_bn_sumsq", _c_void_p), ("a_bn_gamma", _c_void_p), ("a_bn_beta", _c_void_p),
                ("a_bn_running_mean", _c_void_p), ("a_bn_running_var", _c_void_p), ("a_bn_nbt", _c_void_p),
                ("a_bn_out", _c_void_p), ("a_bn_ld", _c_long), ("a_bn_count", _c_long),
                ("a_bn_eps", _c_float), ("a_bn_momentum", _c_float),
                ("c_bn_z", _c_void_p), ("c_bn_aff", _c_void_p), ("c_bn_ld", _c_long),
                ("c_bn_drop_p", _c_float), ("c_bn_drop_site", _c_u32),
                ("c_partial", _c_void_p), ("c_partial_stride", _c_long),
                ("fold_src", _c_void_p), ("fold_count", _c_int),
                ("fold_stride", _c_long), ("fold_len", _c_long), ("fold_len2", _c_long)]


ATTENTION_SYMBOLS = {
    "butd_gemm_grouped": (_c_int, [ctypes.POINTER(GemmProblem), _c_int, _c_void_p, _c_void_p]),
    "butd_gemm_set_tile": (_c_int, [_c_int, _c_int]),
    "butd_attention_fwd": (_c_int, [_c_int] * 5 + [_c_void_p] * 6 + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_attention_bwd": (_c_int, [_c_int] * 5 + [_c_void_p] * 11 + [_c_long, _c_long, _c_float]
                           + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_attention_bwd_short_keys_max": (_c_int, []),
    "butd_attention_bwd_short_keys": (_c_int, [_c_int] * 5 + [_c_void_p] * 11 + [_c_long, _c_long, _c_float]
                                      + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_attention_bwd_long_keys_set_chunk": (_c_int, [_c_int, _c_int]),
    "butd_attention_bwd_long_keys_scratch": (_c_long, [_c_int] * 5 + [_c_long]),
    "butd_attention_bwd_long_keys": (_c_int, [_c_int] * 5 + [_c_void_p] * 10 + [_c_long, _c_long, _c_float]
                                     + [_c_float, _c_u32, _c_void_p, _c_void_p, _c_long, _c_void_p]),
    "butd_attention_bwd_long_keys_bf16_scratch": (_c_long, [_c_int] * 5 + [_c_long]),
    "butd_attention_bwd_long_keys_bf16": (_c_int, [_c_int] * 5 + [_c_void_p] * 10 + [_c_long, _c_long, _c_float]
                                          + [_c_float, _c_u32, _c_void_p, _c_void_p, _c_long, _c_void_p]),
    "butd_attention_fwd_bf16": (_c_int, [_c_int] * 5 + [_c_void_p] * 6 + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_attention_fwd_split_bf16": (_c_int, [_c_int] * 5 + [_c_void_p] * 6 + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_attention_bwd_bf16": (_c_int, [_c_int] * 5 + [_c_void_p] * 11 + [_c_long, _c_long, _c_float]
                                + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_add_dropout_layernorm_fwd": (_c_int, [_c_int, _c_int] + [_c_void_p] * 4 + [_c_float] + [_c_void_p] * 3
                                       + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_add_dropout_layernorm_fwd_pos": (_c_int, [_c_int, _c_int] + [_c_void_p] * 4 + [_c_float] + [_c_void_p] * 3
                                           + [_c_float, _c_u32, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "butd_add_dropout_layernorm_bwd": (_c_int, [_c_int, _c_int] + [_c_void_p] * 10
                                       + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_layernorm_bwd_blocks": (_c_int, [_c_int]),
    "butd_add_dropout_layernorm_bwd_partial": (_c_int, [_c_int, _c_int] + [_c_void_p] * 9
                                               + [_c_float, _c_u32, _c_void_p, _c_void_p]),
}

_P = _c_void_p
SA_SYMBOLS = {
    "butd_sa_group": (_c_int, [_c_int] * 5 + [_P, _P, _P, _c_long, _P, _c_float, _c_int, _P, _c_int, _P]),
    "butd_sa_thin_conv": (_c_int, [_c_long, _c_int, _c_int, _P, _c_int, _P, _P, _P, _P, _P]),
    "butd_sa_colstats": (_c_int, [_c_long, _c_int, _P, _P, _P, _c_int, _P, _P, _P, _P, _P]),
    "butd_sa_bn_finalize": (_c_int, [_c_int, _c_long, _P, _P, _c_int, _c_long, _P, _P, _c_float, _c_float, _c_int]
                            + [_P] * 7 + [_P]),
    "butd_sa_pool_finalize": (_c_int, [_c_int] * 3 + [_P] * 10 + [_P]),
    "butd_sa_pool_bwd_stats": (_c_int, [_c_int] * 3 + [_P] * 8 + [_P]),
    "butd_sa_dz_last": (_c_int, [_c_int] * 4 + [_P] * 11 + [_c_int, _P]),
    "butd_sa_mask_stats": (_c_int, [_c_long, _c_int] + [_P] * 8 + [_P]),
    "butd_sa_last_bwd_supported": (_c_int, [_c_int] * 3),
    "butd_sa_last_fwd": (_c_int, [_c_int] * 5 + [_P] * 11 + [_P]),
    "butd_sa_first_bwd_scratch": (_c_int, [_c_long, _c_int, _c_int, _P, _P]),
    "butd_sa_first_bwd": (_c_int, [_c_long, _c_int, _c_int] + [_P] * 13 + [_P]),
    "butd_sa_first_two_fwd": (_c_int, [_c_long, _c_int, _c_int] + [_P] * 11 + [_c_int, _P]),
    "butd_sa_mid_wide_bwd_scratch": (_c_int, [_c_long, _c_int, _P, _P]),
    "butd_sa_mid_wide_bwd": (_c_int, [_c_long, _c_int] + [_P] * 21 + [_P]),
    "butd_sa_mid_first_bwd_scratch": (_c_int, [_c_long, _c_int, _c_int, _P, _P]),
    "butd_sa_mid_first_bwd": (_c_int, [_c_long, _c_int, _c_int] + [_P] * 23 + [_P]),
    "butd_sa_last_bwd_scratch": (_c_int, [_c_long, _c_int, _c_int, _P, _P]),
    "butd_sa_last_bwd": (_c_int, [_c_int] * 5 + [_P] * 21 + [_P]),
    "butd_sa_first_linear_supported": (_c_int, [_c_int]),
    "butd_sa_first_linear_fwd": (_c_int, [_c_int] * 5 + [_P, _P, _P, _c_float, _c_int, _P, _P, _c_long, _P, _P, _P, _c_int,
                                                        _c_long, _P]),
    "butd_sa_first_linear_bwd_scratch": (_c_int, [_c_int, _c_int, _c_int, _P]),
    "butd_sa_first_linear_bwd": (_c_int, [_c_int] * 5 + [_P, _P, _P, _P, _c_float, _c_int] + [_P] * 9 + [_c_int, _P, _P,
                                                                                                   _c_long, _P, _P]),
    "butd_sa_dz_mid": (_c_int, [_c_long, _c_int] + [_P] * 9 + [_c_int, _P]),
    "butd_sa_scatter_rows": (_c_int, [_c_int] * 5 + [_P, _c_int, _P, _P] + [_P]),
    "butd_sa_inverse_index": (_c_int, [_c_int] * 4 + [_P] * 4 + [_P]),
    "butd_sa_gather_rows": (_c_int, [_c_int] * 3 + [_P, _c_int, _P, _P, _P] + [_P]),
    "butd_sa_fused_eval": (_c_int, [_c_int] * 5 + [_P, _P, _P, _c_long, _P, _c_float, _c_int] + [_P] * 7 + [_P]),
}

OPTIM_SYMBOLS = {
    "butd_adamw_flat": (_c_int, [_P] * 4 + [_c_long, _c_long] + [_c_float] * 5 + [_P, _P, _P, _P]),
    "butd_gather_segments": (_c_int, [_c_int, _P, _P, _P, _c_long]),
    "butd_sum_tensors": (_c_int, [_c_int, _P, _c_long, _P, _P]),
    "butd_clip_workspace_bytes": (ctypes.c_size_t, []),
    "butd_clip_coefficient": (_c_int, [_P, _c_long, _c_float, _c_float, _P, _P, _P, _P]),
}

MLP_MAX_SEGMENTS = 8  # BUTD_MLP_MAX_SEGMENTS


class BnSegment(ctypes.Structure):
    """ctypes mirror of ``butd_bn_segment`` (include/butd_mlp.h)."""
    _fields_ = [("gamma", _c_void_p), ("beta", _c_void_p), ("running_mean", _c_void_p),
                ("running_var", _c_void_p), ("num_batches_tracked", _c_void_p)]


MLP_SYMBOLS = {
    "butd_mlp_bn_finalize": (_c_int, [_c_int, _c_int, _c_long, _P, _P, ctypes.POINTER(BnSegment), _c_float,
                                      _c_float, _c_int, _P, _P, _P, _P, _P]),
    "butd_mlp_mask_stats": (_c_int, [_c_long, _c_int, _c_long] + [_P] * 6 + [_c_float, _c_u32, _c_int, _P, _P, _P, _P]),
    "butd_mlp_dz": (_c_int, [_c_long, _c_int, _c_long] + [_P] * 7 + [_c_int, _P, _P, _P]),
    "butd_mlp_bn_relu_apply": (_c_int, [_c_long, _c_int, _c_long, _P, _P, _P, _P, _P]),
}

LSAP_SYMBOLS = {
    "butd_hungarian_match": (_c_int, [_c_int] * 3 + [_P] * 4 + [_P]),
}

CRITERION_SYMBOLS = {
    "butd_match_cost": (_c_int, [_c_int] * 4 + [_P] * 4 + [_c_float] * 3 + [_P, _P]),
    "butd_box_loss": (_c_int, [_c_int] * 4 + [_P] * 5 + [_P]),
    "butd_box_loss_bwd": (_c_int, [_c_int] * 4 + [_P] * 4 + [_P]),
    "butd_soft_token_ce": (_c_int, [_c_int] * 5 + [_P] * 3 + [_c_int, _c_float, _P, _P, _P]),
    "butd_contrastive_rows": (_c_int, [_c_int] * 5 + [_P] * 3 + [_c_int, _P, _c_float, _P, _P, _P, _P]),
    "butd_seed_objectness": (_c_int, [_c_int] * 5 + [_P] * 10 + [_P]),
    "butd_loss_combine": (_c_int, [_c_int] + [_P] * 6 + [_c_int] + [_c_float] * 3 + [_P, _P]),
    "butd_criterion_reduce": (_c_int, [_c_int, _P, _c_long, _P, _P, _c_long, _P, _c_long, _P, _c_long, _c_float, _P, _P, _c_int]
                              + [_c_float] * 3 + [_P, _P, _P]),
    "butd_criterion_scale": (_c_int, [_c_int, _P, _P, _P, _c_int] + [_c_float] * 4 + [_P, _P, _c_long] * 3 + [_P, _P]),
    "butd_loss_combine_bwd": (_c_int, [_c_int, _P, _P, _c_int] + [_c_float] * 3 + [_P] * 5 + [_P]),
    "butd_contrastive_cols": (_c_int, [_c_int] * 5 + [_P] * 3 + [_c_int, _P, _c_float, _P, _P, _P]),
}

class SceneAugment(ctypes.Structure):
    """ctypes mirror of ``butd_scene_augment`` (include/butd_augment.h)."""
    _fields_ = [("rz", ctypes.c_double * 9), ("rx", ctypes.c_double * 9), ("ry", ctypes.c_double * 9),
                ("shift", ctypes.c_double * 3), ("scale", ctypes.c_double),
                ("flip_yz", ctypes.c_int32), ("flip_xz", ctypes.c_int32)]


_c_double, _c_u64 = ctypes.c_double, ctypes.c_uint64
AUGMENT_SYMBOLS = {
    "butd_augment_points": (_c_int, [_c_int] * 4 + [_P] * 4 + [_c_double] * 3 + [_c_u64, _P, _P]),
    "butd_augment_boxes": (_c_int, [_c_int, _c_int, _P, _P, _P, _P]),
    "butd_instance_boxes": (_c_int, [_c_int] * 4 + [_P] * 6 + [_P]),
    "butd_object_boxes": (_c_int, [_c_int] * 4 + [_P, _P, ctypes.c_longlong] + [_P] * 8 + [_P]),
}

ROWWISE_SYMBOLS = {
    "butd_l2_normalize_fwd": (_c_int, [_c_long, _c_int, _P, _c_float, _P, _P]),
    "butd_l2_normalize_bwd": (_c_int, [_c_long, _c_int, _P, _P, _c_float, _P, _P]),
    "butd_three_nn_weights": (_c_int, [_c_long, _P, _P, _P, _P]),
}

GRAPH_SYMBOLS = {
    "butd_graph_replace_memset_nodes": (_c_int, [_P, ctypes.POINTER(ctypes.c_int)]),
    "butd_graph_node_counts": (_c_int, [_P, ctypes.POINTER(ctypes.c_int * 16)]),
    "butd_runtime_versions": (_c_int, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "butd_timeline_mark": (_c_int, [_P, _c_int, _P]),
    "butd_stream_create": (_c_int, [_c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "butd_stream_destroy": (_c_int, [_P]),
}

ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
ALL_SYMBOLS.update(ATTENTION_SYMBOLS)
ALL_SYMBOLS.update(SA_SYMBOLS)
ALL_SYMBOLS.update(OPTIM_SYMBOLS)
ALL_SYMBOLS.update(MLP_SYMBOLS)
ALL_SYMBOLS.update(LSAP_SYMBOLS)
ALL_SYMBOLS.update(CRITERION_SYMBOLS)
ALL_SYMBOLS.update(AUGMENT_SYMBOLS)
ALL_SYMBOLS.update(GRAPH_SYMBOLS)
ALL_SYMBOLS.update(ROWWISE_SYMBOLS)

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """dlopen the library and bind every declared entry point (loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m butd_detr_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in ALL_SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryMissing(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(err, what):
    if err != 0:
        msg = load().butd_error_string(err)
        raise RuntimeError(f"{what}: HIP error {err} ({msg.decode() if msg else '?'})")
