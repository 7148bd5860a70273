"""ctypes binding of libgps_hip.so (include/gps_hip.h).

The product path has NO CPU fallback: if the shared library cannot be built/loaded, or an op is
called on a non-GPU tensor, this module raises.  (oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "csrc", "libgps_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "gps_hip.h")

GPS_OK = 0
GPS_ERR_INVALID_ARGUMENT = -1
GPS_ERR_UNSUPPORTED = -2
_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float



class GemmArgs(ctypes.Structure):
    """struct gps_gemm_args of include/gps_hip.h, field for field."""
    _fields_ = [("form", _i), ("epilogue", _i), ("M", _i), ("N", _i), ("K", _i), ("splits", _i), ("variant", _i),
                ("reserved", _i),
                ("A", _vp), ("lda", ctypes.c_longlong), ("B", _vp), ("ldb", ctypes.c_longlong),
                ("C", _vp), ("ldc", ctypes.c_longlong), ("bias", _vp),
                ("aux", _vp), ("ldaux", ctypes.c_longlong), ("aux_out", _vp), ("ldaux_out", ctypes.c_longlong),
                ("workspace", _vp), ("colsum", _vp), ("seed_dev", _vp), ("seed", ctypes.c_ulonglong),
                ("p_drop", _f), ("reserved2", _i), ("extent_dev", _vp)]


class WgradProblem(ctypes.Structure):
    """struct gps_wgrad_problem of include/gps_hip.h, field for field."""
    _fields_ = [("M", _i), ("N", _i), ("K", _i), ("accumulate", _i),
                ("A", _vp), ("lda", ctypes.c_longlong), ("B", _vp), ("ldb", ctypes.c_longlong),
                ("C", _vp), ("ldc", ctypes.c_longlong), ("colsum", _vp), ("extent_dev", _vp)]


class LnReduceProblem(ctypes.Structure):
    """struct gps_ln_reduce_problem of include/gps_hip.h, field for field."""
    _fields_ = [("part", _vp), ("out_gamma", _vp), ("out_beta", _vp), ("parts", _i), ("accumulate", _i)]


class VarlenText(ctypes.Structure):
    """struct gps_varlen_text of include/gps_hip.h, field for field."""
    _fields_ = [("ids", _vp), ("mask", _vp), ("mask_elem_bytes", _i), ("mask_is_float", _i), ("n_seq", _i), ("len", _i)]


class AttnArgs(ctypes.Structure):
    """struct gps_attn_args of include/gps_hip.h, field for field."""
    _fields_ = [("B", _i), ("H", _i), ("Lq", _i), ("Lk", _i), ("head_dim", _i), ("dtype", _i), ("compute", _i),
                ("reserved", _i),
                ("q", _vp), ("ld_q", _i), ("k", _vp), ("v", _vp), ("ld_kv", _i),
                ("sw", _vp), ("pl", _vp), ("mask", _vp),
                ("p_drop", _f), ("seed", ctypes.c_ulonglong), ("seed_dev", _vp),
                ("out", _vp), ("ld_o", _i), ("lse", _vp),
                ("dout", _vp), ("dq", _vp), ("ld_dq", _i), ("dk", _vp), ("dv", _vp), ("ld_dkv", _i), ("dsw", _vp),
                ("cu_rows", _vp), ("seq_order", _vp), ("q_limit", _vp),
                ("pl_planes", _vp), ("ld_pl", _i), ("sw16", _vp), ("ld_sw", _i), ("dsw16", _vp), ("ld_dsw", _i),
                ("delta_ws", _vp)]


ATTN_BF16, ATTN_F32 = 0, 1
ATTN_COMPUTE_NATIVE, ATTN_COMPUTE_FP8 = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_DGELU, EPI_DRELU, EPI_F32 = 0, 1, 2, 3, 4, 5
EPI_RELU_SPLIT, EPI_RELU_MAX16 = 6, 7
EPI_BIAS_GELU_FACTOR, EPI_MUL_AUX = 8, 9

# entries of SIGNATURES whose return value is not an `int` status
NON_STATUS_RESTYPES = {"gps_embedding_grad_scratch_ints": ctypes.c_longlong, "gps_point_set_object_extent": None}

# name -> argtypes, mirroring include/gps_hip.h one to one
SIGNATURES = {
    "gps_adamw_step": [_i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp],
    "gps_gemm_pick_splits": [_i, _i, _i, _i],
    "gps_gemm_pick_variant": [_i, _i, _i, _i, _i],
    "gps_gemm_pick_variant_ex": [_i, _i, _i, _i, _i, _i],
    "gps_gemm_wgrad_grouped_set_xcd_queues": [_i],
    "gps_gemm_bf16": [ctypes.POINTER(GemmArgs), _vp],
    "gps_gemm_bf16_grouped": [ctypes.POINTER(GemmArgs), _i, _vp],
    "gps_gemm_wgrad_grouped": [ctypes.POINTER(WgradProblem), _i, _vp],
    "gps_furthest_point_sampling": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_furthest_point_sampling_xyz": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_cloud_compact": [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_point_set_object_extent": [_vp],
    "gps_gather_points": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_gather_points_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_ball_query": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp],
    "gps_group_points": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_group_points_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_three_nn": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "gps_three_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "gps_three_interpolate_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "gps_pairwise_locs": [_i, _i, _vp, _f, _vp, _vp],
    "gps_pairwise_locs_planes": [_i, _i, _vp, _f, _vp, _vp, _i, _vp],
    "gps_pairwise_to_planes": [_i, _i, _vp, _vp, _i, _vp],
    "gps_sa_mlp_pack_layer": [_i, _i, _vp, _vp, _vp, _vp],
    "gps_sa_mlp_forward": [_i] * 8 + [_vp] * 7,
    "gps_sa_mlp_pack_layer_bf16x3": [_i, _i, _vp, _vp, _vp, _vp],
    "gps_sa_mlp_forward_bf16x3": [_i] * 8 + [_vp] * 7,
    "gps_sa_mlp_forward_bf16x3_pm": [_i] * 8 + [_vp] * 3 + [ctypes.c_longlong] + [_vp] * 4,
    "gps_obj_processing_post": [_i, _i, _vp, _vp, _i, _vp, _vp, _vp, ctypes.c_ulonglong, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp],
    "gps_embedding_grad_scratch_ints": [_i, _i, _i],
    "gps_embedding_grad": [_i, _i, _i, _vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, _vp],
    "gps_loc_embed_partial_rows": [_i],
    "gps_loc_embed_forward": [_i, _i, _i] + [_vp] * 5 + [_f] + [_vp] * 4,
    "gps_loc_embed_backward": [_i, _i, _i] + [_vp] * 10,
    "gps_bert_embed_partial_rows": [_i],
    "gps_bert_position_grad": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "gps_varlen_plan": [ctypes.POINTER(VarlenText), _i, _i, _vp, _vp, _vp, _vp],
    "gps_rows_plan": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_rows_move": [_i, ctypes.c_longlong, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "gps_rows_pack2": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_rows_unpack2": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_joint_embed_forward": [_i, _i, _i, _i] + [_vp] * 10,
    "gps_joint_embed_backward": [_i, _i, _i, _i] + [_vp] * 10,
    "gps_bert_embed_forward": [_i, _i] + [_vp] * 7 + [_f, _f, ctypes.c_ulonglong] + [_vp] * 8,
    "gps_bert_embed_backward": [_i, _i] + [_vp] * 10 + [_f, ctypes.c_ulonglong] + [_vp] * 6,
    "gps_colsum_parts": [_i, _i],
    "gps_colsum_bf16": [_i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp],
    "gps_split3_points": [_i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "gps_masked_ce_forward": [_i, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp],
    "gps_masked_ce_backward": [_i, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp,
                               ctypes.c_longlong, _vp],
    "gps_masked_ce_forward_rows": [_i, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_masked_ce_backward_rows": [_i, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp,
                                    ctypes.c_longlong, _vp],
    "gps_lm_row_plan": [_i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp],
    "gps_text_obj_ce_forward": [_i, _i, _i, _vp, _vp, _vp, _vp, _f, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_text_obj_ce_backward": [_i, _i, _i, _vp, _vp, _vp, _f, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp],
    "gps_clip_loss_forward": [_i, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gps_clip_loss_backward": [_i, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp],
    "gps_ln_reduce_partials": [_i, _i, _vp, _vp, _vp, _vp],
    "gps_ln_reduce_partials_grouped": [ctypes.POINTER(LnReduceProblem), _i, _i, _vp, _vp],
    "gps_l2_normalize_forward": [_i, _i, _vp, _f, _vp, _vp, _vp],
    "gps_l2_normalize_backward": [_i, _i, _vp, _vp, _vp, _f, _vp, _vp],
    "gps_add_dropout_layernorm_forward": [_i] * 4 + [_vp] * 4 + [_f, _f, ctypes.c_ulonglong] + [_vp] * 6,
    "gps_add_dropout_layernorm_backward": [_i] * 4 + [_vp] * 7 + [_f, ctypes.c_ulonglong] + [_vp] * 6,
    "gps_add_dropout_layernorm_forward_rows": [_i] * 4 + [_vp] * 4 + [_f, _f, ctypes.c_ulonglong] + [_vp] * 7,
    "gps_add_dropout_layernorm_backward_rows": [_i] * 4 + [_vp] * 7 + [_f, ctypes.c_ulonglong] + [_vp] * 7,
    "gps_add_dropout_layernorm_forward_post": [_i] * 4 + [_vp] * 4 + [_f, _f, ctypes.c_ulonglong] + [_vp] * 8,
    "gps_add_dropout_layernorm_backward_post": [_i] * 4 + [_vp] * 7 + [_f, ctypes.c_ulonglong] + [_vp] * 8,
    "gps_add_dropout_layernorm_backward_post_acc": [_i] * 4 + [_vp] * 7 + [_f, ctypes.c_ulonglong] + [_vp] * 7 + [_i, _vp],
    "gps_attn_forward": [_i] * 4 + [_vp] * 3 + [_i] + [_vp] * 3 + [_f, ctypes.c_ulonglong, _vp, _vp, _i, _vp, _vp],
    "gps_attn_set_plain_blocks": [_i],
    "gps_sa_mlp_set_products": [_i],
    "gps_attn_forward_ex": [ctypes.POINTER(AttnArgs), _vp],
    "gps_attn_backward_ex": [ctypes.POINTER(AttnArgs), _vp],
    "gps_attn_backward": [_i] * 4 + [_vp] * 3 + [_i] + [_vp] * 3 + [_f, ctypes.c_ulonglong, _vp, _vp, _i] + [_vp] * 7,
}


class GpsNativeError(RuntimeError):
    pass


def declared_symbols() -> list[str]:
    """Every function include/gps_hip.h declares (used by the symbol-export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return re.findall(r"GPS_API\s+[\w\s\*]+?\b(gps_\w+)\s*\(", text)


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing:
        from .csrc import build as _build
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(LIB_PATH):
                raise GpsNativeError(f"libgps_hip.so is missing and could not be built: {e}") from e
            # a library that is OLDER than its sources must never be loaded silently (a source that no longer compiles
            # would otherwise be "tested" through yesterday's binary); only a machine without hipcc may use what it has
            if getattr(_build, "hipcc_available", lambda: True)():
                raise GpsNativeError(f"libgps_hip.so is out of date and the rebuild failed: {e}") from e
    if not os.path.exists(LIB_PATH):
        raise GpsNativeError(f"{LIB_PATH} not found; run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(LIB_PATH)
    lib.gps_abi_version.restype = _i
    lib.gps_error_string.restype = ctypes.c_char_p
    lib.gps_error_string.argtypes = [_i]
    lib.gps_last_hip_error.restype = ctypes.c_char_p
    lib.gps_sa_mlp_wpack_floats.restype = ctypes.c_longlong
    lib.gps_sa_mlp_wpack_floats.argtypes = [_i, _i, _i, _i]
    lib.gps_sa_mlp_layer_floats.restype = ctypes.c_longlong
    lib.gps_sa_mlp_layer_floats.argtypes = [_i, _i]
    lib.gps_sa_mlp_layer_floats_bf16x3.restype = ctypes.c_longlong
    lib.gps_sa_mlp_layer_floats_bf16x3.argtypes = [_i, _i]
    lib.gps_gemm_workspace_floats.restype = ctypes.c_longlong
    lib.gps_gemm_workspace_floats.argtypes = [_i, _i, _i, _i]
    lib.gps_adamw_chunk_elems.restype = _i
    lib.gps_adamw_chunk_elems.argtypes = []
    lib.gps_attn_set_stream_min_tiles.restype = None
    lib.gps_attn_set_stream_min_tiles.argtypes = [_i, _i]
    lib.gps_ln_partial_rows.restype = _i
    lib.gps_ln_partial_rows.argtypes = [_i]
    lib.gps_ln_reduce_scratch_bytes.restype = ctypes.c_longlong
    lib.gps_ln_reduce_scratch_bytes.argtypes = [_i]
    lib.gps_ln_reduce_grouped_scratch_bytes.restype = ctypes.c_longlong
    lib.gps_ln_reduce_grouped_scratch_bytes.argtypes = [_i]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    # the entries of SIGNATURES that do NOT return a status (set after the loop above, which would overwrite them:
    # a 64-bit scratch size read as c_int under-allocates for large n * d)
    for name, restype in NON_STATUS_RESTYPES.items():
        getattr(lib, name).restype = restype
    lib.gps_gemm_sk_workspace_bytes.restype = ctypes.c_longlong
    lib.gps_gemm_sk_workspace_bytes.argtypes = []
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status == GPS_OK:
        return
    lib = load()
    msg = lib.gps_error_string(status).decode()
    hip = lib.gps_last_hip_error().decode()
    raise GpsNativeError(f"{what}: {msg}" + (f" [{hip}]" if hip else ""))
