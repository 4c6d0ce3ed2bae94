"""ctypes binding of libquip_mi355.so (include/quip_mi355.h).  This module is the
only place that touches the native library; it never falls back to a CPU path:
a missing library or symbol raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUIP_LIB_PATH") or os.path.join(_HERE, "lib", "libquip_mi355.so")   # override: debug builds

_c = ctypes
_P, _I32, _I64, _F = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_float

# name -> argtypes ; every entry of include/quip_mi355.h
SIGNATURES = {
    "quip_abi_version": [],
    "quip_device_cu_count": [],
    "quip_hadamard_f16": [_P, _P, _I64, _I32, _F, _P],
    "quip_hadamard": [_P, _P, _I64, _I32, _F, _I32, _P],
    "quip_had_transform_f16": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P, _F, _P],
    "quip_had_transform_planes": [_P, _P, _I32, _I32, _I32, _P, _I32, _P, _F, _P],
    "quip_had_transform_fused_f16": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P, _F, _P, _P],
    "quip_had_transform_planes_fused": [_P, _P, _I32, _I32, _I32, _P, _I32, _P, _F, _P, _P],
    "quip_had_transform_group_f16": [_P, _I32, _I64, _I32, _I32, _I32, _P],
    "quip_had_transform_planes_group": [_P, _I32, _I32, _I32, _I32, _P],
    "quip_e8p_gemv_planes_group": [_P, _P, _P, _P, _P, _I32, _I32, _P],
    "quip_e8p_gemv_workspace_bytes": [_I32],
    "quip_e8p_gemv_planes_ws": [_P, _P, _P, _P, _I32, _I32, _P, _c.c_size_t, _P],
    "quip_e8p_gemv_planes_group_ws": [_P, _P, _P, _P, _P, _I32, _I32, _P, _c.c_size_t, _P],
    "quip_had_transform_planes_rows": [_P, _I64, _I32, _I32, _I32, _P],
    "quip_e8p_gemv_max_rows": [_I32, _I32],
    "quip_e8p_quantize_f32": [_P, _I64, _P, _P, _P, _P],
    "quip_gemv_max_rows_mode": [_I32, _I32, _I32],
    "quip_gemv_planes_rows_mode": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "quip_e8p_gemv_planes_rows": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "quip_e8prvq3_gemv_planes_group": [_P, _P, _P, _P, _P, _P, _I32, _I32, _P],
    "quip_e8prvq3_gemv_planes_group_ws": [_P, _P, _P, _P, _P, _P, _I32, _I32, _P, _c.c_size_t, _P],
    "quip_d4_gemv_planes": [_P, _P, _P, _P, _I32, _I32, _P],
    "quip_d4_gemv_planes_group": [_P, _P, _P, _P, _P, _I32, _I32, _P],
    "quip_d4_gemv_planes_v2": [_P, _P, _P, _P, _I32, _I32, _P, _c.c_size_t, _P],
    "quip_d4_gemv_planes_group_ws": [_P, _P, _P, _P, _P, _I32, _I32, _P, _c.c_size_t, _P],
    "quip_e8p_gemv_fused": [_P, _P, _P, _P, _P, _I32, _I32, _P],
    "quip_rope_attn_workspace_bytes": [_I32, _I32],
    "quip_argmax_step_f16": [_P, _I32, _P, _P, _P],
    "quip_rope_attn_decode_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _P, _P],
    "quip_rope_attn_decode_window_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _I32, _P, _P],
    "quip_rope_attn_decode_z_window_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _I32, _P, _P],
    "quip_rope_attn_decode_z_supported": [_I32, _I32, _I32],
    "quip_rope_attn_decode_z_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _P, _P],
    "quip_block_engine_supported": [_I32, _I32, _I32, _I32, _I32, _I32],
    "quip_block_engine_workspace_bytes": [],
    "quip_block_engine_layer_bytes": [],
    "quip_block_engine": [_P, _P],
    "quip_debug_occupy": [_I32, _I32, _c.c_int64, _P, _P],
    "quip_block_engine_gqa_supported": [_I32, _I32, _I32, _I32, _I32, _I32],
    "quip_block_engine_gqa_workspace_bytes": [],
    "quip_block_engine_g8_supported": [_I32, _I32, _I32, _I32, _I32, _I32],
    "quip_block_engine_g8_workspace_bytes": [],
    "quip_e8p_gemv_kernel_choice": [_P, _I32, _I32],
    "quip_ffn_engine_supported": [_I32, _I32, _I32],
    "quip_ffn_engine_workspace_bytes": [_I32, _I32],
    "quip_ffn_engine": [_P, _P],
    "quip_tile_codes": [_P, _P, _I64, _I64, _P],
    "quip_untile_codes": [_P, _P, _I64, _I64, _P],
    "quip_e8p_mm_origorder": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "quip_e8p_mm_batched": [_P, _P, _P, _P, _I64, _I32, _I32, _P],
    "quip_e8prvq4_mm_batched": [_P, _P, _P, _F, _P, _I64, _I32, _I32, _P],
    "quip_e8prvq3_mm_batched": [_P, _P, _P, _P, _F, _P, _I64, _I32, _I32, _P],
    "quip_d4_mm_batched": [_P, _P, _P, _P, _I64, _I32, _I32, _P],
    "quip_hi_mm_batched": [_P, _P, _P, _I64, _I32, _I32, _P],
    "quip_e8p_mm_skinny": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "quip_e8prvq4_mm_skinny": [_P, _P, _P, _F, _P, _I32, _I32, _I32, _P],
    "quip_e8prvq3_mm_skinny": [_P, _P, _P, _P, _F, _P, _I32, _I32, _I32, _P],
    "quip_d4_mm_skinny": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "quip_hi_mm_skinny": [_P, _P, _P, _I32, _I32, _I32, _P],
    "quip_e8p_mm_workspace_bytes": [_I32, _I32, _I32],
    "quip_e8p_mm_origorder_ws": [_P, _P, _P, _P, _I32, _I32, _I32, _P, _c.c_size_t, _P],
    "quip_e8p_planes_bytes": [_I32],
    "quip_e8p_x_to_planes": [_P, _P, _I32, _P],
    "quip_e8p_gemv_planes": [_P, _P, _P, _P, _I32, _I32, _P],
    "quip_e8prvq3_mm_origorder": [_P, _P, _P, _P, _F, _P, _I32, _I32, _I32, _P],
    "quip_e8prvq4_mm_origorder": [_P, _P, _P, _F, _P, _I32, _I32, _I32, _P],
    "quip_d4_mm_origorder": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "quip_hi_mm_origorder": [_P, _P, _P, _I32, _I32, _I32, _P],
    "quip_decompress_e8p_origorder": [_P, _P, _P, _I64, _I32, _P],
    "quip_decompress_e8prvq3_origorder": [_P, _P, _P, _F, _P, _I64, _I32, _P],
    "quip_decompress_e8prvq4_origorder": [_P, _P, _F, _P, _I64, _I32, _P],
    "quip_decompress_d4_origorder": [_P, _P, _P, _I64, _I32, _P],
    "quip_decompress_hi_origorder": [_P, _P, _I64, _I32, _P],
}
# not part of the public header: tuning hook used by the micro-benchmarks only
_INTERNAL = {
    "quip_e8p_x_to_planes_laneorder": [_P, _P, _I32, _P],
    "quip_e8p_gemv_fused_tuned": [_P, _P, _P, _P, _P, _I32, _I32, _P, _P],
    "quip_e8p_gemv_group_tuned": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "quip_e8p_gemv_v2_tuned": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "quip_e8p_gemv_v2_group_tuned": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "quip_e8p_gemv_v2_workspace_bytes": [_I32],
    "quip_e8p_gemv_tuned": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P],
}

class HadProblem(_c.Structure):
    """mirror of quip_had_problem (include/quip_mi355.h)"""
    _fields_ = [("x", _P), ("out", _P), ("had", _P), ("pre_scale", _P), ("pre_scale2", _P), ("post_scale", _P),
                ("bias", _P), ("residual", _P), ("rms_weight", _P), ("gate", _P), ("in_features", _I32),
                ("out_features", _I32), ("scale", _F), ("rms_eps", _F),
                ("z", _P), ("z_post_scale", _P), ("z_residual", _P), ("h_out", _P), ("z_scale", _F),
                ("resid_scale", _F), ("planes_layout", _I32), ("n", _I32)]


MAX_GROUP = 3


class GemvFusedIn(_c.Structure):
    """mirror of quip_gemv_fused_in (include/quip_mi355.h)"""
    _fields_ = [("x", _P), ("z", _P), ("post_scale", _P), ("residual", _P), ("h_out", _P), ("rms_weight", _P),
                ("pre_scale", _P * 3), ("scale", _F * 3), ("z_scale", _F), ("rms_eps", _F)]


class FfnEngineArgs(_c.Structure):
    """mirror of quip_ffn_engine_args (include/quip_mi355.h)"""
    _fields_ = [("w_gate", _P), ("w_up", _P), ("w_down", _P), ("planes_gate", _P), ("planes_up", _P), ("had3", _P),
                ("sv_gate", _P), ("sv_up", _P), ("su_down", _P), ("z_down", _P), ("grid_packed_abs", _P),
                ("workspace", _P), ("dbg", _P), ("out_scale", _F), ("in_scale", _F), ("hidden", _I32),
                ("n_ffn", _I32), ("K", _I32)]


class BlockEngineArgs(_c.Structure):
    """mirror of quip_block_engine_args (include/quip_mi355.h)"""
    _fields_ = [("layers", _P), ("h_in", _P), ("h_out", _P), ("pos", _P), ("cos", _P), ("sin", _P),
                ("grid_packed_abs", _P), ("workspace", _P), ("dbg", _P), ("n_layers", _I32), ("max_len", _I32),
                ("dbg_layer", _I32), ("rms_eps", _F), ("attn_scale", _F), ("codebook", _I32), ("resid_scale", _F),
                ("shape", _I32), ("grid2", _P)]


class HadFusion(_c.Structure):
    """mirror of quip_had_fusion (include/quip_mi355.h)"""
    _fields_ = [("residual", _P), ("rms_weight", _P), ("gate", _P), ("rms_eps", _F)]


_lib = None


class QuipNativeError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QuipNativeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in list(SIGNATURES.items()) + list(_INTERNAL.items()):
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.argtypes = args
            fn.restype = _c.c_size_t if name.endswith("_bytes") else _c.c_int
        L.quip_strerror.argtypes = [_c.c_int]
        L.quip_strerror.restype = _c.c_char_p
        _lib = L
    return _lib


def check_symbols():
    L = lib()
    missing = [n for n in list(SIGNATURES) + ["quip_strerror"] if not hasattr(L, n)]
    if missing:
        raise QuipNativeError(f"libquip_mi355.so lacks symbols: {missing}")
    return sorted(SIGNATURES) + ["quip_strerror"]


def check(code, what):
    if code != 0:
        msg = lib().quip_strerror(code).decode()
        raise QuipNativeError(f"{what} failed: {msg} ({code})")
