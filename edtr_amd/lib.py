"""ctypes binding of libedtr_hip.so (include/edtr_hip.h).  The product path has NO fallback: if the
library is missing or a launch fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDTR_AMD_LIB") or os.path.join(HERE, "libedtr_hip.so")     # (EDTR_AMD_LIB: another build of the same ABI, for A/B runs on one device)

BF16, F16, F32_SPLIT, F32_H1, F32_H2, F32_H3 = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_GEGLU, ACT_SILU, ACT_GELU, ACT_LRELU = 0, 1, 2, 3, 4

DECLARED_SYMBOLS = [
    "edtr_abi_version", "edtr_error_string", "edtr_device_info", "edtr_igemm", "edtr_flash_attn64",
    "edtr_gn_stats", "edtr_gn_apply", "edtr_gn_finalize", "edtr_gn_table", "edtr_layernorm", "edtr_softmax_rows", "edtr_nchw_to_nhwc",
    "edtr_nhwc_to_nchw", "edtr_add", "edtr_timestep_embedding", "edtr_sampler_update", "edtr_axpby", "edtr_q_sample", "edtr_split3", "edtr_cast16",
    "edtr_tile_accumulate", "edtr_divide", "edtr_wavelet_level", "edtr_gn_pool", "edtr_copy3d_f32", "edtr_graph_begin", "edtr_graph_end", "edtr_graph_launch",
    "edtr_graph_destroy", "edtr_zero_bytes", "edtr_embed_tokens", "edtr_window_attn", "edtr_pixel_unshuffle", "edtr_swin_mlp", "edtr_swin_attn", "edtr_swin_layer", "edtr_conv64", "edtr_conv128_out",
    "edtr_split_operand", "edtr_sampler_update_indexed", "edtr_gaussian_sample", "edtr_add_mirror", "edtr_igemm_plan", "edtr_flash_attn512",
    "edtr_ffn", "edtr_ffn_plan", "edtr_add_stats", "edtr_lin320", "edtr_lin320_plan",
]


class IgemmParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("taps", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("n_valid", C.c_int32), ("Z", C.c_int32), ("zdiv", C.c_int32),
        ("a1", C.c_void_p), ("a2", C.c_void_p),
        ("C1", C.c_int32), ("C2", C.c_int32), ("ld1", C.c_int32), ("ld2", C.c_int32),
        ("a_zs_outer", C.c_int64), ("a_zs_inner", C.c_int64),
        ("IH", C.c_int32), ("IW", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("upsample2x", C.c_int32),
        ("w", C.c_void_p), ("ldw", C.c_int32),
        ("w_zs_outer", C.c_int64), ("w_zs_inner", C.c_int64),
        ("alpha", C.c_float),
        ("bias_n", C.c_void_p), ("bias_m", C.c_void_p),
        ("rowvec", C.c_void_p), ("rowvec_ld", C.c_int32), ("rows_per_image", C.c_int32),
        ("act", C.c_int32),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("out", C.c_void_p), ("ldc", C.c_int32), ("out_f32", C.c_int32),
        ("o_zs_outer", C.c_int64), ("o_zs_inner", C.c_int64),
        ("tile", C.c_int32), ("splitk", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("gn_partial", C.c_void_p),
        ("act_slope", C.c_float),
        ("residual_f32", C.c_int32),
        ("vt_out", C.c_void_p), ("vt_col0", C.c_int32), ("vt_ld", C.c_int32), ("vt_alpha", C.c_float),
        ("row_stats", C.c_void_p),
        ("ln_stats", C.c_void_p), ("ln_slots", C.c_int32), ("ln_C", C.c_int32), ("ln_eps", C.c_float),
        ("ln_c1", C.c_void_p), ("ln_c2", C.c_void_p),
        ("stagger", C.c_int32), ("debug_flags", C.c_int32),
        ("w_phase_stride", C.c_int64),
        ("out16", C.c_void_p), ("ld16", C.c_int32),
        ("a_wrap", C.c_int32),
        ("a_gn", C.c_void_p), ("a_gn_silu", C.c_int32),
        ("gn_slot_rows", C.c_int32), ("gn_ld", C.c_int32),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32),
        ("q", C.c_void_p), ("q_bs", C.c_int64), ("q_ld", C.c_int32),
        ("k", C.c_void_p), ("k_bs", C.c_int64), ("k_ld", C.c_int32),
        ("vt", C.c_void_p), ("vt_bs", C.c_int64), ("vt_ld", C.c_int32),
        ("out", C.c_void_p), ("o_bs", C.c_int64), ("o_ld", C.c_int32),
        ("scale", C.c_float),
        ("causal", C.c_int32),
        ("q_prescaled", C.c_int32),
        ("q_lo", C.c_void_p), ("k_lo", C.c_void_p), ("vt_lo", C.c_void_p),
        ("out_f32", C.c_int32),
    ]


class WindowAttnParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("heads", C.c_int32), ("head_dim", C.c_int32), ("shift", C.c_int32),
        ("qkv", C.c_void_p), ("ld_qkv", C.c_int32),
        ("out", C.c_void_p), ("ld_out", C.c_int32), ("c_pad", C.c_int32),
        ("bias", C.c_void_p), ("labels", C.c_void_p),
        ("scale", C.c_float),
    ]


class SwinMlpParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("rows", C.c_int32), ("C", C.c_int32), ("hidden", C.c_int32), ("c_valid", C.c_int32),
        ("eps", C.c_float),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("w1", C.c_void_p), ("w2", C.c_void_p),
        ("c1", C.c_void_p), ("c2b", C.c_void_p), ("b2", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("row_stats", C.c_void_p),
    ]


class FfnParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("M", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("eps", C.c_float),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("w1", C.c_void_p), ("w2", C.c_void_p),
        ("cst", C.c_void_p), ("b2", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
    ]


class Lin320Params(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("ln", C.c_int32), ("eps", C.c_float), ("alpha", C.c_float),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("w", C.c_void_p), ("cvec", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("vt_out", C.c_void_p), ("vt_col0", C.c_int32), ("vt_ld", C.c_int32), ("vt_alpha", C.c_float), ("rows_per_image", C.c_int32),
        ("gn_table", C.c_void_p),
    ]


class SwinAttnParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("heads", C.c_int32), ("head_dim", C.c_int32), ("shift", C.c_int32),
        ("C", C.c_int32), ("c_valid", C.c_int32),
        ("eps", C.c_float),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("wqkv", C.c_void_p), ("wproj", C.c_void_p),
        ("c1", C.c_void_p), ("c2b", C.c_void_p), ("bproj", C.c_void_p),
        ("bias", C.c_void_p), ("labels", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
    ]


class Conv64Params(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("upsample2x", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("act", C.c_int32), ("act_slope", C.c_float), ("alpha", C.c_float),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("out_nchw_f32", C.c_int32), ("n_valid", C.c_int32),
    ]


class Conv128OutParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("gn_table", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("alpha", C.c_float),
        ("out", C.c_void_p), ("n_valid", C.c_int32),
    ]


class GnParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("groups", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("sums", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("eps", C.c_float), ("silu", C.c_int32),
        ("y", C.c_void_p), ("ldy", C.c_int32),
        ("sums_zeroed", C.c_int32),
        ("partial", C.c_void_p), ("tiles_per_image", C.c_int32),
    ]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (building nothing: see edtr_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; import it FIRST so that libedtr_hip.so binds to the same (already
    # loaded) HIP runtime instead of pulling a second one from /opt/rocm ("no ROCm-capable device" otherwise).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the EDTR MI355X path has no CPU/PyTorch fallback. "
            "Build it with `python -m edtr_amd.build` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.edtr_abi_version.restype = i32
    lib.edtr_error_string.restype = C.c_char_p
    lib.edtr_error_string.argtypes = [i32]
    lib.edtr_device_info.argtypes = [C.POINTER(i32), C.POINTER(i64), C.c_char_p, i32]
    lib.edtr_igemm.argtypes = [C.POINTER(IgemmParams), vp]
    lib.edtr_igemm_plan.argtypes = [C.POINTER(IgemmParams)]
    lib.edtr_flash_attn64.argtypes = [C.POINTER(AttnParams), vp]
    lib.edtr_flash_attn512.argtypes = [C.POINTER(AttnParams), vp]
    lib.edtr_gn_stats.argtypes = [C.POINTER(GnParams), vp]
    lib.edtr_gn_apply.argtypes = [C.POINTER(GnParams), vp]
    lib.edtr_gn_finalize.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.edtr_gn_table.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, vp, f32, vp, vp]
    lib.edtr_layernorm.argtypes = [i32, vp, i64, i32, i32, i32, vp, vp, f32, vp, i32, vp]
    lib.edtr_softmax_rows.argtypes = [i32, vp, i64, i32, i64, vp, i64, i32, vp]
    lib.edtr_nchw_to_nhwc.argtypes = [i32, vp, i32, i32, i64, vp, i32, i32, i32, f32, f32, vp]
    lib.edtr_nhwc_to_nchw.argtypes = [i32, vp, i32, i32, i32, i64, i32, vp, f32, vp]
    lib.edtr_add.argtypes = [i32, vp, i32, vp, i32, vp, i32, i64, i32, vp]
    lib.edtr_add_mirror.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i64, i32, vp]
    lib.edtr_timestep_embedding.argtypes = [i32, vp, i32, i32, vp, i32, vp]
    lib.edtr_sampler_update.argtypes = [vp, vp, vp, f32, f32, f32, f32, f32, vp, vp, i64, vp]
    lib.edtr_axpby.argtypes = [vp, vp, f32, f32, vp, i64, vp]
    lib.edtr_q_sample.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, i64, vp]
    lib.edtr_split3.argtypes = [i32, vp, i64, i32, i64, i32, vp, i64, vp]
    lib.edtr_cast16.argtypes = [i32, vp, i64, i32, i64, vp, i64, vp]
    lib.edtr_split_operand.argtypes = [i32, vp, i64, i32, i64, i32, vp, i64, vp]
    lib.edtr_sampler_update_indexed.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, i32, i64, vp]
    lib.edtr_gaussian_sample.argtypes = [vp, i32, vp, vp, i32, i32, i64, f32, vp]
    lib.edtr_tile_accumulate.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.edtr_divide.argtypes = [vp, vp, vp, i64, vp]
    lib.edtr_wavelet_level.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.edtr_gn_pool.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.edtr_copy3d_f32.argtypes = [vp, i64, i64, vp, i64, i64, i32, i32, i32, vp]
    lib.edtr_graph_begin.argtypes = [vp]
    lib.edtr_graph_end.argtypes = [vp, C.POINTER(vp)]
    lib.edtr_graph_launch.argtypes = [vp, vp]
    lib.edtr_graph_destroy.argtypes = [vp]
    for name in DECLARED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("edtr_error_string",):
            fn.restype = i32
    lib.edtr_zero_bytes.argtypes = [vp, i64, vp]
    lib.edtr_embed_tokens.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]
    lib.edtr_window_attn.argtypes = [C.POINTER(WindowAttnParams), vp]
    lib.edtr_pixel_unshuffle.argtypes = [i32, vp, i32, i32, i32, i32, i32, vp, f32, vp, i32, i32, vp]
    lib.edtr_swin_mlp.argtypes = [C.POINTER(SwinMlpParams), vp]
    lib.edtr_swin_attn.argtypes = [C.POINTER(SwinAttnParams), vp]
    lib.edtr_conv64.argtypes = [C.POINTER(Conv64Params), vp]
    lib.edtr_conv128_out.argtypes = [C.POINTER(Conv128OutParams), vp]
    lib.edtr_swin_layer.argtypes = [C.POINTER(SwinAttnParams), C.POINTER(SwinMlpParams), vp]
    lib.edtr_add_stats.argtypes = [i32, vp, i32, vp, i32, vp, i32, i64, i32, vp, i32, i32, vp]
    lib.edtr_ffn.argtypes = [C.POINTER(FfnParams), vp]
    lib.edtr_ffn_plan.argtypes = [C.POINTER(FfnParams)]
    lib.edtr_lin320.argtypes = [C.POINTER(Lin320Params), vp]
    lib.edtr_lin320_plan.argtypes = [C.POINTER(Lin320Params)]
    if lib.edtr_abi_version() != 10:
        raise RuntimeError("libedtr_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().edtr_error_string(code).decode()
        raise RuntimeError(f"libedtr_hip {what} failed with code {code}: {msg}")
