"""torch-free access to the GPU for experiments: numpy + ctypes on libamdhip64 and libedtr_hip.

Why: `import torch` dominates the cost of a short `gpurun` call (and takes minutes on a cold box), while a script built on
this module starts in about a second — an 18-second GPU charge per experiment instead of several minutes.  The C ABI has no
torch types in its signatures, so every kernel can be driven from here: device buffers (`Dev`), 16-bit conversions
(`to16` / `from16`), HIP-event timing on a private stream (`time_launches`).  `HIPFREE_DRY=1` runs the host side only (no
device), for checking a script in the build container.

Never imported by the product (edtr_amd/) or by tests; tools only.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from edtr_amd import lib as L  # noqa: E402  (struct definitions only; L.load() would import torch and is not used here)

DRY = os.environ.get("HIPFREE_DRY") == "1"
vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
edtr = C.CDLL(os.environ.get("EDTR_LIB") or L.LIB_PATH)       # EDTR_LIB: a diagnostic / variant build (tools/exp/attn_variants.py)
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipFree.argtypes = [vp]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [vp, C.c_int, C.c_size_t]
hip.hipStreamCreate.argtypes = [C.POINTER(vp)]
hip.hipStreamSynchronize.argtypes = [vp]
hip.hipEventCreate.argtypes = [C.POINTER(vp)]
hip.hipEventRecord.argtypes = [vp, vp]
hip.hipEventSynchronize.argtypes = [vp]
hip.hipEventElapsedTime.argtypes = [C.POINTER(f32), vp, vp]
edtr.edtr_error_string.restype = C.c_char_p
edtr.edtr_igemm.argtypes = [C.POINTER(L.IgemmParams), vp]
edtr.edtr_flash_attn64.argtypes = [C.POINTER(L.AttnParams), vp]
edtr.edtr_window_attn.argtypes = [C.POINTER(L.WindowAttnParams), vp]
edtr.edtr_gn_stats.argtypes = [C.POINTER(L.GnParams), vp]
edtr.edtr_gn_apply.argtypes = [C.POINTER(L.GnParams), vp]
edtr.edtr_layernorm.argtypes = [i32, vp, i64, i32, i32, i32, vp, vp, f32, vp, i32, vp]
edtr.edtr_pixel_unshuffle.argtypes = [i32, vp, i32, i32, i32, i32, i32, vp, f32, vp, i32, i32, vp]
edtr.edtr_zero_bytes.argtypes = [vp, i64, vp]


def chk(code: int, what: str = "") -> None:
    if code != 0 and not DRY:
        msg = edtr.edtr_error_string(code).decode() if code < 0 else f"hipError {code}"
        raise RuntimeError(f"{what}: {msg}")


class Dev:
    """A device buffer (hipMalloc), optionally initialised from a numpy array or a byte fill."""

    def __init__(self, arr: np.ndarray | None = None, nbytes: int = 0, fill: int | None = None):
        self.n = int(arr.nbytes if arr is not None else nbytes)
        self.p = vp()
        if DRY:
            return
        chk(hip.hipMalloc(C.byref(self.p), max(self.n, 256)), "hipMalloc")
        if arr is not None:
            a = np.ascontiguousarray(arr)
            chk(hip.hipMemcpy(self.p, a.ctypes.data_as(vp), a.nbytes, 1), "hipMemcpy H2D")
        elif fill is not None:
            chk(hip.hipMemset(self.p, fill, self.n), "hipMemset")

    def get(self, dtype, shape) -> np.ndarray:
        out = np.zeros(shape, dtype=dtype)
        if not DRY:
            chk(hip.hipDeviceSynchronize(), "sync")
            chk(hip.hipMemcpy(out.ctypes.data_as(vp), self.p, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self) -> None:
        if not DRY and self.p:
            hip.hipFree(self.p)
            self.p = vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to16(x: np.ndarray, dt: int) -> np.ndarray:
    """fp32 -> 16-bit storage bits (uint16), round to nearest even; dt 0 = bf16, 1 = fp16."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dt == 1:
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from16(b: np.ndarray, dt: int) -> np.ndarray:
    if dt == 1:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)


def rand16(rng: np.random.Generator, shape, dt: int, scale: float = 1.0) -> np.ndarray:
    """Random 16-bit storage bits with N(0, scale^2) values.  Large buffers repeat a 1 Mi-element random block (host-side
    generation would otherwise dominate a short GPU call); values, not their arrangement, are what the kernels' timing sees."""
    n = int(np.prod(shape))
    block = to16(rng.standard_normal(min(n, 1 << 20), dtype=np.float32) * np.float32(scale), dt)
    return np.resize(block, n).reshape(shape)


_stream = None


def stream() -> vp:
    global _stream
    if _stream is None:
        _stream = vp()
        if not DRY:
            chk(hip.hipStreamCreate(C.byref(_stream)), "hipStreamCreate")
    return _stream


def time_launches(launchers, iters: int = 20, warm: int = 3) -> float:
    """Average milliseconds of one round of `launchers` (callables taking the stream) over `iters` rounds, HIP events on
    the launch stream.  Several launchers = buffer rotation (each round touches every buffer set once)."""
    if DRY:
        for fn in launchers:
            fn(None)
        return float("nan")
    s = stream()
    for _ in range(warm):
        for fn in launchers:
            fn(s)
    e0, e1 = vp(), vp()
    chk(hip.hipEventCreate(C.byref(e0)), "eventCreate")
    chk(hip.hipEventCreate(C.byref(e1)), "eventCreate")
    chk(hip.hipEventRecord(e0, s), "eventRecord")
    for _ in range(iters):
        for fn in launchers:
            fn(s)
    chk(hip.hipEventRecord(e1, s), "eventRecord")
    chk(hip.hipEventSynchronize(e1), "eventSynchronize")
    ms = f32(0)
    chk(hip.hipEventElapsedTime(C.byref(ms), e0, e1), "elapsed")
    return ms.value / (iters * len(launchers))
