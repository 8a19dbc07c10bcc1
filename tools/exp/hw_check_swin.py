#!/usr/bin/env python3
"""Hardware check of the SwinIR kernels WITHOUT torch (numpy + ctypes on libamdhip64 / libedtr_hip): starts in about a
second on a fresh box, so it fits a very small GPU budget.  Each kernel is compared with a numpy restatement of the same
op on the same 16-bit-rounded inputs:

    edtr_pixel_unshuffle, edtr_layernorm (c_valid), edtr_igemm + EDTR_ACT_LRELU (main epilogue and split-K reducer),
    edtr_window_attn (bf16 / fp16, shift 0 / 4, non-square token grid, head width 30)

Run on the GPU box:  python3 tools/exp/hw_check_swin.py   (writes gpurun_out/hw_check_swin.json)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DRY = os.environ.get("HW_CHECK_DRY") == "1"      # no device: exercise the numpy side only (every check then FAILs by design)
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
lib = C.CDLL(os.path.join(ROOT, "edtr_amd", "libedtr_hip.so"))
vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [vp, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [vp]
lib.edtr_error_string.restype = C.c_char_p


def chk(code, what):
    if code != 0 and not DRY:
        raise RuntimeError(f"{what}: code {code} {lib.edtr_error_string(code) if code < 0 else ''}")


class Dev:
    def __init__(self, arr=None, nbytes=0, fill=None):
        self.n = arr.nbytes if arr is not None else nbytes
        self.p = vp()
        if DRY:
            return
        chk(hip.hipMalloc(C.byref(self.p), max(self.n, 16)), "hipMalloc")
        if arr is not None:
            a = np.ascontiguousarray(arr)
            chk(hip.hipMemcpy(self.p, a.ctypes.data_as(vp), a.nbytes, 1), "H2D")
        elif fill is not None:
            chk(hip.hipMemset(self.p, fill, self.n), "memset")

    def get(self, dtype, shape):
        out = np.zeros(shape, dtype=dtype)
        if DRY:
            return out
        chk(hip.hipDeviceSynchronize(), "sync")
        chk(hip.hipMemcpy(out.ctypes.data_as(vp), self.p, out.nbytes, 2), "D2H")
        return out


def to16(x, dt):
    """fp32 -> 16-bit storage bits (uint16), RNE."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dt == 1:
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from16(b, dt):
    if dt == 1:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)


class IgemmParams(C.Structure):
    _fields_ = [("dtype", i32), ("taps", i32), ("M", i32), ("N", i32), ("K", i32), ("n_valid", i32), ("Z", i32), ("zdiv", i32),
                ("a1", vp), ("a2", vp), ("C1", i32), ("C2", i32), ("ld1", i32), ("ld2", i32), ("a_zs_outer", i64), ("a_zs_inner", i64),
                ("IH", i32), ("IW", i32), ("OH", i32), ("OW", i32), ("stride", i32), ("pad_t", i32), ("pad_l", i32), ("upsample2x", i32),
                ("w", vp), ("ldw", i32), ("w_zs_outer", i64), ("w_zs_inner", i64), ("alpha", f32), ("bias_n", vp), ("bias_m", vp),
                ("rowvec", vp), ("rowvec_ld", i32), ("rows_per_image", i32), ("act", i32), ("residual", vp), ("ldr", i32),
                ("out", vp), ("ldc", i32), ("out_f32", i32), ("o_zs_outer", i64), ("o_zs_inner", i64), ("tile", i32), ("splitk", i32),
                ("workspace", vp), ("workspace_bytes", i64), ("gn_partial", vp), ("act_slope", f32)]


class WindowAttnParams(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("heads", i32), ("head_dim", i32), ("shift", i32),
                ("qkv", vp), ("ld_qkv", i32), ("out", vp), ("ld_out", i32), ("c_pad", i32), ("bias", vp), ("labels", vp), ("scale", f32)]


lib.edtr_igemm.argtypes = [C.POINTER(IgemmParams), vp]
lib.edtr_window_attn.argtypes = [C.POINTER(WindowAttnParams), vp]
lib.edtr_layernorm.argtypes = [i32, vp, i64, i32, i32, i32, vp, vp, f32, vp, i32, vp]
lib.edtr_pixel_unshuffle.argtypes = [i32, vp, i32, i32, i32, i32, i32, vp, f32, vp, i32, i32, vp]

results = {}
rng = np.random.default_rng(0)


def report(name, err, tol, extra=""):
    ok = bool(np.isfinite(err) and err <= tol)
    results[name] = {"err": float(err), "tol": tol, "ok": ok}
    print(f"{'PASS' if ok else 'FAIL'}  {name}: err {err:.3e} (tol {tol:.1e}) {extra}", flush=True)


def check_unshuffle(dt):
    B, Cc, H, W, r = 2, 3, 64, 96, 8
    x = rng.random((B, Cc, H, W), dtype=np.float32)
    sub = np.array([0.4488, 0.4371, 0.4040], dtype=np.float32)
    ld = 200                                    # 192 real + 8 pad columns
    d_x, d_sub = Dev(x), Dev(sub)
    d_out = Dev(nbytes=B * (H // r) * (W // r) * ld * 2, fill=0xFF)
    chk(lib.edtr_pixel_unshuffle(dt, d_x.p, B, Cc, H, W, r, d_sub.p, 1.0, d_out.p, ld, ld, None), "pixel_unshuffle")
    got = from16(d_out.get(np.uint16, (B, H // r, W // r, ld)), dt)
    ref = (x - sub[None, :, None, None]).reshape(B, Cc, H // r, r, W // r, r).transpose(0, 2, 4, 1, 3, 5).reshape(B, H // r, W // r, Cc * r * r)
    ref = from16(to16(ref, dt), dt)
    err = max(np.abs(got[..., :192] - ref).max(), np.abs(got[..., 192:]).max())
    report(f"pixel_unshuffle dt{dt}", err, 0.0)


def check_layernorm(dt):
    rows, Cp, Cv = 1000, 192, 180
    x = rng.standard_normal((rows, Cp)).astype(np.float32) * 2 + 0.5
    x[:, Cv:] = 7.0                              # garbage in the pad columns must not matter
    xb = to16(x, dt)
    g = np.zeros(Cp, np.float32); b = np.zeros(Cp, np.float32)
    g[:Cv] = 1 + 0.1 * rng.standard_normal(Cv); b[:Cv] = 0.1 * rng.standard_normal(Cv)
    d_x, d_g, d_b = Dev(xb), Dev(g), Dev(b)
    d_y = Dev(nbytes=rows * Cp * 2, fill=0xFF)
    chk(lib.edtr_layernorm(dt, d_x.p, rows, Cp, Cv, Cp, d_g.p, d_b.p, 1e-5, d_y.p, Cp, None), "layernorm")
    got = from16(d_y.get(np.uint16, (rows, Cp)), dt)
    xr = from16(xb, dt)[:, :Cv].astype(np.float64)
    ref = (xr - xr.mean(1, keepdims=True)) / np.sqrt(xr.var(1, keepdims=True) + 1e-5) * g[:Cv] + b[:Cv]
    err = max(np.abs(got[:, :Cv] - ref).max(), np.abs(got[:, Cv:]).max())
    report(f"layernorm c_valid dt{dt}", err, 4e-2 if dt == 0 else 5e-3)


def check_lrelu(dt, M, N, K, splitk):
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    ab, wb = to16(a, dt), to16(w, dt)
    d_a, d_w, d_b = Dev(ab), Dev(wb), Dev(bias)
    d_o = Dev(nbytes=M * N * 2, fill=0xFF)
    p = IgemmParams()
    p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = dt, 1, M, N, K, 1, 1
    p.a1, p.C1, p.ld1, p.w, p.ldw = d_a.p, K, K, d_w.p, K
    p.alpha, p.bias_n, p.act, p.act_slope = 1.0, d_b.p, 4, 0.2
    p.out, p.ldc, p.splitk = d_o.p, N, splitk
    ws = None
    if splitk > 1:
        ws = Dev(nbytes=splitk * M * N * 4)
        p.workspace, p.workspace_bytes = ws.p, splitk * M * N * 4
    chk(lib.edtr_igemm(C.byref(p), None), "igemm lrelu")
    got = from16(d_o.get(np.uint16, (M, N)), dt)
    y = from16(ab, dt).astype(np.float64) @ from16(wb, dt).astype(np.float64).T + bias
    ref = np.where(y > 0, y, 0.2 * y)
    report(f"igemm lrelu dt{dt} M{M} N{N} K{K} sk{splitk}", np.abs(got - ref).max(), 3e-2 if dt == 0 else 4e-3)


def region_labels(H, W, ws, shift):
    def band(n):
        i = np.arange(n)
        return np.where(i < n - ws, 0, np.where(i < n - shift, 1, 2))
    return (band(H)[:, None] * 3 + band(W)[None, :]).astype(np.uint8)


def window_attn_ref(qkv, bias, lab, shift, d):
    """numpy restatement of reference model/swinir.py:254-279 + :120-148 (without qkv / proj): qkv [B, H, W, 3, heads, 32]
    -> [B, H, W, heads*d]."""
    B, H, W, _, heads, HPAD = qkv.shape
    x = qkv.astype(np.float64)
    if shift:
        x = np.roll(x, (-shift, -shift), (1, 2))
    win = x.reshape(B, H // 8, 8, W // 8, 8, 3, heads, HPAD).transpose(0, 1, 3, 2, 4, 5, 6, 7).reshape(B, (H // 8) * (W // 8), 64, 3, heads, HPAD)
    q, k, v = (win[:, :, :, s].transpose(0, 1, 3, 2, 4) for s in range(3))          # [B, nW, heads, 64, 32]
    att = q @ k.transpose(0, 1, 2, 4, 3) * d ** -0.5 + bias[None, None]
    if shift:
        lw = lab.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        att = att + np.where(lw[:, None, :] != lw[:, :, None], -100.0, 0.0)[None, :, None]
    att = np.exp(att - att.max(-1, keepdims=True))
    att /= att.sum(-1, keepdims=True)
    o = (att @ v)[..., :d].transpose(0, 1, 3, 2, 4).reshape(B, H // 8, W // 8, 8, 8, heads * d)
    o = o.transpose(0, 1, 3, 2, 4, 5).reshape(B, H, W, heads * d)
    if shift:
        o = np.roll(o, (shift, shift), (1, 2))
    return o


def check_window_attn(dt, shift, B=2, H=16, W=24, heads=6, d=30):
    HPAD, cp = 32, 192
    ld = 3 * heads * HPAD
    qkv = np.zeros((B * H * W, 3, heads, HPAD), np.float32)
    qkv[..., :d] = rng.standard_normal((B * H * W, 3, heads, d)).astype(np.float32) * 1.5
    qb = to16(qkv.reshape(B * H * W, ld), dt)
    bias = (rng.standard_normal((heads, 64, 64)) * 0.7).astype(np.float32)
    lab = region_labels(H, W, 8, shift) if shift else None
    d_q, d_bias = Dev(qb), Dev(bias)
    d_lab = Dev(lab) if shift else None
    d_o = Dev(nbytes=B * H * W * cp * 2, fill=0xFF)
    p = WindowAttnParams()
    p.dtype, p.B, p.H, p.W, p.heads, p.head_dim, p.shift = dt, B, H, W, heads, d, shift
    p.qkv, p.ld_qkv, p.out, p.ld_out, p.c_pad = d_q.p, ld, d_o.p, cp, cp
    p.bias, p.labels, p.scale = d_bias.p, (d_lab.p if shift else None), d ** -0.5
    chk(lib.edtr_window_attn(C.byref(p), None), "window_attn")
    got = from16(d_o.get(np.uint16, (B, H, W, cp)), dt)
    o = window_attn_ref(from16(qb, dt).reshape(B, H, W, 3, heads, HPAD), bias, lab, shift, d)
    err = np.abs(got[..., :heads * d] - o).max() / np.abs(o).max()
    pad = np.abs(got[..., heads * d:]).max()
    report(f"window_attn dt{dt} shift{shift}", max(err, pad), 2e-2 if dt == 0 else 3e-3, f"(pad max {pad})")


def main():
    t0 = time.time()
    n = i32(0)
    chk(hip.hipGetDeviceCount(C.byref(n)), "hipGetDeviceCount")
    print("devices", n.value, "abi", lib.edtr_abi_version(), flush=True)
    tests = []
    for dt in (0, 1):
        tests += [lambda dt=dt: check_unshuffle(dt), lambda dt=dt: check_layernorm(dt),
                  lambda dt=dt: check_lrelu(dt, 256, 64, 192, 1), lambda dt=dt: check_lrelu(dt, 128, 64, 1536, 2),
                  lambda dt=dt: check_window_attn(dt, 0), lambda dt=dt: check_window_attn(dt, 4)]
    for t in tests:
        try:
            t()
        except Exception as e:  # keep going: one shot on the hardware
            print("ERROR", repr(e), flush=True)
            results[f"error{len(results)}"] = {"ok": False, "err": repr(e)}
    results["seconds"] = time.time() - t0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "hw_check_swin.json"), "w") as f:
        json.dump(results, f, indent=1)
    bad = [k for k, v in results.items() if isinstance(v, dict) and not v.get("ok", True)]
    print("ALL PASS" if not bad else f"FAILED: {bad}", f"({results['seconds']:.1f} s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
