#!/usr/bin/env python3
"""torch-free kernel micro-benchmarks (numpy + ctypes, see tools/hipfree.py): one `gpurun` call of ~20 s instead of
minutes, so tile / split-K / shape A-Bs can be run by the dozen.

    python3 tools/hipbench.py CASE [CASE ...]   [--iters N] [--rotate R] [--dt bf16|fp16] [--json PATH]

CASE is one comma-separated spec (no spaces):
    conv,B,H,W,Cin,Cout[,key=value...]     3x3 stride-1 pad-1 conv; keys: tile, splitk, up2=1, gnp=1, res=1, act, slope
    gemm,M,N,K[,key=value...]              linear; keys: tile, splitk, act, slope, res=1, Z
    attn,B,heads,Nq,Nk[,causal=1]          edtr_flash_attn64
    wattn,B,H,W,heads,d[,shift=4]          edtr_window_attn
    ln,rows,C[,c_valid=...]                edtr_layernorm
    gn,B,HW,C                              edtr_gn_stats + edtr_gn_apply
    preset:vae | preset:unet | preset:swin  the hot shapes of profiles/r01/*breakdown* (batch 8)
A value list `tile=3|6|8` expands into one run per value (A/B on ONE device inside one call).
--rotate R cycles R independent buffer sets per timed round so that operands do not stay in L2 / MALL between launches
(in-pipeline behaviour); R = 1 re-reads hot buffers.
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hipfree as H  # noqa: E402
from hipfree import C, L  # noqa: E402

PRESETS = {
    "vae": ["conv,8,512,512,128,128,gnp=1", "conv,8,256,256,256,256,gnp=1", "conv,8,128,128,512,512,gnp=1", "conv,8,64,64,512,512,gnp=1",
            "conv,8,256,256,256,256,up2=1,gnp=1", "conv,8,512,512,256,128,gnp=1"],
    "unet": ["conv,8,64,64,320,320,gnp=1", "conv,8,32,32,640,640,gnp=1", "conv,8,16,16,1280,1280,splitk=3", "conv,8,8,8,1280,1280,splitk=6",
             "gemm,32768,320,320", "gemm,8192,640,640", "gemm,2048,1280,1280", "gemm,32768,2560,320,act=1", "gemm,8192,5120,640,act=1",
             "gemm,2048,10240,1280,act=1", "gemm,2048,1280,5120,splitk=3", "attn,8,5,4096,4096", "attn,8,10,1024,1024", "attn,8,5,4096,77"],
    "swin": ["gemm,32768,576,192", "gemm,32768,192,192,res=1", "gemm,32768,384,192,act=3", "gemm,32768,192,384,res=1", "wattn,8,64,64,6,30,shift=4",
             "ln,32768,192,c_valid=180", "conv,8,64,64,192,192,res=1", "conv,8,512,512,64,64,act=4,slope=0.2"],
}


def parse_case(spec: str):
    parts = spec.split(",")
    kind, pos, kw = parts[0], [], {}
    for p in parts[1:]:
        if "=" in p:
            k, v = p.split("=", 1)
            kw[k] = v.split("|")
        else:
            pos.append(int(p))
    keys = list(kw)
    for combo in itertools.product(*(kw[k] for k in keys)) if keys else [()]:
        yield kind, pos, {k: (float(v) if k == "slope" else int(v)) for k, v in zip(keys, combo)}


class Case:
    def __init__(self, name, launch, flops=0.0, nbytes=0.0):
        self.name, self.launch, self.flops, self.bytes = name, launch, flops, nbytes


def make_igemm(rng, dt, *, M, N, K, taps, spatial, kw, C1):
    """One buffer set + a launcher for edtr_igemm."""
    rows_in = M if not spatial else (M // (spatial[2] * spatial[3])) * spatial[0] * spatial[1]
    a = H.Dev(H.rand16(rng, (rows_in, C1), dt))
    w = H.Dev(H.rand16(rng, (N, K), dt, 1.0 / np.sqrt(K)))
    act = kw.get("act", 0)
    n_out = N // 2 if act == 1 else N
    out = H.Dev(nbytes=M * n_out * 2)
    bias = H.Dev(np.zeros(N, np.float32))
    p = L.IgemmParams()
    p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = dt, taps, M, N, K, kw.get("Z", 1), 1
    p.a1, p.C1, p.ld1, p.w, p.ldw = a.p, C1, C1, w.p, K
    if spatial:
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l, p.upsample2x = spatial
    p.alpha, p.bias_n, p.act, p.act_slope = 1.0, bias.p, act, kw.get("slope", 0.0)
    p.out, p.ldc, p.tile, p.splitk = out.p, n_out, kw.get("tile", 0), kw.get("splitk", 1)
    keep = [a, w, out, bias]
    if kw.get("res"):
        r = H.Dev(H.rand16(rng, (M, n_out), dt))
        p.residual, p.ldr = r.p, n_out
        keep.append(r)
    if p.splitk > 1:
        ws = H.Dev(nbytes=p.splitk * M * N * 4)
        p.workspace, p.workspace_bytes = ws.p, p.splitk * M * N * 4
        keep.append(ws)
    if kw.get("gnp"):
        g = H.Dev(nbytes=(M // 128) * N * 2 * 4)
        p.gn_partial = g.p
        keep.append(g)

    def launch(s, p=p, keep=keep):
        H.chk(H.edtr.edtr_igemm(C.byref(p), s), "edtr_igemm")
    return launch, 2.0 * M * N * K * p.Z, 2.0 * (rows_in * C1 + N * K + M * n_out * (2 if kw.get("res") else 1))


def build(kind, pos, kw, dt, rng, rotate):
    tag = ",".join([kind] + [str(v) for v in pos] + [f"{k}={v}" for k, v in kw.items()])
    launchers, flops, nbytes = [], 0.0, 0.0
    for _ in range(rotate):
        if kind == "conv":
            B, Hh, Ww, Cin, Cout = pos
            up = kw.get("up2", 0)
            OH, OW = (Hh * 2, Ww * 2) if up else (Hh, Ww)
            fn, flops, nbytes = make_igemm(rng, dt, M=B * OH * OW, N=Cout, K=9 * Cin, taps=9, spatial=(Hh, Ww, OH, OW, 1, 1, 1, up), kw=kw, C1=Cin)
        elif kind == "gemm":
            M, N, K = pos
            fn, flops, nbytes = make_igemm(rng, dt, M=M, N=N, K=K, taps=1, spatial=None, kw=kw, C1=K)
        elif kind == "attn":
            B, heads, Nq, Nk = pos
            Cc = heads * 64
            ldv = (Nk + 7) // 8 * 8
            q, k = H.Dev(H.rand16(rng, (B * Nq, Cc), dt)), H.Dev(H.rand16(rng, (B * Nk, Cc), dt))
            vt_host = np.zeros((B * Cc, ldv), np.uint16)
            vt_host[:, :Nk] = H.rand16(rng, (B * Cc, Nk), dt)
            vt, o = H.Dev(vt_host), H.Dev(nbytes=B * Nq * Cc * 2)
            p = L.AttnParams()
            p.dtype, p.B, p.H, p.Nq, p.Nk = dt, B, heads, Nq, Nk
            p.q, p.q_bs, p.q_ld, p.k, p.k_bs, p.k_ld = q.p, Nq * Cc, Cc, k.p, Nk * Cc, Cc
            p.vt, p.vt_bs, p.vt_ld, p.out, p.o_bs, p.o_ld = vt.p, Cc * ldv, ldv, o.p, Nq * Cc, Cc
            p.scale, p.causal = 0.125, kw.get("causal", 0)

            def fn(s, p=p, keep=(q, k, vt, o)):
                H.chk(H.edtr.edtr_flash_attn64(C.byref(p), s), "flash_attn64")
            flops, nbytes = 4.0 * B * heads * Nq * Nk * 64, 2.0 * Cc * B * (2 * Nq + 2 * Nk)
        elif kind == "wattn":
            B, Hh, Ww, heads, d = pos
            shift = kw.get("shift", 0)
            host = np.zeros((B * Hh * Ww, 3, heads, 32), np.uint16)
            host[..., :d] = H.rand16(rng, (B * Hh * Ww, 3, heads, d), dt)
            cp = (heads * d + 63) // 64 * 64
            qkv, o = H.Dev(host), H.Dev(nbytes=B * Hh * Ww * cp * 2)
            bias = H.Dev(rng.standard_normal((heads, 64, 64), dtype=np.float32))
            i = np.arange(Hh)
            j = np.arange(Ww)
            lab = (np.where(i < Hh - 8, 0, np.where(i < Hh - shift, 1, 2))[:, None] * 3 + np.where(j < Ww - 8, 0, np.where(j < Ww - shift, 1, 2))[None, :]).astype(np.uint8)
            labd = H.Dev(lab)
            p = L.WindowAttnParams()
            p.dtype, p.B, p.H, p.W, p.heads, p.head_dim, p.shift = dt, B, Hh, Ww, heads, d, shift
            p.qkv, p.ld_qkv, p.out, p.ld_out, p.c_pad = qkv.p, 3 * heads * 32, o.p, cp, cp
            p.bias, p.labels, p.scale = bias.p, (labd.p if shift else None), d ** -0.5

            def fn(s, p=p, keep=(qkv, o, bias, labd)):
                H.chk(H.edtr.edtr_window_attn(C.byref(p), s), "window_attn")
            flops, nbytes = 4.0 * B * Hh * Ww * 64 * heads * d, 2.0 * B * Hh * Ww * (3 * heads * 32 + heads * d)
        elif kind == "ln":
            rows, Cc = pos
            x, y = H.Dev(H.rand16(rng, (rows, Cc), dt)), H.Dev(nbytes=rows * Cc * 2)
            g, b = H.Dev(np.ones(Cc, np.float32)), H.Dev(np.zeros(Cc, np.float32))

            def fn(s, a=(dt, x.p, rows, Cc, kw.get("c_valid", 0), Cc, g.p, b.p, 1e-5, y.p, Cc), keep=(x, y, g, b)):
                H.chk(H.edtr.edtr_layernorm(*a, s), "layernorm")
            nbytes = 4.0 * rows * Cc
        elif kind == "gn":
            B, HW, Cc = pos
            x, y = H.Dev(H.rand16(rng, (B * HW, Cc), dt)), H.Dev(nbytes=B * HW * Cc * 2)
            g, b, sums = H.Dev(np.ones(Cc, np.float32)), H.Dev(np.zeros(Cc, np.float32)), H.Dev(nbytes=B * 32 * 2 * 8)
            p = L.GnParams()
            p.dtype, p.B, p.HW, p.C, p.groups = dt, B, HW, Cc, 32
            p.x, p.ldx, p.sums, p.gamma, p.beta, p.eps, p.silu, p.y, p.ldy = x.p, Cc, sums.p, g.p, b.p, 1e-5, 1, y.p, Cc

            def fn(s, p=p, keep=(x, y, g, b, sums)):
                H.chk(H.edtr.edtr_gn_stats(C.byref(p), s), "gn_stats")
                H.chk(H.edtr.edtr_gn_apply(C.byref(p), s), "gn_apply")
            nbytes = 6.0 * B * HW * Cc
        else:
            raise SystemExit(f"unknown case kind {kind!r}")
        launchers.append(fn)
    return tag, launchers, flops, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rotate", type=int, default=1)
    ap.add_argument("--dt", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--json", default=os.path.join(H.ROOT, "gpurun_out", "hipbench.json"))
    args = ap.parse_args()
    dt = 0 if args.dt == "bf16" else 1
    rng = np.random.default_rng(0)
    specs = []
    for c in args.cases:
        specs += PRESETS[c.split(":", 1)[1]] if c.startswith("preset:") else [c]
    rows = []
    t0 = time.time()
    for spec in specs:
        for kind, pos, kw in parse_case(spec):
            try:
                tag, launchers, flops, nbytes = build(kind, pos, kw, dt, rng, args.rotate)
                ms = H.time_launches(launchers, iters=args.iters)
                row = {"case": tag, "us": ms * 1e3, "tflops": flops / ms / 1e9 if flops else 0.0, "gbps": nbytes / ms / 1e6}
                print(f"{tag:58s} {row['us']:9.1f} us  {row['tflops']:8.1f} TFLOP/s  {row['gbps']:8.1f} GB/s", flush=True)
            except Exception as e:  # one bad case must not waste the GPU call
                row = {"case": spec, "error": repr(e)}
                print(f"{spec}: ERROR {e!r}", flush=True)
            rows.append(row)
    print(f"total {time.time() - t0:.1f} s")
    os.makedirs(os.path.dirname(args.json), exist_ok=True)
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
