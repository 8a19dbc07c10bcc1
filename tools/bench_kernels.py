#!/usr/bin/env python3
"""Kernel micro-benchmarks on the MI355X (run through gpurun): times individual libedtr_hip launches at the real
hot-path shapes with HIP events on the launch stream.  Usage:  python tools/bench_kernels.py [conv|gemm|attn|gn|all]"""
from __future__ import annotations

import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edtr_amd import ops  # noqa: E402

DT = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(rec, iters=20, warm=3):
    s = ops.stream_ptr()
    for _ in range(warm):
        rec.launch(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        rec.launch(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(DT)


def bench_conv(B, H, Cin, Cout, stride=1, ups=False, tile=0, extra=None):
    W = H
    x = rnd(B * H * W, Cin)
    w = rnd(Cout, 9 * Cin, scale=1 / math.sqrt(9 * Cin))
    LH = H * 2 if ups else H
    OH = (LH + 2 - 3) // stride + 1
    out = torch.empty((B * OH * OH, Cout), dtype=DT, device=dev)
    bias = torch.zeros(Cout, device=dev)
    kw = dict(extra or {})
    if kw.get("splitk", 1) > 1:
        kw["workspace"] = torch.empty(kw["splitk"] * B * OH * OH * Cout, dtype=torch.float32, device=dev)
    rec = ops.make_igemm(dtype=DT, a1=x, w=w, out=out, taps=9, M=B * OH * OH, N=Cout, C1=Cin, ld1=Cin, ldw=9 * Cin,
                         ldc=Cout, spatial=(H, W, OH, OH, stride, 1, 1, int(ups)), bias_n=bias, tile=tile, **kw)
    ms = timeit(rec)
    return ms, rec.flops / ms / 1e9


def bench_gemm(M, N, K, tile=0, act=0, residual=False, splitk=1):
    a = rnd(M, K)
    w = rnd(N, K, scale=1 / math.sqrt(K))
    n_out = N // 2 if act == 1 else N
    out = torch.empty((M, n_out), dtype=DT, device=dev)
    res = rnd(M, n_out) if residual else None
    bias = torch.zeros(N, device=dev)
    ws = torch.empty(splitk * M * N, dtype=torch.float32, device=dev) if splitk > 1 else None
    rec = ops.make_igemm(dtype=DT, a1=a, w=w, out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=n_out, bias_n=bias, act=act,
                         residual=res, ldr=n_out, tile=tile, splitk=splitk, workspace=ws)
    ms = timeit(rec)
    return ms, rec.flops / ms / 1e9


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    B = int(os.environ.get("B", "8"))
    tiles = [int(t) for t in os.environ.get("TILES", "3,6").split(",")]
    if what == "one":      # python tools/bench_kernels.py one H Cin Cout tile [stride] [ups]   (for rocprofv3 --pmc runs)
        H, ci, co, t = (int(v) for v in sys.argv[2:6])
        st = int(sys.argv[6]) if len(sys.argv) > 6 else 1
        up = bool(int(sys.argv[7])) if len(sys.argv) > 7 else False
        ms, tf = bench_conv(B, H, ci, co, stride=st, ups=up, tile=t)
        print(f"one: H={H} {ci}->{co} tile{t} s{st} up{int(up)}: {ms:.3f} ms {tf:.1f} TF")
        return
    if what == "oneattn":   # python tools/bench_kernels.py oneattn HW C NK
        hw, c, nk = (int(v) for v in sys.argv[2:5])
        H = c // 64
        q, k = rnd(B * hw, c), rnd(B * nk, c)
        ldv = ops.round_up(nk, 8)
        vt = rnd(B * c, ldv)
        out = torch.empty((B * hw, c), dtype=DT, device=dev)
        rec = ops.make_flash_attn(dtype=DT, q=q, k=k, vt=vt, out=out, B=B, H=H, Nq=hw, Nk=nk, q_bs=hw * c, q_ld=c,
                                  k_bs=nk * c, k_ld=c, vt_bs=c * ldv, vt_ld=ldv, o_bs=hw * c, o_ld=c, scale=0.125)
        ms = timeit(rec)
        print(f"oneattn N={hw} Nk={nk} heads={H}: {ms:.3f} ms {rec.flops / ms / 1e9:.1f} TF")
        return
    if what in ("conv", "all"):
        print(f"--- conv3x3 (B={B}) ---")
        shapes = [(64, 320, 320), (64, 640, 320), (64, 960, 320), (64, 640, 640), (32, 320, 640), (32, 640, 640),
                  (32, 1280, 640), (32, 1920, 640), (32, 960, 640), (16, 640, 1280), (16, 1280, 1280), (16, 2560, 1280),
                  (16, 1920, 1280), (8, 1280, 1280), (8, 2560, 1280),
                  (512, 128, 128), (256, 256, 256), (128, 512, 512), (64, 512, 512), (256, 128, 256), (128, 256, 512)]
        for H, ci, co in shapes:
            row = f"H={H:4d} {ci:5d}->{co:5d}  M={B * H * H:8d}"
            for t in tiles:
                ms, tf = bench_conv(B, H, ci, co, tile=t)
                row += f" | tile{t}: {ms:8.3f} ms {tf:7.1f} TF"
            print(row, flush=True)
        for H, ci, co, st, up in [(64, 320, 320, 2, False), (32, 640, 640, 2, False), (16, 1280, 1280, 2, False),
                                  (8, 1280, 1280, 1, True), (16, 1280, 1280, 1, True), (32, 640, 640, 1, True),
                                  (256, 512, 512, 1, False), (256, 256, 256, 1, True)]:
            row = f"H={H:4d} {ci:5d}->{co:5d} s{st} up{int(up)}"
            for t in tiles:
                ms, tf = bench_conv(B, H, ci, co, stride=st, ups=up, tile=t)
                row += f" | tile{t}: {ms:8.3f} ms {tf:7.1f} TF"
            print(row, flush=True)
    if what in ("splitk", "all"):
        print(f"--- conv3x3 split-K sweep (B={B}) ---")
        for H, ci, co in [(8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280), (16, 2560, 1280), (16, 640, 1280), (32, 640, 640)]:
            row = f"H={H:4d} {ci:5d}->{co:5d}"
            for t in (1, 3):
                for S in (1, 2, 4, 8, 16):
                    ms, tf = bench_conv(B, H, ci, co, tile=t, extra={"splitk": S})
                    row += f" | t{t}s{S}: {ms:6.3f} {tf:5.0f}"
            print(row, flush=True)
    if what in ("gsplitk",):
        print(f"--- gemm split-K sweep (B={B}) ---")
        for M, N, K in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (512, 1280, 1280), (512, 1280, 5120),
                        (512, 2560, 1280), (8192, 640, 640), (8192, 640, 2560)]:
            row = f"M={M:5d} N={N:5d} K={K:5d}"
            for t in tiles:
                for S in (1, 2, 3, 4, 6, 8):
                    if S > K // 128:
                        continue
                    ms, tf = bench_gemm(M, N, K, tile=t, residual=True, splitk=S)
                    row += f" | t{t}s{S}: {ms * 1e3:5.1f}"
            print(row, flush=True)
        print("--- conv3x3 split-K sweep, us ---")
        for H, ci, co in [(8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280), (16, 2560, 1280), (16, 640, 1280), (16, 1920, 1280)]:
            row = f"H={H:4d} {ci:5d}->{co:5d}"
            for t in tiles:
                for S in (1, 2, 3, 4, 6, 8, 12):
                    ms, tf = bench_conv(B, H, ci, co, tile=t, extra={"splitk": S})
                    row += f" | t{t}s{S}: {ms * 1e3:5.1f}"
            print(row, flush=True)
        return
    if what in ("gemm", "all"):
        print(f"--- gemm (B={B}) ---")
        for hw, c in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
            M = B * hw
            for name, N, K, act, res in [("proj/out", c, c, 0, True), ("qk", 2 * c, c, 0, False), ("geglu", 8 * c, c, 1, False),
                                         ("ff.out", c, 4 * c, 0, True)]:
                row = f"{name:9s} M={M:6d} N={N:6d} K={K:5d}"
                for t in tiles:
                    if act == 1 and t in (2, 6, 7, 8, 9, 10):
                        continue
                    ms, tf = bench_gemm(M, N, K, tile=t, act=act, residual=res)
                    row += f" | tile{t}: {ms:8.3f} ms {tf:7.1f} TF"
                print(row, flush=True)
    if what in ("attn", "all"):
        print(f"--- flash attention (B={B}) ---")
        for hw, c in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
            H = c // 64
            for nk in (hw, 77):
                q, k = rnd(B * hw, c), rnd(B * nk, c)
                ldv = ops.round_up(nk, 8)
                vt = rnd(B * c, ldv)
                out = torch.empty((B * hw, c), dtype=DT, device=dev)
                rec = ops.make_flash_attn(dtype=DT, q=q, k=k, vt=vt, out=out, B=B, H=H, Nq=hw, Nk=nk, q_bs=hw * c, q_ld=c,
                                          k_bs=nk * c, k_ld=c, vt_bs=c * ldv, vt_ld=ldv, o_bs=hw * c, o_ld=c, scale=0.125)
                ms = timeit(rec)
                print(f"N={hw:5d} Nk={nk:5d} heads={H:3d}: {ms:8.3f} ms {rec.flops / ms / 1e9:7.1f} TF", flush=True)
    if what in ("gn", "all"):
        print(f"--- groupnorm / layernorm (B={B}) ---")
        for hw, c in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1920), (256, 1280), (256, 2560), (64, 1280),
                      (64, 2560), (262144, 128), (262144, 256), (65536, 256), (65536, 512), (16384, 512), (4096, 512)]:
            x = rnd(B * hw, c)
            y = torch.empty_like(x)
            sums = torch.empty((B, 32, 2), dtype=torch.float64, device=dev)
            g = torch.ones(c, device=dev)
            st, ap = ops.make_gn(dtype=DT, x=x, ldx=c, B=B, HW=hw, C=c, sums=sums, gamma=g, beta=g, eps=1e-5, silu=True, y=y, ldy=c)
            m1, m2 = timeit(st), timeit(ap)
            nb = B * hw * c * 2
            print(f"GN HW={hw:7d} C={c:5d}: stats {m1:7.3f} ms {nb / m1 / 1e6:8.1f} GB/s | apply {m2:7.3f} ms "
                  f"{2 * nb / m2 / 1e6:8.1f} GB/s", flush=True)
        for hw, c in [(4096, 320), (1024, 640), (256, 1280)]:
            x = rnd(B * hw, c)
            y = torch.empty_like(x)
            g = torch.ones(c, device=dev)
            rec = ops.make_layernorm(dtype=DT, x=x, rows=B * hw, C=c, ldx=c, gamma=g, beta=g, eps=1e-5, y=y, ldy=c)
            ms = timeit(rec)
            print(f"LN rows={B * hw:6d} C={c:5d}: {ms:7.3f} ms {2 * B * hw * c * 2 / ms / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
