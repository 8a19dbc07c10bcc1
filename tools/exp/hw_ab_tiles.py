#!/usr/bin/env python3
"""torch-free A/B of two edtr_igemm tile variants on ONE device: same random operands through tile A (reference, default 3) and
tile B, outputs compared (bit-exact expected when both walk K in the same order, e.g. 3 vs 11; otherwise a tolerance), both timed.

    python3 tools/exp/hw_ab_tiles.py --a 3 --b 11 [--tol 0] [--dt bf16]
    python3 tools/exp/hw_ab_tiles.py --a 3 --b 14 --tol 4e-3      # other MFMA shape: not bit-exact

Cases cover 1..40 K-tiles, M / N tails, 3x3 convs with halo (stride 1, nearest-x2), residual + bias + GroupNorm partials and
split-K, i.e. every path the variant shares with tile 3.  Writes gpurun_out/hw_ab_tiles.json."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hipfree as H  # noqa: E402
from hipfree import C, L  # noqa: E402

GEMMS = [(256, 128, 64), (300, 72, 192), (4096, 320, 320), (130, 136, 128), (77, 640, 1024), (8, 1280, 320), (2048, 1280, 1280),
         (512, 1280, 1280), (8192, 640, 640), (2048, 1280, 2560)]
HALO = [(2, 64, 64, 128, 128, 0), (1, 32, 32, 64, 128, 0), (2, 16, 48, 64, 128, 0), (1, 16, 16, 128, 256, 0), (3, 32, 16, 192, 320, 0),
        (8, 512, 512, 128, 128, 0), (8, 512, 512, 256, 128, 0), (8, 256, 256, 256, 256, 0), (8, 128, 128, 512, 512, 0), (8, 64, 64, 512, 512, 0),
        (8, 64, 64, 320, 320, 0), (8, 32, 32, 640, 640, 0), (8, 16, 16, 1280, 1280, 0)]
CONVS = [(8, 512, 512, 128, 8, 0), (2, 64, 64, 128, 8, 0), (1, 32, 32, 64, 64, 0), (2, 16, 24, 64, 96, 0), (1, 32, 32, 128, 128, 0), (2, 8, 8, 1280, 1280, 0), (1, 16, 16, 64, 64, 1), (8, 16, 16, 1280, 1280, 0)]


def run(dt, tile, *, M, N, K, taps, spatial, C1, a, w, bias, res, splitk=1, gnp=False):
    out = H.Dev(nbytes=M * N * 2, fill=0xFF)
    p = L.IgemmParams()
    p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = dt, taps, M, N, K, 1, 1
    p.a1, p.C1, p.ld1, p.w, p.ldw = a.p, C1, C1, w.p, K
    if spatial:
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l, p.upsample2x = spatial
    p.alpha, p.bias_n, p.residual, p.ldr = 1.0, bias.p, (None if os.environ.get('AB_NORES') else res.p), N
    p.out, p.ldc, p.tile, p.splitk = out.p, N, tile, splitk
    keep = [out]
    if splitk > 1:
        ws = H.Dev(nbytes=splitk * M * N * 4)
        p.workspace, p.workspace_bytes = ws.p, splitk * M * N * 4
        keep.append(ws)
    g = None
    if gnp:
        g = H.Dev(nbytes=(M // 128) * N * 2 * 4, fill=0)
        p.gn_partial = g.p
    code = H.edtr.edtr_igemm(C.byref(p), H.stream())
    if code != 0 and not H.DRY:
        return None, None, None, code
    ms = H.time_launches([lambda s: H.chk(H.edtr.edtr_igemm(C.byref(p), s), "igemm")], iters=10, warm=2)
    got = out.get(np.uint16, (M, N))
    gp = g.get(np.float32, ((M // 128), N, 2)) if gnp else None
    return got, gp, ms, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", type=int, default=3)
    ap.add_argument("--b", type=int, default=14)
    ap.add_argument("--tol", type=float, default=0.0, help="0 = bit-exact; else max |a-b| / max|a|")
    ap.add_argument("--dt", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--cases", default="all", choices=["all", "halo", "smallm", "halo160", "b4", "up2", "smallgemm", "halo512", "halo512big", "halo160b", "bign"], help="halo: stride-1 3x3 convs on 16-pixel-aligned images only")
    args = ap.parse_args()
    dt = 0 if args.dt == "bf16" else 1
    rng = np.random.default_rng(0)
    rows, bad = [], 0
    cases = [("gemm", g, 1) for g in GEMMS] + [("gemm", (2048, 1280, 5120), 3), ("gemm", (512, 1280, 1280), 2)]
    cases += [("conv", c, 1) for c in CONVS] + [("conv", (8, 8, 8, 1280, 1280, 0), 6)]
    if args.cases == "smallm":      # the 8x8-latent convolutions: which split-K count?
        cases = [("conv", (8, 8, 8, 1280, 1280, 0), k) for k in (4, 5, 6, 8, 10, 12, 20)] + [("conv", (8, 8, 8, 2560, 1280, 0), k) for k in (5, 6, 8, 10, 13, 20)]
        cases += [("conv", (4, 8, 8, 1280, 1280, 0), k) for k in (6, 10, 20)] + [("conv", (4, 8, 8, 2560, 1280, 0), k) for k in (10, 20)]
    if args.cases == "smallgemm":   # plain GEMMs with at most ~one 128x128 tile per CU
        cases = [("gemm", g, 1) for g in [(2048, 1280, 1280), (1024, 1280, 1280), (512, 1280, 1280), (256, 1280, 1280), (2048, 2560, 1280), (2048, 1280, 2560),
                                           (2048, 1280, 5120), (4096, 640, 640), (4096, 640, 2560), (8192, 640, 640), (1000, 520, 1152), (130, 136, 128), (77, 640, 1024),
                                           (300, 72, 192), (2048, 1280, 320)]]
        cases += [("gemm", (2048, 1280, 5120), 2), ("gemm", (1024, 1280, 5120), 3), ("gemm", (512, 1280, 1280), 2)]
    if args.cases == "bign":        # the wide-N linears (GEGLU projection, fused qkv) as plain GEMMs: which tile would they want?
        cases = [("gemm", g, 1) for g in [(32768, 2560, 320), (8192, 5120, 640), (2048, 10240, 1280), (32768, 960, 320), (8192, 1920, 640), (2048, 3840, 1280),
                                           (32768, 1280, 320), (8192, 2560, 640), (2048, 5120, 1280)]]
    if args.cases == "up2":         # nearest-2x upsample convolutions (VAE decoder / UNet Upsample)
        cases = [("conv", c, 1) for c in [(2, 16, 16, 64, 128, 1), (1, 24, 8, 128, 256, 1), (8, 256, 256, 256, 256, 1), (8, 128, 128, 512, 512, 1),
                                           (8, 64, 64, 512, 512, 1), (8, 32, 32, 640, 640, 1), (8, 16, 16, 1280, 1280, 1), (8, 8, 8, 1280, 1280, 1)]]
    if args.cases == "b4":          # batch-4 UNet shapes (det512s50): is the halo tile's 96-unit threshold right?
        cases = [("conv", c, 1) for c in [(4, 32, 32, 640, 640, 0), (4, 32, 32, 1280, 640, 0), (4, 32, 32, 960, 640, 0), (4, 64, 64, 128, 128, 0), (4, 64, 64, 512, 512, 0),
                                           (1, 64, 64, 512, 512, 0), (1, 128, 128, 512, 512, 0), (2, 64, 64, 512, 512, 0)]]
        cases += [("conv", (4, 16, 16, 1280, 1280, 0), k) for k in (3, 4, 6)] + [("conv", (4, 16, 16, 2560, 1280, 0), k) for k in (3, 6)]
    if args.cases == "halo160":     # N = 320: the 160-column halo variant against the 128x160 tile
        cases = [("conv", c, 1) for c in [(8, 64, 64, 320, 320, 0), (8, 64, 64, 640, 320, 0), (8, 64, 64, 960, 320, 0), (2, 32, 48, 128, 320, 0),
                                           (1, 16, 16, 64, 160, 0), (8, 32, 32, 320, 960, 0)]] + [("conv", (2, 32, 32, 640, 320, 0), 2)]
    if args.cases == "halo512":     # tile 17 (32 x 16-pixel units, 32-channel chunks): borders, 1 / 3 / 5 chunks, several column tiles
        cases = [("conv", c, 1) for c in [(2, 16, 32, 32, 128, 0), (1, 32, 64, 96, 128, 0), (3, 16, 64, 160, 256, 0), (2, 48, 32, 64, 384, 0), (1, 64, 64, 128, 128, 0),
                                           (8, 64, 64, 512, 512, 0), (2, 128, 128, 256, 256, 0)]]
    if args.cases == "halo160b":    # tile 20 (16 x 16 pixels x 160 channels): borders, 1 / 3 / 10 chunks, one / two / four column tiles, the bench shapes
        cases = [("conv", c, 1) for c in [(2, 16, 16, 32, 160, 0), (1, 32, 48, 96, 320, 0), (3, 16, 32, 64, 640, 0), (8, 64, 64, 320, 320, 0), (8, 64, 64, 640, 320, 0),
                                           (8, 64, 64, 960, 320, 0), (4, 64, 64, 320, 320, 0), (8, 32, 32, 640, 640, 0)]]
    if args.cases == "halo512big":  # the VAE's ResnetBlock convolutions at batch 8
        cases = [("conv", c, 1) for c in [(8, 512, 512, 128, 128, 0), (8, 512, 512, 256, 128, 0), (8, 256, 256, 256, 256, 0), (8, 256, 256, 512, 256, 0),
                                           (8, 128, 128, 512, 512, 0), (8, 64, 64, 512, 512, 0)]]
    if args.cases == "halo":
        cases = [("conv", c, 1) for c in HALO] + [("conv", (8, 16, 16, 1280, 1280, 0), 3), ("conv", (8, 16, 16, 2560, 1280, 0), 3),
                                                  ("conv", (8, 16, 16, 1920, 1280, 0), 3), ("conv", (8, 16, 16, 1280, 1280, 0), 2),
                                                  ("conv", (8, 16, 16, 1280, 1280, 0), 4), ("conv", (2, 32, 32, 640, 640, 0), 3)]
    for kind, shp, splitk in cases:
        if kind == "gemm":
            M, N, K = shp
            taps, spatial, C1, rows_in = 1, None, K, M
        else:
            B, Hh, Ww, Cin, Cout, up = shp
            OH, OW = (Hh * 2, Ww * 2) if up else (Hh, Ww)
            M, N, K, taps, C1, rows_in = B * OH * OW, Cout, 9 * Cin, 9, Cin, B * Hh * Ww
            spatial = (Hh, Ww, OH, OW, 1, 1, 1, up)
        a = H.Dev(H.rand16(rng, (rows_in, C1), dt))
        w = H.Dev(H.rand16(rng, (N, K), dt, 1.0 / np.sqrt(K)))
        bias = H.Dev(rng.standard_normal(N, dtype=np.float32))
        res = H.Dev(H.rand16(rng, (M, N), dt))
        gnp = splitk == 1 and M % 128 == 0 and N % 32 == 0 and not os.environ.get('AB_NORES')
        kw = dict(M=M, N=N, K=K, taps=taps, spatial=spatial, C1=C1, a=a, w=w, bias=bias, res=res, splitk=splitk, gnp=gnp)
        ya, ga, ta, ca = run(dt, args.a, **kw)
        yb, gb, tb, cb = run(dt, args.b, **kw)
        name = f"{kind}{shp} sk{splitk}{' gnp' if gnp else ''}"
        if ca or cb:
            print(f"SKIP  {name}: tile {args.a} rc {ca}, tile {args.b} rc {cb}", flush=True)
            rows.append({"case": name, "rc": [ca, cb]})
            continue
        fa, fb = H.from16(ya, dt), H.from16(yb, dt)
        err = float(np.abs(fa - fb).max() / max(np.abs(fa).max(), 1e-30))
        if gnp and kind == "conv":      # tiles may place their per-tile partials differently: compare the per-image sums
            ga, gb = (x.reshape(shp[0], -1, N, 2).astype(np.float64).sum(1) for x in (ga, gb))
        gerr = float(np.abs(ga - gb).max() / max(np.abs(ga).max(), 1e-30)) if gnp else 0.0
        ok = (np.array_equal(ya, yb) if args.tol == 0 else err <= args.tol) and (gerr <= max(args.tol, 1e-6))
        bad += not ok
        print(f"{'PASS' if ok else 'FAIL'}  {name:44s} err {err:.2e} gnp {gerr:.1e}   tile {args.a}: {ta * 1e3:8.1f} us   tile {args.b}: {tb * 1e3:8.1f} us  ({ta / tb:4.2f}x)", flush=True)
        rows.append({"case": name, "ok": bool(ok), "err": err, "us_a": ta * 1e3, "us_b": tb * 1e3})
    os.makedirs(os.path.join(H.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(H.ROOT, "gpurun_out", f"hw_ab_tiles_{args.a}_vs_{args.b}_{args.dt}.json"), "w") as f:
        json.dump({"a": args.a, "b": args.b, "rows": rows}, f, indent=1)
    print("ALL PASS" if not bad else f"{bad} FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
