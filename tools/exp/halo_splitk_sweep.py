#!/usr/bin/env python3
"""torch-free sweep of the split-K count of the halo tile (16) on the UNet's 3x3 convolutions whose unit count does not fill the
256 CUs at batch 8 / 4 (32x32 latents: 32 / 16 patches x 5 column tiles = 160 / 80 units; 16x16: 8 / 4 x 10).
    python3 tools/exp/halo_splitk_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hipfree as H  # noqa: E402
from hipfree import C, L  # noqa: E402


def run(tile, B, Hh, Ww, Cin, Cout, sk, rng):
    M, N, K = B * Hh * Ww, Cout, 9 * Cin
    a = H.Dev(H.rand16(rng, (M, Cin), 0))
    w = H.Dev(H.rand16(rng, (N, K), 0, 1.0 / np.sqrt(K)))
    bias = H.Dev(rng.standard_normal(N, dtype=np.float32))
    res = H.Dev(H.rand16(rng, (M, N), 0))
    out = H.Dev(nbytes=M * N * 2, fill=0)
    p = L.IgemmParams()
    p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = 0, 9, M, N, K, 1, 1
    p.a1, p.C1, p.ld1, p.w, p.ldw = a.p, Cin, Cin, w.p, K
    p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l, p.upsample2x = Hh, Ww, Hh, Ww, 1, 1, 1, 0
    p.alpha, p.bias_n, p.residual, p.ldr = 1.0, bias.p, res.p, N
    p.out, p.ldc, p.tile, p.splitk = out.p, N, tile, sk
    keep = [a, w, bias, res, out]
    if sk > 1:
        ws = H.Dev(nbytes=sk * M * N * 4)
        p.workspace, p.workspace_bytes = ws.p, sk * M * N * 4
        keep.append(ws)
    code = H.edtr.edtr_igemm(C.byref(p), H.stream())
    if code != 0:
        return None
    return H.time_launches([lambda s: H.chk(H.edtr.edtr_igemm(C.byref(p), s), "igemm")], iters=10, warm=2) * 1e3


def main():
    rng = np.random.default_rng(0)
    if "--n320" in sys.argv:      # the 64x64-latent convolutions with 320 output channels: 128x160 tile against the halo tile's ragged third column tile
        for shp in [(8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 64, 64, 960, 320), (4, 64, 64, 320, 320), (4, 64, 64, 640, 320), (4, 64, 64, 960, 320),
                    (2, 64, 64, 320, 320), (1, 64, 64, 320, 320)]:
            B, Hh, Ww, Cin, Cout = shp
            flops = 2.0 * B * Hh * Ww * Cout * 9 * Cin
            row = []
            for tile, sk in ((8, 1), (16, 1), (16, 2)):
                us = run(tile, B, Hh, Ww, Cin, Cout, sk, rng)
                row.append(f"t{tile}/sk{sk} " + (f"{us:6.1f}us {flops / us / 1e6:5.0f}TF" if us else "   n/a"))
            print(f"conv{shp} halo units {(B * Hh * Ww // 256) * 3:4d}: " + " | ".join(row), flush=True)
        return
    shapes = [(8, 32, 32, 640, 640), (8, 32, 32, 1280, 640), (8, 32, 32, 1920, 640), (8, 32, 32, 320, 640), (8, 32, 32, 960, 640),
              (4, 32, 32, 640, 640), (4, 32, 32, 1280, 640), (4, 64, 64, 320, 320), (8, 16, 16, 1280, 1280), (8, 16, 16, 2560, 1280),
              (4, 16, 16, 1280, 1280), (4, 16, 16, 2560, 1280)]
    for shp in shapes:
        B, Hh, Ww, Cin, Cout = shp
        flops = 2.0 * B * Hh * Ww * Cout * 9 * Cin
        row = []
        for tile, sk in ((8, 1), (16, 1), (16, 2), (16, 3), (16, 4), (16, 5), (16, 6)):
            if sk > Cin // 64:
                continue
            us = run(tile, B, Hh, Ww, Cin, Cout, sk, rng)
            row.append(f"t{tile}/sk{sk} " + (f"{us:6.1f}us {flops / us / 1e6:5.0f}TF" if us else "   n/a"))
        units = (B * Hh * Ww // 256) * ((Cout + 127) // 128)
        print(f"conv{shp} units {units:4d}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
