"""Round 6: the register-staged 64 x 64 tile (edtr_igemm tile 2) against the LDS-DMA 128 x 128 loop (tile 3, with and without split-K) on the
transformer blocks' square linears whose 128-row grids leave most of the chip idle.  torch-free (tools/hipfree.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hipfree as H  # noqa: E402
from hipfree import L  # noqa: E402
from hw_ab_tiles import run  # noqa: E402


def main():
    dt = L.BF16
    rng = np.random.default_rng(0)
    shapes = [(512, 1280, 1280), (1024, 1280, 1280), (2048, 1280, 1280), (256, 1280, 1280), (4096, 640, 640), (8192, 640, 640), (16384, 320, 320),
              (2048, 1280, 5120), (512, 1280, 5120)]
    for M, N, K in shapes:
        a = H.Dev(H.rand16(rng, (M, K), dt))
        w = H.Dev(H.rand16(rng, (N, K), dt, 1.0 / np.sqrt(K)))
        bias = H.Dev(rng.standard_normal(N).astype(np.float32))
        res = H.Dev(H.rand16(rng, (M, N), dt))
        row = []
        ref = None
        for tile, sk in [(3, 1), (2, 1), (1, 1), (3, 2), (3, 3), (2, 2)]:
            if sk > (K + 63) // 64:
                continue
            got, _, ms, rc = run(dt, tile, M=M, N=N, K=K, taps=1, spatial=None, C1=K, a=a, w=w, bias=bias, res=res, splitk=sk)
            if rc != 0:
                row.append(f"t{tile}/sk{sk}: rc {rc}")
                continue
            if ref is None:
                ref = H.from16(got, dt)
            err = float(np.abs(H.from16(got, dt) - ref).max() / (np.abs(ref).max() + 1e-30))
            row.append(f"t{tile}/sk{sk}: {ms * 1e3:6.1f} us (err {err:.1e})")
        print(f"M{M:6d} N{N:5d} K{K:5d}  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
