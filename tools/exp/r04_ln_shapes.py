#!/usr/bin/env python3
"""Round 4: timing of edtr_layernorm on the transformer-block shapes of a denoise step (batch 8, bf16), torch-free.
EDTR_LN_ROWS_PER_WAVE=1 is the one-row-per-wave form (A/B = two runs in one call); also checks that both forms agree bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hipfree as H  # noqa: E402
from hipfree import C, L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(B * 4096, 320, 84), (B * 1024, 640, 84), (B * 256, 1280, 84), (B * 64, 1280, 24), (B * 4096, 192, 0)]


def main():
    rng = np.random.default_rng(0)
    tot = 0.0
    print(f"# B={B} bf16, EDTR_LN_ROWS_PER_WAVE={os.environ.get('EDTR_LN_ROWS_PER_WAVE', 'auto')}")
    for rows, Cc, per_pass in SHAPES:
        x = H.Dev(H.rand16(rng, (rows, Cc), 0))
        y = H.Dev(nbytes=rows * Cc * 2, fill=0)
        g = H.Dev(np.ones(Cc, np.float32) * 1.5)
        b = H.Dev(np.full(Cc, 0.25, np.float32))
        ms = H.time_launches([lambda s: H.chk(H.edtr.edtr_layernorm(0, x.p, rows, Cc, Cc, Cc, g.p, b.p, 1e-5, y.p, Cc, s), "layernorm")], iters=30, warm=3)
        out = y.get(np.uint16, (rows, Cc))
        tot += ms * per_pass
        print(f"rows {rows:6d} C {Cc:5d}: {ms * 1e3:7.1f} us  {4.0 * rows * Cc / ms / 1e6:7.1f} GB/s  x {per_pass} per pass = {ms * per_pass:6.3f} ms   checksum {int(out.astype(np.uint64).sum())}", flush=True)
    print(f"sum per pass: {tot:.3f} ms")


if __name__ == "__main__":
    main()
