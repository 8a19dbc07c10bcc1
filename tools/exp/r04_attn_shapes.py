#!/usr/bin/env python3
"""Round 4: timing of edtr_flash_attn64 on the eight attention shapes of a denoise step (BASELINE configs[1]: batch 8, SD-2.1
widths) in the PIPELINE's operand layout — q / k of a self-attention are the two halves of one [M, 2C] projection, the
cross-attention keys are a column slice of the per-net [B * 77, sumC] matrix, V^T is key-major — torch-free (tools/hipfree.py).
Prints microseconds per launch (back-to-back launches, HIP events), TFLOP/s and the algorithmic GB/s.

    python3 tools/exp/r04_attn_shapes.py [B]        # EDTR_ATTN_* environment switches select kernels (A/B = two runs in one call)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hipfree as H  # noqa: E402
from hipfree import C, L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SUMC = 12480       # UNet: 5 x 320 + 5 x 640 + 6 x 1280 columns of cross-attention keys
LEVELS = [(4096, 320, 5, 7), (1024, 640, 10, 7), (256, 1280, 20, 7), (64, 1280, 20, 2)]     # tokens, channels, heads, layers per step (ControlNet + UNet)


def time_case(dt, N, Cc, heads, cross, rng):
    Nk = 77 if cross else N
    ldv = (Nk + 7) // 8 * 8
    p = L.AttnParams()
    p.dtype, p.B, p.H, p.Nq, p.Nk = dt, B, heads, N, Nk
    if cross:
        q = H.Dev(H.rand16(rng, (B * N, Cc), dt, 0.42))
        k = H.Dev(H.rand16(rng, (B * 77, SUMC), dt, 0.42))
        vt = H.Dev(H.rand16(rng, (B * SUMC, ldv), dt))
        p.q, p.q_bs, p.q_ld = q.p, N * Cc, Cc
        p.k, p.k_bs, p.k_ld = k.p, 77 * SUMC, SUMC
        p.vt, p.vt_bs, p.vt_ld = vt.p, SUMC * ldv, ldv
        keep = (q, k, vt)
    else:
        qk = H.Dev(H.rand16(rng, (B * N, 2 * Cc), dt, 0.42))
        vt = H.Dev(H.rand16(rng, (B * Cc, ldv), dt))
        p.q, p.q_bs, p.q_ld = qk.p, N * 2 * Cc, 2 * Cc
        p.k, p.k_bs, p.k_ld = C.c_void_p((qk.p.value or 0) + 2 * Cc), N * 2 * Cc, 2 * Cc
        p.vt, p.vt_bs, p.vt_ld = vt.p, Cc * ldv, ldv
        keep = (qk, vt)
    o = H.Dev(nbytes=B * N * Cc * 2, fill=0)
    p.out, p.o_bs, p.o_ld = o.p, N * Cc, Cc
    p.scale, p.causal, p.q_prescaled = 0.125, 0, 1
    ms = H.time_launches([lambda s: H.chk(H.edtr.edtr_flash_attn64(C.byref(p), s), "flash_attn64")], iters=30, warm=3)
    del keep
    fl = 4.0 * B * heads * N * Nk * 64
    nb = 2.0 * B * heads * 64 * (2 * N + 2 * Nk)
    return ms, fl, nb


def main():
    rng = np.random.default_rng(0)
    sw = {k: v for k, v in os.environ.items() if k.startswith("EDTR_ATTN")}
    tot = 0.0
    print(f"# B={B} bf16, switches {sw}")
    for cross in (False, True):
        for N, Cc, heads, layers in LEVELS:
            ms, fl, nb = time_case(0, N, Cc, heads, cross, rng)
            tot += ms * layers * 4
            print(f"{'cross' if cross else 'self '} N={N:5d} C={Cc:5d} H={heads:3d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {nb / ms / 1e6:7.1f} GB/s   "
                  f"x {layers * 4} launches per pass = {ms * layers * 4:6.3f} ms", flush=True)
    print(f"sum per pass (4 steps): {tot:.3f} ms")


if __name__ == "__main__":
    main()
