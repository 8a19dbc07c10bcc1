"""edtr_ffn at the bench shape (M = 32768) and at batch 4 (16384): microseconds per launch, back to back (HIP events via torch)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from edtr_amd import ops  # noqa: E402


def main():
    d = torch.device("cuda:0")
    dtype = torch.bfloat16
    D, H = 320, 1280
    g = torch.Generator().manual_seed(0)
    w1 = torch.randn((2 * H, D), generator=g) / math.sqrt(D)
    w2 = torch.randn((D, H), generator=g) / math.sqrt(H)
    w1p = ops.pack_linear_weight(w1[ops.geglu_perm(H)], dtype).to(d)
    w2p = ops.pack_ffn_w2(w2, dtype).to(d)
    cst = ops.pack_ffn_constants(torch.randn(2 * H, generator=g)).to(d)
    b2 = torch.zeros(D, device=d)
    for M in (32768, 16384):
        x = torch.randn((M, D), generator=g).to(dtype).to(d)
        out = torch.empty_like(x)
        rec = ops.make_ffn(dtype=dtype, x=x, ldx=D, M=M, w1=w1p, w2=w2p, cst=cst, b2=b2, out=out, ldo=D)
        for _ in range(5):
            ops.launch(rec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            ops.launch(rec)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"M {M}: {us:.1f} us per launch = {2.0 * M * D * 3 * H / us / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
