"""edtr_lin320 against the launches it replaces (edtr_igemm, and edtr_layernorm + edtr_igemm) at the bench shapes: microseconds per launch,
back to back (HIP events via torch), rotating over 8 buffer sets so that the operands do not sit in the caches from the launch before."""
import math
import sys

import torch

sys.path.insert(0, ".")
from edtr_amd import ops  # noqa: E402


def timed(recs_sets, n=40):
    for rs in recs_sets:
        for r in rs:
            ops.launch(r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        for r in recs_sets[i % len(recs_sets)]:
            ops.launch(r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    d = torch.device("cuda:0")
    dtype = torch.bfloat16
    K = 320
    g = torch.Generator().manual_seed(0)
    for M in (32768, 16384):
        for N, ln, res in ((320, False, True), (320, True, False), (320, False, False), (960, True, False)):
            w = torch.randn((N, K), generator=g) / math.sqrt(K)
            wi = ops.pack_linear_weight(w, dtype).to(d)
            wl = ops.pack_lin320_w(w, dtype).to(d)
            gamma, beta = torch.ones(K, device=d), torch.zeros(K, device=d)
            cvec = torch.randn(N, generator=g).to(d)
            old, new = [], []
            for _ in range(8):
                x = torch.randn((M, K), generator=g).to(dtype).to(d)
                r = torch.randn((M, N), generator=g).to(dtype).to(d) if res else None
                o1, o2, xn = torch.empty((M, N), dtype=dtype, device=d), torch.empty((M, N), dtype=dtype, device=d), torch.empty_like(x)
                rs = []
                if ln:
                    rs.append(ops.make_layernorm(dtype=dtype, x=x, rows=M, C=K, ldx=K, gamma=gamma, beta=beta, eps=1e-5, y=xn, ldy=K))
                rs.append(ops.make_igemm(dtype=dtype, a1=xn if ln else x, w=wi, out=o1, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, alpha=0.5, bias_n=cvec,
                                         residual=r, ldr=N))
                old.append(rs)
                new.append([ops.make_lin320(dtype=dtype, x=x, ldx=K, M=M, N=N, w=wl, cvec=cvec, alpha=0.5, ln=ln, eps=1e-5, residual=r, ldr=N, out=o2, ldo=N)])
            t_old, t_new = timed(old), timed(new)
            fl = 2.0 * M * N * K
            print(f"M {M} N {N} ln {int(ln)} res {int(res)}: igemm form {t_old:6.1f} us | lin320 {t_new:6.1f} us = {fl / t_new / 1e6:5.0f} TFLOP/s | x{t_old / t_new:.2f}", flush=True)


if __name__ == "__main__":
    main()
