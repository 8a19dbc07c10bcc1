"""Times edtr_swin_mlp alone (HIP events, 32768 tokens = B 8 at 512^2) and the fc1 / fc2 edtr_igemm pair it replaces."""
import math
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from edtr_amd import ops, lib as L

d = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for dtype in (torch.bfloat16, torch.float16):
    CP, HP = ops.SWIN_MLP_C, ops.SWIN_MLP_HIDDEN
    g = torch.Generator().manual_seed(0)
    x = torch.randn((rows, CP), generator=g).to(dtype).to(d)
    w1g = torch.randn((HP, CP), generator=g) / math.sqrt(CP)
    w2 = torch.randn((CP, HP), generator=g) / math.sqrt(HP)
    i1, i2 = ops.pack_swin_mlp_weights(w1g, w2, dtype)
    c1 = w1g.to(dtype).float().sum(1).contiguous().to(d)
    c2b = torch.randn(HP, generator=g).to(d)
    b2 = torch.randn(CP, generator=g).to(d)
    out = torch.empty((rows, CP), dtype=dtype, device=d)
    stats = torch.empty((rows, CP // 32, 2), dtype=torch.float32, device=d)
    rec = ops.make_swin_mlp(dtype=dtype, x=x, ldx=CP, rows=rows, c_valid=180, eps=1e-5, w1=i1.to(d), w2=i2.to(d), c1=c1, c2b=c2b, b2=b2, out=out,
                            ldo=CP, row_stats=stats)
    # the pair it replaces
    w1p, w2p = w1g.to(dtype).to(d).contiguous(), w2.to(dtype).to(d).contiguous()
    hid = torch.empty((rows, HP), dtype=dtype, device=d)
    lnst = torch.zeros((rows, CP // 32, 2), dtype=torch.float32, device=d)
    xf = x.float()
    lnst[:, 0, 0], lnst[:, 0, 1] = xf.sum(1), (xf * xf).sum(1)
    r1 = ops.make_igemm(dtype=dtype, a1=x, w=w1p, out=hid, M=rows, N=HP, C1=CP, ld1=CP, ldw=CP, ldc=HP, bias_n=c2b, act=L.ACT_GELU, ln_stats=lnst,
                        ln_C=CP, ln_valid=180, ln_c1=c1, ln_c2=c2b)
    r2 = ops.make_igemm(dtype=dtype, a1=hid, w=w2p, out=out, M=rows, N=CP, C1=HP, ld1=HP, ldw=HP, ldc=CP, bias_n=b2, residual=x, ldr=CP, row_stats=stats)

    def timeit(recs, n=200):
        for _ in range(10):
            for r in recs:
                ops.launch(r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            for r in recs:
                ops.launch(r)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_f = timeit([rec])
    t_p = timeit([r1, r2])
    fl = 4.0 * rows * CP * HP
    print(f"{dtype} rows {rows}: fused {t_f:.1f} us ({fl / t_f / 1e6:.0f} TFLOP/s)   fc1 + fc2 {t_p:.1f} us ({fl / t_p / 1e6:.0f} TFLOP/s)")
