import sys, math, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from edtr_amd import ops
d = torch.device("cuda:0")
dtype = torch.bfloat16
def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale
def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())
def check(B, H, cin, cout, tile, splitk=1, gnp=False, rowvec=True):
    W = H
    x = rnd((B, cin, H, W), 1).to(dtype); w = rnd((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin)).to(dtype); bias = rnd((cout,), 3)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    x16 = x.permute(0, 2, 3, 1).contiguous().to(d)
    wp = ops.pack_conv_weight(w.float(), dtype).to(d)
    N = wp.shape[0]
    M = B * H * W
    emb = rnd((B, N), 4).to(d) if rowvec else None
    out = torch.empty((M, N), dtype=dtype, device=d)
    g = torch.zeros((M // 128, N, 2), dtype=torch.float32, device=d) if gnp else None
    ws = torch.empty(splitk * M * N, dtype=torch.float32, device=d) if splitk > 1 else None
    ops.launch(ops.make_igemm(dtype=dtype, a1=x16.reshape(M, cin), w=wp, out=out, taps=9, M=M, N=N, C1=cin, ld1=cin, ldw=wp.shape[1], ldc=N,
                              spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=ops.pad_bias(bias, N).to(d), rowvec=emb, rowvec_ld=N if rowvec else 0,
                              rows_per_image=H * W, tile=tile, splitk=splitk, workspace=ws, gn_partial=g))
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(B, H, W, N)[..., :cout].permute(0, 3, 1, 2)
    full = ref + (emb.cpu()[:, :cout, None, None] if rowvec else 0)
    e = rel(got, full)
    extra = ""
    if gnp:
        cs = g.cpu()[..., 0].reshape(B, (H * W) // 128, N).sum(1)[:, :cout]
        extra = f" gnp colsum err {rel(cs, got.double().sum((2, 3)).float()):.1e}"
    print(f"B={B} H={H} {cin}->{cout} tile{tile} sk{splitk} gnp={int(gnp)}: rel err {e:.2e}{extra}", flush=True)
for B in (2, 3, 8):
    check(B, 64, 320, 320, 0)
    check(B, 64, 320, 320, 0, gnp=True)
    check(B, 64, 320, 320, 3, gnp=True)
    check(B, 64, 320, 320, 8, gnp=True)
    check(B, 64, 640, 320, 0, gnp=True)
    check(B, 32, 640, 640, 0, gnp=True)
    check(B, 16, 1280, 1280, 0, gnp=False)
    check(B, 16, 1280, 1280, 0, splitk=3)
    check(B, 8, 1280, 1280, 0, splitk=6)
