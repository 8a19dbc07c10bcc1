import sys, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from edtr_amd import ops
d = torch.device("cuda:0"); dt = torch.bfloat16
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for B, HW, C, zeroed in [(2, 1024, 320, False), (3, 1024, 320, False), (3, 1024, 320, True), (4, 1024, 320, True), (8, 1024, 320, True), (3, 4096, 320, True), (3, 256, 1280, True), (5, 64, 1280, True)]:
    x = torch.randn(B, HW, C, generator=torch.Generator().manual_seed(1)).to(dt)
    x[1:] = x[:1]            # identical images
    xg = x.reshape(B * HW, C).to(d)
    y = torch.empty_like(xg)
    pool = torch.zeros((4, B, 32, 2), dtype=torch.float64, device=d)
    sums = pool[2]
    g = torch.ones(C, device=d); b = torch.zeros(C, device=d)
    st, ap = ops.make_gn(dtype=dt, x=xg, ldx=C, B=B, HW=HW, C=C, sums=sums, gamma=g, beta=b, eps=1e-5, silu=True, y=y, ldy=C, sums_zeroed=zeroed)
    ops.launch(st); ops.launch(ap); torch.cuda.synchronize()
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1).reshape(B, C, HW), 32, eps=1e-5)).reshape(B, C, HW).permute(0, 2, 1)
    yy = y.float().cpu().reshape(B, HW, C)
    print(f"B={B} HW={HW} C={C} zeroed={zeroed}: err " + " ".join(f"{rel(yy[i], ref[i]):.1e}" for i in range(B)), " sums row0", sums[0, 0].tolist(), "last", sums[B - 1, 0].tolist(), flush=True)
