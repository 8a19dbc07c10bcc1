#!/usr/bin/env python3
"""Timing of edtr_gn_apply (+SiLU) on the bench's GroupNorm shapes (HIP events, 20 launches each).
    python3 tools/exp/gn_apply_time.py                      # power-of-two fast path where it applies
    EDTR_GN_APPLY_GENERIC=1 python3 tools/exp/gn_apply_time.py   # the generic loop everywhere (A/B on one device)
Also checks the result against torch group_norm + silu on one shape per width."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from edtr_amd import ops

d = torch.device("cuda:0")
dt = torch.bfloat16
SHAPES = [(8, 512 * 512, 128), (8, 256 * 256, 256), (8, 128 * 128, 512), (8, 64 * 64, 512), (8, 64 * 64, 320), (8, 64 * 64, 640),
          (8, 32 * 32, 640), (8, 32 * 32, 1280), (8, 16 * 16, 1280), (8, 16 * 16, 2560), (8, 32 * 32, 960), (8, 64 * 64, 960)]
tot = 0.0
for B, HW, C in SHAPES:
    x = torch.randn(B * HW, C, device=d, generator=torch.Generator(device=d).manual_seed(1)).to(dt)
    y = torch.empty_like(x)
    sums = torch.zeros((B, 32, 2), dtype=torch.float64, device=d)
    g = torch.rand(C, device=d) + 0.5
    b = torch.randn(C, device=d)
    st, ap = ops.make_gn(dtype=dt, x=x, ldx=C, B=B, HW=HW, C=C, sums=sums, gamma=g, beta=b, eps=1e-5, silu=True, y=y, ldy=C, sums_zeroed=True)
    ops.launch(st)
    for _ in range(3):
        ops.launch(ap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.launch(ap)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    err = ""
    if HW <= 64 * 64:
        xr = x.float().reshape(B, HW, C).permute(0, 2, 1)
        ref = F.silu(F.group_norm(xr, 32, g, b, eps=1e-5)).permute(0, 2, 1).reshape(B * HW, C)
        err = f"  rel err {float((y.float() - ref).norm() / ref.norm()):.2e}"
    print(f"B={B} HW={HW:7d} C={C:5d}: {us:8.1f} us  {4.0 * B * HW * C / us / 1e6:6.2f} TB/s{err}", flush=True)
print(f"sum {tot:.1f} us")
