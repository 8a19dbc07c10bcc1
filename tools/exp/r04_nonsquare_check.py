#!/usr/bin/env python3
"""Round 4 sanity run: non-square and odd-multiple image sizes through the full path in every precision mode (the sub-pixel upsample
geometry, the fp16 mirrors and the split attention all have shape conditions with fallbacks): finite results, and each mode within
its usual distance of precision="high"."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from edtr_amd import synth, workloads  # noqa: E402
from edtr_amd.diffusion import Diffusion  # noqa: E402
from edtr_amd.sampler import SpacedSampler  # noqa: E402
from edtr_amd.testing import build_synthetic_cldm, err_stats, injected_noise  # noqa: E402

d = torch.device("cuda:0")
cfg = synth.sd21_config()
diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(d)
sampler = SpacedSampler(diffusion.betas)
bad = 0
for (B, H, W) in [(2, 384, 512), (1, 320, 448), (3, 256, 256), (1, 512, 768)]:
    pre = synth.synth_input("ns:pre", (B, 3, H, W), 0.0, 1.0).to(d)
    c_txt = synth.synth_normal("ns:c_txt", (1, 77, 1024)).expand(B, -1, -1).contiguous().to(d)
    noises = [synth.synth_normal(f"ns:noise{i}", (B, 4, H // 8, W // 8)).to(d) for i in range(5)]
    res = {}
    for mode, dt in (("high", None), ("mixed", None), ("fast", torch.bfloat16), ("fast", torch.float16)):
        cldm = build_synthetic_cldm(cfg, d, dt, precision=mode)
        z_pre = cldm.vae_encode(pre * 2 - 1, sample=False)
        x_T = diffusion.q_sample(z_pre, torch.full((B,), 200, dtype=torch.int64), noises[0])
        with injected_noise(noises[1:]):
            z = sampler.manual_sample_with_timesteps(model=cldm, device=d, x_T=x_T, steps=4, used_timesteps=[50, 100, 150, 200], batch_size=B,
                                                     cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False)
        img = cldm.vae_decode(z)
        torch.cuda.synchronize()
        res[(mode, dt)] = img.float().cpu()
        ok = bool(torch.isfinite(img).all())
        bad += not ok
        tags = sorted({r.tag.split(" ")[-1] for e in cldm._vae_engines.values() for r in e.prog.recs if "up2" in r.tag})
        line = f"B{B} {H}x{W} {mode:5s} {str(dt):15s} finite={ok}  upsample forms {tags}"
        if mode != "high":
            st = err_stats(res[(mode, dt)], res[("high", None)])
            lim = {"mixed": 1e-3, "fast": 2.5e-3 if dt == torch.float16 else 2e-2}[mode]
            line += f"  vs high: l2 {st['l2']:.2e} max {st['max']:.2e}"
            bad += not (st["l2"] < lim)
        print(line, flush=True)
        cldm.release_engines()
        del cldm
print("ALL OK" if not bad else f"{bad} PROBLEM(S)")
sys.exit(1 if bad else 0)
