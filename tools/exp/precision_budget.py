#!/usr/bin/env python3
"""Where does the 16-bit error come from?  Runs the REFERENCE modules (CPU fp32, /root/reference) at SD-2.1 widths with
roundings injected by hooks, and reports the relative L2 error of eps / vae_z / vae_dec against the unrounded run:

  W        weights rounded to the 16-bit type, activations fp32
  W+A      + every Conv2d / Linear INPUT rounded (MFMA operands are 16-bit), everything else fp32  (= fp32 residual stream)
  W+A+O    + every Conv2d / Linear / GroupNorm / LayerNorm OUTPUT rounded                      (= 16-bit storage everywhere)
  W+A+attn + q, k, v and the probabilities rounded inside attention
  A        inputs only (weights fp32)

Build container only (tools/, never shipped to the GPU box)."""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from edtr_amd import synth  # noqa: E402
import ref_import  # noqa: E402
from make_goldens import build_reference_cldm  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    torch.set_num_threads(os.cpu_count())
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else "sd21"
    cldm, cfg = build_reference_cldm(cfg_name)
    ctx = cfg["unet_cfg"]["context_dim"]
    hw = 64 if cfg_name == "sd21" else 16
    x = synth.synth_normal("sd21:x", (1, 4, hw, hw))
    c_img = synth.synth_normal("sd21:c_img", (1, 4, hw, hw))
    c_txt = synth.synth_input("sd21:c_txt", (1, 77, ctx), -1.0, 1.0)
    t = torch.tensor([200], dtype=torch.int64)
    img = synth.synth_input("sd21:img", (1, 3, 256, 256), -1.0, 1.0)
    zin = synth.synth_normal("sd21:zdec", (1, 4, 32, 32))

    def run():
        with torch.no_grad():
            return (cldm(x, t, {"c_txt": c_txt, "c_img": c_img}), cldm.vae_encode(img, sample=False), cldm.vae_decode(zin))

    ref = run()
    fp32_w = {k: v.clone() for k, v in cldm.state_dict().items()}
    mods = [m for m in cldm.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
    norms = [m for m in cldm.modules() if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm))]
    orig_sdpa = F.scaled_dot_product_attention

    for dt in (torch.float16, torch.bfloat16):
        rnd = lambda v: v.to(dt).float()   # noqa: E731

        def set_weights(rounded):
            with torch.no_grad():
                for m in mods:
                    m.weight.copy_(rnd(fp32_w_of[m][0]) if rounded else fp32_w_of[m][0])
        fp32_w_of = {m: (m.weight.detach().clone(),) for m in mods}

        def sdpa_rounded(q, k, v, *a, **kw):
            q, k, v = rnd(q), rnd(k), rnd(v)
            s = (q @ k.transpose(-1, -2)) / np.sqrt(q.shape[-1])
            p = torch.softmax(s, dim=-1)
            # the kernel rounds the UNNORMALISED probabilities (relative rounding, same error) and divides at the end
            return rnd(p) @ v

        def experiment(name, w, a_in, a_out, attn):
            set_weights(w)
            hs = []
            if a_in:
                hs += [m.register_forward_pre_hook(lambda m_, i: (rnd(i[0]),) + tuple(i[1:])) for m in mods]
            if a_out:
                hs += [m.register_forward_hook(lambda m_, i, o: rnd(o)) for m in mods + norms]
            if attn:
                F.scaled_dot_product_attention = sdpa_rounded
            try:
                out = run()
            finally:
                for h in hs:
                    h.remove()
                F.scaled_dot_product_attention = orig_sdpa
                set_weights(False)
            print(f"{str(dt):16s} {name:10s} eps {rel(out[0], ref[0]):.2e}  vae_z {rel(out[1], ref[1]):.2e}  vae_dec {rel(out[2], ref[2]):.2e}", flush=True)

        experiment("W", True, False, False, False)
        experiment("A", False, True, False, False)
        experiment("W+A", True, True, False, False)
        experiment("W+A+attn", True, True, False, True)
        experiment("W+A+O", True, True, True, False)
        experiment("W+A+O+attn", True, True, True, True)


if __name__ == "__main__":
    with contextlib.redirect_stderr(io.StringIO()):
        main()
