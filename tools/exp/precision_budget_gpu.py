#!/usr/bin/env python3
"""Per-layer-class precision sensitivities of the mixed mode, measured ON the MI355X with the product kernels.

For BASELINE configs[1] at full size (images 3 and 7 of the bench batch, run as a batch of 2 like
tests/test_gpu_precision.py::test_det512_full_size_meets_the_north_star) against the REFERENCE's output
(tests/golden/full_det512.npz):

  base           every GEMM class at 3 parts (~22-bit operands)            -> e_base
  class c at p   class c lowered to p in {1, 2} parts, the rest at 3        -> e(c, p)
  var(c, p) = e(c, p)^2 - e_base^2   (independent roundings add in quadrature)

plus the three constant policies.  Writes gpurun_out/r03/precision_sensitivity.json; tools/exp/precision_allocate.py turns it
(with the per-class launch times of `bench.py --precision mixed --breakdown-json`) into the shipped allocation.

    python tools/exp/precision_budget_gpu.py [--classes a,b,...] [--out path]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03", "precision_sensitivity.json"))
    ap.add_argument("--levels", action="store_true", help="also split the VAE convolutions per resolution level (name@M)")
    args = ap.parse_args()
    from edtr_amd import synth, workloads
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.precision import ConstPolicy, PrecisionPolicy
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, rel_err
    d = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_det512.npz"))
    cldm = build_synthetic_cldm(synth.sd21_config(), d, precision="mixed")
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(d)
    sampler = SpacedSampler(diffusion.betas)
    full = workloads.make_inputs("det512", 1024, d, 8, 512)
    sel = [int(k) for k in g["images"]]
    inp = workloads.Inputs(full.pre_res[sel].contiguous(), full.c_txt[sel].contiguous(), [n[sel].contiguous() for n in full.noises], [],
                           full.t_start[:len(sel)])

    def run(policy):
        cldm.precision_policy = policy
        img, z, tr = workloads.restore_pass(cldm, diffusion, sampler, inp, "det512")
        torch.cuda.synchronize()
        return {"z_pre": rel_err(tr["z_pre"], g["z_pre"]), "z": rel_err(z, g["z"]),
                "img": rel_err(img[:, :, 1::4, 2::4], g["img_samples"].astype(np.float32))}

    out = {"note": "relative L2 vs tests/golden/full_det512.npz (images 3, 7); img golden is stored as fp16 samples (floor ~2.8e-4)",
           "const": {}, "classes": {}}
    t0 = time.time()
    for p in (3, 2, 1):
        out["const"][str(p)] = run(ConstPolicy(p))
        print(f"const {p}: {out['const'][str(p)]}  [{time.time() - t0:.0f}s]", flush=True)
    # the GEMM classes of the programs just built (names of the igemm launches) with their shapes
    names = {}
    for eng in list(cldm._cldm_engines.values()) + list(cldm._vae_engines.values()):
        for prog in [getattr(eng, "step_prog", None), getattr(eng, "ctx_prog", None), getattr(eng, "prog", None)]:
            if prog is None:
                continue
            for r in prog.recs:
                if r.tag and r.tag.startswith("taps"):
                    m = int(r.tag.split(" M")[1].split()[0])
                    names.setdefault(r.name, set()).add(m)
    classes = sorted(names)
    if args.classes:
        classes = [c for c in args.classes.split(",") if c]
    keys = []
    for c in classes:
        keys.append(c)
        if args.levels and c in ("vae.conv1", "vae.conv2") and c in names:
            keys += [f"{c}@{m}" for m in sorted(names[c])]
    out["class_shapes"] = {c: sorted(names.get(c, [])) for c in classes}
    for key in keys:
        ent = {}
        for p in (1, 2):
            ent[str(p)] = run(PrecisionPolicy(3, {key: p}, f"{key}={p}"))
        out["classes"][key] = ent
        print(f"{key:28s} p1 {ent['1']}  p2 {ent['2']}  [{time.time() - t0:.0f}s]", flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    print("written", args.out)


if __name__ == "__main__":
    main()
