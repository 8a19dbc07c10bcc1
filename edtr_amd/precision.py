"""Per-layer operand precision of the mixed mode (`ControlLDM.precision = "mixed"`, EDTR_AMD_PRECISION=mixed).

The reference computes in fp32 (its GPU path: fp16 autocast with fp32 GroupNorm / LayerNorm / softmax and an fp32 VAE and
sampler, main/det/test_edtr.py:95-136, model/util.py:161-163).  16-bit MFMA operands cannot reproduce it to the north-star
1e-3: every fp16 rounding of a weight or of an activation that enters a GEMM costs ~2^-12 relative, and the ~250 GEMMs /
convolutions in series add up to 1.5e-3 on the decoded image.  The matrix cores have no wider operand, so
precision is bought with MORE PRODUCTS over the same fp16 MFMA kernel (edtr_igemm, fp32 accumulation, K = parts * C):

    parts 1   x16 . W16                      one fp16 rounding of the activation and one of the weight
    parts 2   [xh | xl] . [Wh | Wh]          activation exact (~22 bits), weight rounded once          2 x the MFMA work
    parts 3   [xh | xl | xh] . [Wh | Wh | Wl]  both exact (~22 bits)                                    3 x the MFMA work

over an fp32 activation stream (no rounding of layer outputs).  Errors of independent roundings add in quadrature while the
cost of a layer is its FLOP count, so the allocation is a knapsack: `tools/exp/precision_budget_gpu.py` measures, on the
MI355X with the product kernels, what each GEMM class costs in output error when it alone runs at 1 or 2 parts
(profiles/r03/precision_sensitivity.json), `tools/exp/precision_allocate.py` picks the cheapest part counts under an error
budget, and the table at the bottom of this file is the result (measured again as a whole: DESIGN.md §5).

A policy maps the emitter's launch-class name (the `name=` of Emitter.gemm / conv: "res.conv1", "attn1.qk", "vae.conv2", ...)
and the GEMM shape to a part count.  EDTR_AMD_POLICY (JSON: {"default": 2, "vae.conv1": 1, "vae.conv1@2097152": 1, ...};
"<name>@<M>" addresses one resolution level) overrides it for experiments.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional


class PrecisionPolicy:
    def __init__(self, default: int = 2, table: Optional[Dict[str, int]] = None, label: str = "custom", attn_split: Optional[int] = None):
        self.default = int(default)
        self.table = dict(table or {})
        self.label = label
        # attention operands: None = the mode's default (one fp16 part in the mixed mode); 1 = q / k as hi + lo pairs (three QK^T products),
        # 2 = q / k and p / v (Emitter.attn_split; EDTR_AMD_ATTN_SPLIT overrides)
        self.attn_split = attn_split
        for k, v in list(self.table.items()) + [("default", self.default)]:
            if v not in (1, 2, 3, 4):
                raise ValueError(f"precision policy: {k} -> {v!r} (parts must be 1, 2, 3 or 4 = the weights-exact two-part product)")

    def parts(self, name: str, M: int = 0, N: int = 0, K: int = 0) -> int:
        hit = self.table.get(f"{name}@{M}")
        if hit is None:
            hit = self.table.get(name)
        return self.default if hit is None else hit

    def key(self):
        return (self.default, tuple(sorted(self.table.items())), self.attn_split)

    def describe(self) -> dict:
        return {"label": self.label, "default": self.default, "table": dict(sorted(self.table.items())), "attn_split": self.attn_split}


class ConstPolicy(PrecisionPolicy):
    """Every GEMM the same part count (fast mode: 1; high mode: 3)."""

    def __init__(self, parts: int):
        super().__init__(parts, {}, f"const{parts}")


# The shipped allocation (see the module docstring; measured numbers in DESIGN.md §5 and profiles/r03/precision_*).
# What the sensitivities say: a GEMM that carries the WHOLE residual stream (1x1 skip / nin_shortcut convolutions, the input and
# output convolutions, the VAE's upsample convolutions, quant / post_quant) passes its operand roundings straight into the
# signal — each of these few, cheap classes costs 1e-4 .. 5e-4 of image error at one part — while the FLOP-heavy convolutions
# and linears sit inside residual branches whose output is a fraction of the stream (vae.conv1/2, res.conv1/2, ff.*: 1.4e-4 ..
# 2.2e-4 for ALL their launches together).  So: three parts on the stream carriers, one part everywhere else.
# Measured at full size (BASELINE configs[1], images 3 and 7 vs the reference): latent 5.2e-4, image 5.7e-4 at 81 images/s
# (all-1: 7.5e-4 / 1.14e-3 at 91; all-2: 6.6e-4 / 8.8e-4 at 58; all-3: 1.0e-4 / 1.0e-4 at 44; fast bf16: 6.2e-3 / 1.2e-2 at 107).
# Round 4: the UNet / ControlNet 1x1 skip convolutions run the WEIGHTS-EXACT two-part product (4 = ops.PARTS_2W: x16 . [Wh | Wl]
# with the activation's fp16 mirror read twice — no operand-formation launch, 2/3 of the MFMA work): their weight rounding carried
# 8.3e-8 of the 12.3e-8 variance they cost at one part (profiles/r03/precision_sensitivity.json), the activation rounding 2.9e-8.
# Measured as a whole (profiles/r04/ab_mixed_policies.log): image error 5.8e-4 -> 6.0e-4, +1.4 % throughput.
MIXED_TABLE: Dict[str, int] = {
    "conv_in": 3, "res.skip1x1": 4, "unet.out_conv": 3,
    "time_embed.0": 3, "time_embed.2": 3, "emb_layers(all)": 3, "ctx_k(all)": 3, "ctx_vT(all)": 3,
    "vae.conv_in": 3, "vae.conv_out": 3, "vae.nin_shortcut": 3, "vae.upsample.conv": 3, "vae.downsample": 3,
    "vae.quant_conv": 3, "vae.post_quant_conv": 3,
}
MIXED_DEFAULT = 1


def mixed_policy() -> PrecisionPolicy:
    env = os.environ.get("EDTR_AMD_POLICY")
    if env:
        spec = json.loads(env)
        table = dict(MIXED_TABLE) if spec.pop("base", None) == "shipped" else {}      # {"base": "shipped", ...}: overrides of the shipped table
        default = int(spec.pop("default", MIXED_DEFAULT))
        table.update({k: int(v) for k, v in spec.items()})
        return PrecisionPolicy(default, table, "env")
    return PrecisionPolicy(MIXED_DEFAULT, MIXED_TABLE, "mixed-r04")


# The ROBUST allocation (round 6, precision="robust"): what holds the north-star 1e-3 on OUTLIER-BEARING weights at the least cost.
# On the moderate-outlier weight set (edtr_amd.synth.synth_param_moderate; reference outputs in tests/golden/moderate.npz) the shipped
# table above leaves 1.6 - 1.9e-3 on the denoiser's output, and no re-allocation short of three parts on EVERY denoiser class closes it
# (profiles/r05/moderate_policies.log; profiles/r06/moderate_policies.log: the weights-exact two-part product on every linear still
# leaves eps at 1.2e-3 — with outlier rows in the weights the ACTIVATION rounding of the products they amplify matters too).  The VAE,
# on the other hand, is inside 1e-3 on the shipped allocation (encoder 6.4e-4, decoder 3.0e-4).  So: three parts on every denoiser
# class, the shipped table in the VAE, q / k of the attention as hi + lo pairs.  Measured (profiles/r06/moderate_policies2.log):
# moderate set z_pre 6.4e-4, latent 6.5e-4, image 6.3e-4, eps 7.8e-4, VAE 6.4e-4 / 3.0e-4 (without the q / k split: eps 9.6e-4) — every
# figure inside 1e-3, asserted against 1e-3 itself in tests/test_gpu_heavy.py; smooth set at full size 2.5e-4 / 3.8e-4.  Against
# precision="high" (bf16 split-3 everywhere, 40.9 images/s) it keeps the FLOP-heavy VAE convolutions at one product.
ROBUST_VAE_ONE_PART = ("vae.conv1", "vae.conv2", "vae.attn.qk", "vae.attn.vT", "vae.attn.proj_out", "vae.attn.flash")


def robust_policy() -> PrecisionPolicy:
    table = dict(MIXED_TABLE)
    table.update({k: 1 for k in ROBUST_VAE_ONE_PART})
    return PrecisionPolicy(3, table, "robust-r06", attn_split=1)
