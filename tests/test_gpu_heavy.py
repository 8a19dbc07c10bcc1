"""Range robustness on the MI355X (pytest -m gpu): the HEAVY-TAILED synthetic weight set (edtr_amd.synth.synth_param_heavy:
outlier channels, large norm gains, sharp attention; residual stream peaking at ~2e4, the edge of fp16) against the
reference's outputs on the same weights (tests/golden/heavy.npz, tools/make_goldens.py gen_heavy).  Every precision mode must
stay finite and inside its envelope; the attention kernel is also driven to logits of +-1e4 on its own.

The smooth weight set of the other fixtures has benign statistics (no outlier channels); released checkpoints do not exist
offline, so this is the closest stand-in for their operand range (VERDICT r02 missing 5 / next 9)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

USED = [50, 100, 150, 200]
# mode -> (compute dtype, precision, tolerances).  This network amplifies a perturbation ~10-30x more than the smooth set
# (sharper attention: the fp32 oracle itself is 6.6e-5 from the reference here, 2e-5 there).  Rounds 1-3: the attention operands
# were ONE fp16 part in every mode, and that bounded the parity modes (high: eps 2.6e-3).  Round 4:
# tests/heavy_attention_budget.py shows on the CPU oracle that the fp16 rounding of q and k ALONE costs 2.7e-3 here (p 4.8e-4,
# v 4.4e-4); the high mode now feeds the attention kernels hi + lo fp16 pairs (three MFMA products per product, edtr_hip.h q_lo /
# k_lo / vt_lo): eps 2.6e-3 -> 6.6e-5 — the north-star 1e-3 holds on this weight set too (sd21_eps), with the fp32 oracle's
# own distance as the floor.  The mixed mode keeps one-part attention operands (its projections write fp16 directly): it is the
# FAST parity mode, validated on the smooth set; EDTR_AMD_ATTN_SPLIT=1 buys the split there at the cost of throughput.
# Tolerances.  The VAE and single-evaluation figures are stable and held to <= 1.5 x measured; the 4-step PIPELINE on this
# weight set is not: two builds of the same arithmetic (separate q / k / v^T launches vs the fused projection, v_rsq vs
# 1 / sqrt in GroupNorm) moved its fp16 error from 6.9e-3 / 1.08e-2 to 1.37e-2 / 2.3e-2 (latent / image) — the amplification
# is that of the network, not of a kernel — so the pipeline latents / images get 2 - 2.5 x the measurement.
# Measured (round 4):
#   tiny pipeline (z_pre, z, img):  bf16 9.0e-3 6.2e-2 1.16e-1 | fp16 1.1e-3 1.34e-2 2.1e-2 | mixed 5.1e-4 3.4e-3 5.3e-3 | high 4.8e-5 7.0e-4 1.1e-3
#   SD-2.1 widths (eps, vae_z, vae_dec): bf16 6.6e-2 1.2e-2 1.1e-2 | fp16 5.4e-3 1.3e-3 1.4e-3 | mixed 4.3e-3 5.7e-4 4.6e-4 | high 6.6e-5 1.2e-4 1.9e-4
MODES = {"bf16": (torch.bfloat16, "fast", dict(z_pre=1.35e-2, z=1.5e-1, img=2.5e-1, eps=9.9e-2, vae=1.75e-2)),
         "fp16": (torch.float16, "fast", dict(z_pre=1.65e-3, z=3.4e-2, img=5.3e-2, eps=8.2e-3, vae=2.1e-3)),
         "mixed": (None, "mixed", dict(z_pre=7.6e-4, z=8.5e-3, img=1.35e-2, eps=6.4e-3, vae=8.6e-4)),
         "high": (None, "high", dict(z_pre=7.2e-5, z=1.5e-3, img=2.4e-3, eps=1.0e-4, vae=2.9e-4))}

def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def finite(*ts):
    return all(bool(torch.isfinite(t).all()) for t in ts)


@pytest.mark.parametrize("mode", list(MODES))
def test_tiny_pipeline_heavy_weights(golden_dir, mode):
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise
    d = dev()
    dtype, precision, tol = MODES[mode]
    g = np.load(os.path.join(golden_dir, "heavy.npz"))
    cldm = build_synthetic_cldm(synth.tiny_config(), d, dtype, precision=precision, weights="heavy")
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(d)
    sampler = SpacedSampler(diffusion.betas)
    B, H, W = 2, 128, 128
    pre_res = synth.synth_input("heavy:pre_res", (B, 3, H, W), 0.0, 1.0).to(d)
    c_txt = synth.synth_input("heavy:c_txt", (B, 77, 64), -1.0, 1.0).to(d)
    noises = [synth.synth_normal(f"heavy:noise{i}", (B, 4, H // 8, W // 8)).to(d) for i in range(5)]
    z_pre = cldm.vae_encode(pre_res * 2 - 1, sample=False)
    x_T = diffusion.q_sample(z_pre, torch.full((B,), 200, dtype=torch.int64, device=d), noises[0])
    with injected_noise(noises[1:]):
        z = sampler.manual_sample_with_timesteps(model=cldm, device=d, x_T=x_T, steps=4, used_timesteps=USED, batch_size=B,
                                                 cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False)
    img = cldm.vae_decode(z)
    torch.cuda.synchronize()
    assert finite(z_pre, z, img), "non-finite values with the heavy-tailed weights"
    errs = {"z_pre": rel(z_pre, g["z_pre"]), "z": rel(z, g["z"]), "img": rel(img, g["img"])}
    print(f"\n[heavy tiny, {mode}] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert all(v < tol[k] for k, v in errs.items()), errs


@pytest.mark.parametrize("mode", list(MODES))
def test_sd21_width_step_and_vae_heavy_weights(golden_dir, mode):
    """SD-2.1 widths: one ControlNet + UNet evaluation (residual stream up to ~2e4) and the VAE, heavy weights."""
    from edtr_amd import synth
    from edtr_amd.testing import build_synthetic_cldm
    d = dev()
    dtype, precision, tol = MODES[mode]
    g = np.load(os.path.join(golden_dir, "heavy.npz"))
    cldm = build_synthetic_cldm(synth.sd21_config(), d, dtype, precision=precision, weights="heavy")
    x = synth.synth_normal("heavy:x", (1, 4, 32, 32)).to(d)
    c_img = synth.synth_normal("heavy:c_img", (1, 4, 32, 32)).to(d)
    c_txt = synth.synth_input("heavy:c_txt", (1, 77, 1024), -1.0, 1.0).to(d)
    eps = cldm.forward(x, torch.tensor([200], device=d), {"c_txt": c_txt, "c_img": c_img})
    z = cldm.vae_encode(synth.synth_input("heavy:img", (1, 3, 128, 128), -1.0, 1.0).to(d), sample=False)
    dec = cldm.vae_decode(synth.synth_normal("heavy:zdec", (1, 4, 16, 16)).to(d))
    torch.cuda.synchronize()
    assert finite(eps, z, dec), "non-finite values with the heavy-tailed weights"
    errs = {"eps": rel(eps, g["sd21_eps"]), "vae_z": rel(z, g["sd21_vae_z"]), "vae_dec": rel(dec, g["sd21_vae_dec"])}
    print(f"\n[heavy sd21 widths, {mode}; reference stream peak {float(g['sd21_mid_absmax'][0]):.0f} at the middle block] "
          + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert errs["eps"] < tol["eps"] and errs["vae_z"] < tol["vae"] and errs["vae_dec"] < tol["vae"], errs


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,peak", [(256, 1.0e4), (4096, 1.0e4), (1024, 300.0)])
def test_flash_attention_with_extreme_logits(dtype, N, peak):
    """Logits up to +-peak (softmax essentially one-hot, exp of the raw logit overflows fp32 at 88): both attention kernels
    (the general one and, at N = 4096, the generated-asm large-N one incl. its re-maximise path) against torch softmax in
    fp64 on the same 16-bit inputs."""
    from edtr_amd import ops
    d = dev()
    B, H = 1, 2
    g = torch.Generator().manual_seed(7)
    s = (peak * 8.0 / 64.0) ** 0.5           # q.k / 8 over 64 dims of magnitude s*s: |logit| up to ~peak for aligned vectors
    q = (torch.randn((B, N, H * 64), generator=g) * s).to(dtype)
    k = (torch.randn((B, N, H * 64), generator=g) * s).to(dtype)
    k[:, ::7] = q[:, ::7]                    # aligned pairs: the largest logits sit on these keys
    v = torch.randn((B, N, H * 64), generator=g).to(dtype)
    vt = v.transpose(1, 2).contiguous()
    out = torch.empty((B * N, H * 64), dtype=dtype, device=d)
    qd, kd, vtd = q.to(d).reshape(B * N, -1), k.to(d).reshape(B * N, -1), vt.to(d).reshape(B * H * 64, N)
    ops.launch(ops.make_flash_attn(dtype=dtype, q=qd, k=kd, vt=vtd, out=out, B=B, H=H, Nq=N, Nk=N, q_bs=N * H * 64, q_ld=H * 64,
                                   k_bs=N * H * 64, k_ld=H * 64, vt_bs=H * 64 * N, vt_ld=N, o_bs=N * H * 64, o_ld=H * 64,
                                   scale=1.0 / 8.0))
    torch.cuda.synchronize()
    assert finite(out)
    qf, kf, vf = (t.double().reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in (q, k, v))
    logits = qf @ kf.transpose(-1, -2) / 8.0
    want = (torch.softmax(logits, dim=-1) @ vf).permute(0, 2, 1, 3).reshape(B * N, H * 64)
    e = rel(out.float(), want)
    print(f"\n[attention, {dtype}, N={N}] max |logit| {float(logits.abs().max()):.0f}; rel err {e:.2e}")
    assert e < (1.6e-3 if dtype == torch.bfloat16 else 2.2e-4)          # measured <= 1.03e-3 / 1.4e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_layernorm_fold_on_heavy_weights(golden_dir, dtype, monkeypatch):
    """the LayerNorm fold (on by default for batches of <= 4 images: det512s50, the tiled workload, SwinIR) takes
    its per-row mean / variance from single-pass sums written by the producing GEMM.  On the heavy-tailed weight set (outlier
    channels: rows whose mean dwarfs their spread) it must stay as close to the reference as the launched two-pass LayerNorm;
    since round 4 the sums are those of the STORED 16-bit values, i.e. of what the consuming GEMM multiplies."""
    from edtr_amd import synth
    from edtr_amd.testing import build_synthetic_cldm
    d = dev()
    g = np.load(os.path.join(golden_dir, "heavy.npz"))
    x = synth.synth_normal("heavy:x", (1, 4, 32, 32)).to(d)
    c_img = synth.synth_normal("heavy:c_img", (1, 4, 32, 32)).to(d)
    c_txt = synth.synth_input("heavy:c_txt", (1, 77, 1024), -1.0, 1.0).to(d)
    errs = {}
    for fold in ("0", "1"):
        monkeypatch.setenv("EDTR_LN_FOLD", fold)
        cldm = build_synthetic_cldm(synth.sd21_config(), d, dtype, weights="heavy")
        eps = cldm.forward(x, torch.tensor([200], device=d), {"c_txt": c_txt, "c_img": c_img})
        torch.cuda.synchronize()
        assert finite(eps)
        errs[fold] = rel(eps, g["sd21_eps"])
        cldm.release_engines()
    print(f"\n[heavy sd21 widths, {dtype}] eps error with the LayerNorm launched {errs['0']:.2e}, folded {errs['1']:.2e}")
    assert errs["1"] < 1.3 * errs["0"], errs


# ---------------------------------------------------------------------------------------------------------------------
# The MODERATE-outlier weight set (edtr_amd.synth.synth_param_moderate; tests/golden/moderate.npz from the reference, round 5):
# the parity modes' pinned claims OFF the smooth set on everything the reference produced: the tiny
# 4-step pipeline (encoded latent, final latent, decoded image), one ControlNet + UNet evaluation at the SD-2.1 widths, VAE encode and
# decode at the SD-2.1 widths.
# ---------------------------------------------------------------------------------------------------------------------
NORTH_STAR = 1e-3
# Measured (round 5; z_pre z img | eps vae_z vae_dec):
#   mixed 6.4e-4 1.88e-3 1.63e-3 | 1.69e-3 6.5e-4 2.9e-4      high 1.3e-4 8.3e-5 1.1e-4 | 2.8e-5 6.3e-5 1.0e-4
#   fp16  1.24e-3 3.4e-3 2.8e-3 | 2.4e-3 1.5e-3 1.3e-3         bf16 1.0e-2 2.8e-2 2.4e-2 | 2.0e-2 1.1e-2 9.9e-3
# What this pins: with outlier channels in the weights the FAST parity mode (one fp16 product on the FLOP-heavy classes) does NOT
# hold the north-star 1e-3 — 1.6 - 1.9e-3 on the denoiser's output — and no cheap re-allocation does (tools/exp/r05_moderate_policies*.sh,
# profiles/r05/moderate_policies.log: every UNet GEMM at three parts still leaves 1.2e-3, because the weight AND the activation
# rounding of every class contribute); three fp16 parts everywhere (5.4e-4 / 4.5e-4 / 6.4e-4) or the robust mode (`high`) do.  So the
# 1e-3 claim of `mixed` stays scoped to well-conditioned weights, and `high` is asserted here against the north star WITH a 4 x margin.
# Every figure is held to <= 1.5 x its measurement.
MODERATE = {"mixed": (None, "mixed", dict(z_pre=9.6e-4, z=2.8e-3, img=2.45e-3, eps=2.55e-3, vae=9.7e-4)),
            "high": (None, "high", dict(z_pre=1.9e-4, z=1.25e-4, img=1.6e-4, eps=4.3e-5, vae=1.5e-4)),
            # robust (round 6: three parts on every denoiser class, the shipped allocation in the VAE, q / k split): every figure is asserted
            # against the north-star 1e-3 ITSELF below (measured 6.4e-4 6.5e-4 6.3e-4 | 7.8e-4 6.4e-4 3.0e-4)
            "robust": (None, "robust", dict(z_pre=1e-3, z=1e-3, img=1e-3, eps=1e-3, vae=1e-3)),
            # hybrid (round 6: fp16 encoder + denoiser, mixed decoder) is fp16 on everything but the decoder: like `mixed`, scoped to well-conditioned weights
            "hybrid": (None, "hybrid", dict(z_pre=1.9e-3, z=5.1e-3, img=4.2e-3, eps=3.7e-3, vae=2.2e-3)),
            "fp16": (torch.float16, "fast", dict(z_pre=1.9e-3, z=5.1e-3, img=4.2e-3, eps=3.7e-3, vae=2.2e-3)),
            "bf16": (torch.bfloat16, "fast", dict(z_pre=1.55e-2, z=4.2e-2, img=3.6e-2, eps=2.95e-2, vae=1.7e-2))}


@pytest.mark.parametrize("mode", list(MODERATE))
def test_moderate_outlier_weights_pipeline_and_sd21_widths(golden_dir, mode):
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise
    d = dev()
    dtype, precision, tol = MODERATE[mode]
    g = np.load(os.path.join(golden_dir, "moderate.npz"))
    cldm = build_synthetic_cldm(synth.tiny_config(), d, dtype, precision=precision, weights="moderate")
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(d)
    sampler = SpacedSampler(diffusion.betas)
    B, H, W = 2, 128, 128
    pre_res = synth.synth_input("moderate:pre_res", (B, 3, H, W), 0.0, 1.0).to(d)
    c_txt = synth.synth_input("moderate:c_txt", (B, 77, 64), -1.0, 1.0).to(d)
    noises = [synth.synth_normal(f"moderate:noise{i}", (B, 4, H // 8, W // 8)).to(d) for i in range(5)]
    z_pre = cldm.vae_encode(pre_res * 2 - 1, sample=False)
    x_T = diffusion.q_sample(z_pre, torch.full((B,), 200, dtype=torch.int64, device=d), noises[0])
    with injected_noise(noises[1:]):
        z = sampler.manual_sample_with_timesteps(model=cldm, device=d, x_T=x_T, steps=4, used_timesteps=USED, batch_size=B,
                                                 cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False)
    img = cldm.vae_decode(z)
    torch.cuda.synchronize()
    assert finite(z_pre, z, img)
    errs = {"z_pre": rel(z_pre, g["z_pre"]), "z": rel(z, g["z"]), "img": rel(img, g["img"])}
    cldm.release_engines()
    del cldm
    big = build_synthetic_cldm(synth.sd21_config(), d, dtype, precision=precision, weights="moderate")
    x = synth.synth_normal("moderate:x", (1, 4, 32, 32)).to(d)
    c_img = synth.synth_normal("moderate:c_img", (1, 4, 32, 32)).to(d)
    c_txt = synth.synth_input("moderate:c_txt", (1, 77, 1024), -1.0, 1.0).to(d)
    eps = big.forward(x, torch.tensor([200], device=d), {"c_txt": c_txt, "c_img": c_img})
    vz = big.vae_encode(synth.synth_input("moderate:img", (1, 3, 128, 128), -1.0, 1.0).to(d), sample=False)
    dec = big.vae_decode(synth.synth_normal("moderate:zdec", (1, 4, 16, 16)).to(d))
    torch.cuda.synchronize()
    assert finite(eps, vz, dec)
    errs.update(eps=rel(eps, g["sd21_eps"]), vae_z=rel(vz, g["sd21_vae_z"]), vae_dec=rel(dec, g["sd21_vae_dec"]))
    print(f"\n[moderate, {mode}; reference stream peak {float(g['sd21_mid_absmax'][0]):.0f} at the SD-2.1 middle block] "
          + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    lim = dict(tol, vae_z=tol["vae"], vae_dec=tol["vae"])
    assert all(v < lim[k] for k, v in errs.items()), errs
    if mode == "high":          # the bf16 split-3 mode meets the north star on outlier-bearing weights with a 4 x margin
        assert max(errs.values()) < 0.25 * NORTH_STAR, errs
    if mode == "robust":        # ... and the cheapest allocation that holds it at all holds it on every figure
        assert max(errs.values()) < NORTH_STAR, errs
