"""The parity ("high") precision mode on the MI355X (pytest -m gpu): fp32 activation stream, bf16 split-3 GEMM operands
([hi | lo | hi] x [Wh | Wh | Wl], fp32 accumulation), fp16 attention operands — against the reference's fp32 outputs
(tests/golden/*.npz).  This is the mode that must meet the north-star tolerance: relative L2 error < 1e-3 on every stage
and on the decoded image.  (Why 16-bit operands cannot: tools/exp/precision_budget.py)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NORTH_STAR = 1e-3
# what the mode is held to: 1.5 x its largest measured error (round 4, attention operands split: per-step eps 3.0e-5; latents /
# images 1e-5 .. 2.9e-5; before the split: eps 2.1e-4, latents / images up to 7.5e-5)
HIGH_TOL = 4.5e-5
USED = [50, 100, 150, 200]


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def test_split3_operand_and_product():
    """edtr_split3 and the split product through edtr_igemm: x @ w^T to ~16 mantissa bits (vs 8 for one bf16 term)."""
    from edtr_amd import ops
    d = dev()
    M, N, K = 300, 72, 200
    x, w = rnd((M, K), 1), rnd((N, K), 2, 0.1)
    xd = x.to(d)
    x3 = torch.empty((M, 3 * K), dtype=torch.bfloat16, device=d)
    ops.launch(ops.make_split3(src=xd, rows=M, C=K, dst=x3))
    torch.cuda.synchronize()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert torch.equal(x3.cpu(), torch.cat([hi, lo, hi], dim=1))
    # pattern 1 and a 16-bit source
    x16 = x.to(torch.float16).to(d)
    y3 = torch.empty((M, 3 * K), dtype=torch.bfloat16, device=d)
    ops.launch(ops.make_split3(src=x16, rows=M, C=K, dst=y3, pattern=1))
    torch.cuda.synchronize()
    xf = x.to(torch.float16).float()
    h2 = xf.to(torch.bfloat16)
    assert torch.equal(y3.cpu(), torch.cat([h2, h2, (xf - h2.float()).to(torch.bfloat16)], dim=1))
    w3 = ops.pack_linear_weight(w, ops.F32S).to(d)
    assert tuple(w3.shape) == (N, 3 * K)
    out = torch.empty((M, N), dtype=torch.float32, device=d)
    ops.launch(ops.make_igemm(dtype=torch.bfloat16, a1=x3, w=w3, out=out, M=M, N=N, C1=3 * K, ld1=3 * K, ldw=3 * K, ldc=N,
                              out_f32=True))
    torch.cuda.synchronize()
    want = x.double() @ w.double().t()
    e3 = rel(out, want)
    e1 = rel(x.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t(), want)
    print(f"\n[split product] rel err {e3:.2e} (single bf16 term: {e1:.2e})")
    assert e3 < 3e-5 and e1 > 50 * e3


def test_norms_fp32_in_split_out():
    """GroupNorm(+SiLU) and LayerNorm in the high-precision mode: fp32 rows in, [hi | lo | hi] operand out."""
    import torch.nn.functional as F
    from edtr_amd import ops
    d = dev()
    B, C, H, W = 2, 64, 12, 10
    x = rnd((B, C, H, W), 3, 2.0) + 0.5
    gamma, beta = 1 + 0.1 * rnd((C,), 4), 0.1 * rnd((C,), 5)
    xn = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(d)
    sums = torch.zeros((B, 32, 2), dtype=torch.float64, device=d)
    y3 = torch.empty((B * H * W, 3 * C), dtype=torch.bfloat16, device=d)
    st, ap = ops.make_gn(dtype=ops.F32S, x=xn, ldx=C, B=B, HW=H * W, C=C, sums=sums, gamma=gamma.to(d), beta=beta.to(d),
                         eps=1e-6, silu=True, y=y3, ldy=3 * C)
    ops.launch(st)
    ops.launch(ap)
    torch.cuda.synchronize()
    want = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-6)).permute(0, 2, 3, 1).reshape(B * H * W, C)
    y = y3.float().cpu()
    assert torch.equal(y[:, :C], y[:, 2 * C:])
    e = rel(y[:, :C] + y[:, C:2 * C], want)
    print(f"\n[gn hp] rel err {e:.2e}")
    assert e < 2e-5
    rows, Cl = 70, 320
    t = rnd((rows, Cl), 6, 3.0)
    g2, b2 = 1 + 0.1 * rnd((Cl,), 7), 0.1 * rnd((Cl,), 8)
    o3 = torch.empty((rows, 3 * Cl), dtype=torch.bfloat16, device=d)
    ops.launch(ops.make_layernorm(dtype=ops.F32S, x=t.to(d), rows=rows, C=Cl, ldx=Cl, gamma=g2.to(d), beta=b2.to(d), eps=1e-5,
                                  y=o3, ldy=3 * Cl))
    torch.cuda.synchronize()
    o = o3.float().cpu()
    e = rel(o[:, :Cl] + o[:, Cl:2 * Cl], F.layer_norm(t.double(), (Cl,), g2.double(), b2.double(), 1e-5))
    print(f"[ln hp] rel err {e:.2e}")
    assert e < 2e-5


def _pipeline(golden_dir, name, tag, B, H, W):
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise
    d = dev()
    g = np.load(os.path.join(golden_dir, name))
    cldm = build_synthetic_cldm(synth.tiny_config(), d, precision="high")
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(d)
    sampler = SpacedSampler(diffusion.betas)
    pre_res = synth.synth_input(f"{tag}:pre_res", (B, 3, H, W), 0.0, 1.0).to(d)
    c_txt = synth.synth_input(f"{tag}:c_txt", (B, 77, 64), -1.0, 1.0).to(d)
    noises = [synth.synth_normal(f"{tag}:noise{i}", (B, 4, H // 8, W // 8)).to(d) for i in range(5)]
    z_pre = cldm.vae_encode(pre_res * 2 - 1, sample=False)
    x_T = diffusion.q_sample(z_pre, torch.full((B,), 200, dtype=torch.int64, device=d), noises[0])
    eps_log = []
    fwd = cldm.forward

    def logging_forward(x, t, cond, woSD=False):
        e = fwd(x, t, cond)
        eps_log.append(e.clone())
        return e

    cldm.forward = logging_forward
    with injected_noise(noises[1:]):
        z = sampler.manual_sample_with_timesteps(model=cldm, device=d, x_T=x_T, steps=4, used_timesteps=USED, batch_size=B,
                                                 cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False)
    img = cldm.vae_decode(z)
    torch.cuda.synchronize()
    errs = {"z_pre": rel(z_pre, g["z_pre"]), "z": rel(z, g["z"]), "img": rel(img, g["img"])}
    for i in range(4):
        errs[f"eps{i}"] = rel(eps_log[i], g[f"eps{i}"])
    print(f"\n[high precision {name}] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    return errs


@pytest.mark.parametrize("name,tag,B,H,W", [("tiny_pipeline.npz", "tiny", 2, 128, 128), ("tiny_pipeline_rect.npz", "tinyrect", 1, 192, 128)])
def test_tiny_pipeline_meets_the_north_star(golden_dir, name, tag, B, H, W):
    errs = _pipeline(golden_dir, name, tag, B, H, W)
    assert all(v < HIGH_TOL for v in errs.values()), errs


def test_sd21_width_networks_meet_the_north_star(golden_dir):
    """Full SD-2.1 widths, one denoise step at latent 64x64 + VAE encode / decode, high-precision mode."""
    from edtr_amd import synth
    from edtr_amd.testing import build_synthetic_cldm
    d = dev()
    g = np.load(os.path.join(golden_dir, "sd21_blocks.npz"))
    cldm = build_synthetic_cldm(synth.sd21_config(), d, precision="high")
    x = synth.synth_normal("sd21:x", (1, 4, 64, 64)).to(d)
    c_img = synth.synth_normal("sd21:c_img", (1, 4, 64, 64)).to(d)
    c_txt = synth.synth_input("sd21:c_txt", (1, 77, 1024), -1.0, 1.0).to(d)
    eps = cldm.forward(x, torch.tensor([200], device=d), {"c_txt": c_txt, "c_img": c_img})
    z = cldm.vae_encode(synth.synth_input("sd21:img", (1, 3, 256, 256), -1.0, 1.0).to(d), sample=False)
    dec = cldm.vae_decode(synth.synth_normal("sd21:zdec", (1, 4, 32, 32)).to(d))
    cldm.controlnet.compute_dtype = cldm.compute_dtype
    ctrl = cldm.controlnet(x=x, hint=c_img, timesteps=torch.tensor([200], device=d), context=c_txt)
    torch.cuda.synchronize()
    errs = {"eps": rel(eps, g["eps"]), "vae_z": rel(z, g["vae_z"]), "vae_dec": rel(dec, g["vae_dec"]), "ctrl12": rel(ctrl[12], g["ctrl12"])}
    print(f"\n[high precision sd21 widths] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert all(v < HIGH_TOL for v in errs.values()), errs


def test_det512_full_size_meets_the_north_star(golden_dir):
    """BASELINE configs[1] at full size in the parity mode: images 3 and 7 of the bench batch (run as a batch of 2) against the
    reference's outputs."""
    from edtr_amd import synth, workloads
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm
    d = dev()
    g = np.load(os.path.join(golden_dir, "full_det512.npz"))
    cldm = build_synthetic_cldm(synth.sd21_config(), d, precision="high")
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(d)
    sampler = SpacedSampler(diffusion.betas)
    full = workloads.make_inputs("det512", 1024, d, 8, 512)
    sel = [int(k) for k in g["images"]]
    inp = workloads.Inputs(full.pre_res[sel].contiguous(), full.c_txt[sel].contiguous(), [n[sel].contiguous() for n in full.noises], [],
                           full.t_start[:len(sel)])
    img, z, tr = workloads.restore_pass(cldm, diffusion, sampler, inp, "det512")
    torch.cuda.synchronize()
    errs = {"z_pre": rel(tr["z_pre"], g["z_pre"]), "z": rel(z, g["z"]),
            "img": rel(img[:, :, 1::4, 2::4], g["img_samples"].astype(np.float32))}
    print(f"\n[high precision det512 full size] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    # the image golden is stored as fp16 samples (rounding 2^-11 relative per sample -> 2.8e-4 rms): budget it
    assert errs["z_pre"] < 2.8e-5 and errs["z"] < 1.4e-5 and errs["img"] < 3.1e-5, errs      # measured 1.8e-5 / 9.2e-6 / 2.0e-5 (round 3: 7.2e-5 / 7.3e-5)


def test_tiled_paths_meet_the_north_star(golden_dir):
    """The tiled VAE (VAEHook: padded tiles, GroupNorm statistics pooled across tiles) and the latent-tiled sampler in the
    high-precision mode against the reference's tiled outputs."""
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise
    d = dev()
    cldm = build_synthetic_cldm(synth.tiny_config(), d, precision="high")
    g = np.load(os.path.join(golden_dir, "tiled_vae.npz"))
    img = synth.synth_input("tvae:img", (1, 3, 192, 256), -1.0, 1.0).to(d)
    zin = synth.synth_normal("tvae:z", (1, 4, 32, 40)).to(d)
    z_t = cldm.vae_encode(img, sample=False, tiled=True, tile_size=64)
    d_t = cldm.vae_decode(zin, tiled=True, tile_size=8)
    torch.cuda.synchronize()
    e_enc, e_dec = rel(z_t, g["z_tiled"]), rel(d_t, g["dec_tiled"])
    g2 = np.load(os.path.join(golden_dir, "tiled.npz"))
    sampler = SpacedSampler(Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).betas)
    B, h, w = 1, 16, 24
    x_T = synth.synth_normal("tiled:x_T", (B, 4, h, w)).to(d)
    c_img = synth.synth_normal("tiled:c_img", (B, 4, h, w)).to(d)
    c_txt = synth.synth_input("tiled:c_txt", (B, 77, 64), -1.0, 1.0).to(d)
    noises = [synth.synth_normal(f"tiled:noise{i}", (B, 4, h, w)).to(d) for i in range(4)]
    with injected_noise(noises):
        z = sampler.manual_sample_with_timesteps(model=cldm, device=d, x_T=x_T, steps=4, used_timesteps=USED, batch_size=B,
                                                 cond={"c_txt": c_txt, "c_img": c_img}, uncond=None, cfg_scale=1.0, tiled=True,
                                                 tile_size=8, tile_stride=4, progress=False)
    torch.cuda.synchronize()
    e_z = rel(z, g2["z_tiled"])
    print(f"\n[high precision tiled] tiled vae enc {e_enc:.2e} dec {e_dec:.2e}; latent-tiled sampler {e_z:.2e}")
    assert e_enc < 5e-5 and e_dec < 5e-5 and e_z < 5.5e-5      # measured 3.0e-5 / 1.7e-5 / 3.4e-5


@pytest.mark.parametrize("B,H,Nq,Nk,causal,sharp", [(2, 3, 300, 300, False, 1.0), (1, 2, 1024, 77, False, 4.0), (1, 5, 256, 256, False, 8.0),
                                                    (2, 1, 130, 130, True, 2.0), (1, 2, 64, 64, False, 1.0)])
def test_flash_attention_split_operands(B, H, Nq, Nk, causal, sharp):
    """ABI 7, the robust parity mode's attention: q / k (and p / v) as fp16 hi + lo pairs, three MFMA products per product
    (include/edtr_hip.h: q_lo / k_lo / vt_lo), against fp64 attention of the UNROUNDED fp32 operands.  ``sharp`` scales the logits
    (sharp softmax is where the fp16 rounding of q and k hurts: tests/heavy_attention_budget.py).  Errors must order as
    one part > q, k split > everything split, and the fully split form must reach fp32-arithmetic accuracy."""
    from edtr_amd import ops
    d = dev()
    Cc = H * 64
    g = torch.Generator().manual_seed(11)
    q32 = torch.randn((B * Nq, Cc), generator=g) * sharp
    k32 = torch.randn((B * Nk, Cc), generator=g)
    v32 = torch.randn((B * Nk, Cc), generator=g)
    ldv = ops.round_up(Nk, 8)
    vt32 = torch.zeros((B * Cc, ldv))
    vt32.view(B, Cc, ldv)[:, :, :Nk] = v32.view(B, Nk, Cc).transpose(1, 2)
    dt = torch.float16

    def pair(x):            # [rows, C] fp32 -> [rows, 2C] fp16 = [hi | lo] on the device through the product's own kernel
        y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=dt, device=d)
        ops.launch(ops.make_split_operand(src=x.to(d), rows=x.shape[0], C=x.shape[1], dst=y, fmt=ops.F32H[2]))
        return y

    q2, k2, v2 = pair(q32), pair(k32), pair(vt32)
    qf, kf, vf = (t.double().reshape(B, -1, H, 64).transpose(1, 2) for t in (q32, k32, v32))
    logits = qf @ kf.transpose(-1, -2) / 8.0
    if causal:
        logits = logits + torch.triu(torch.full((Nq, Nk), float("-inf"), dtype=torch.float64), 1)
    want = (torch.softmax(logits, dim=-1) @ vf).transpose(1, 2).reshape(B * Nq, Cc)
    errs = {}
    for mode in ("one", "qk", "qkpv"):
        out = torch.full((B * Nq, Cc), float("nan"), dtype=torch.float32 if mode == "qkpv" else dt, device=d)
        kw = {}
        if mode != "one":
            kw.update(q_lo=q2[:, Cc:], k_lo=k2[:, Cc:])
        if mode == "qkpv":
            kw.update(vt_lo=v2[:, ldv:], out_f32=True)
        ops.launch(ops.make_flash_attn(dtype=dt, q=q2[:, :Cc], k=k2[:, :Cc], vt=v2[:, :ldv], out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                       q_bs=Nq * 2 * Cc, q_ld=2 * Cc, k_bs=Nk * 2 * Cc, k_ld=2 * Cc, vt_bs=Cc * 2 * ldv, vt_ld=2 * ldv,
                                       o_bs=Nq * Cc, o_ld=Cc, scale=0.125, causal=causal, **kw))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all(), mode
        errs[mode] = rel(out.float(), want)
    print(f"\n[split attention B{B} H{H} {Nq}x{Nk} sharp {sharp}] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert errs["qkpv"] < 3e-6, errs                       # fp32 accumulation + fp32 softmax only
    assert errs["qk"] < 4.5e-4, errs                       # p and v carry one fp16 rounding each
    assert errs["one"] > errs["qk"] > errs["qkpv"], errs
