#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by running the *reference*
(/root/reference, imported on CPU via tools/ref_import.py) on synthetic weights and inputs.

Run in the build container only (the reference does not exist on the GPU box):

    python tools/make_goldens.py [--only schedule,tiny,sd21,tiled,tiledvae,vaesample,heavy,wavelet,clip,psnr,swinir,full,tokens]

Fixtures are data (inputs are regenerated from edtr_amd.synth formulas, expected outputs
are stored); nothing from the reference's source travels.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from edtr_amd import synth  # noqa: E402
import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
USED_TIMESTEPS = [50, 100, 150, 200]


def build_reference_cldm(cfg_name: str, weights: str = "smooth"):
    ControlLDM, _, _, _ = ref_import.import_reference()
    cfg = synth.CONFIGS[cfg_name]()
    with contextlib.redirect_stdout(io.StringIO()):
        cldm = ControlLDM(**cfg)
    cldm.eval()
    gen = synth.WEIGHT_SETS[weights]
    with torch.no_grad():
        for key, val in cldm.state_dict().items():
            if key.startswith("clip."):
                continue
            val.copy_(gen(key, tuple(val.shape)))
    return cldm, cfg


def manifest(cldm) -> dict:
    out = {}
    for part in ("unet", "controlnet", "vae"):
        sd = getattr(cldm, part).state_dict()
        out[part] = [[k, list(v.shape)] for k, v in sd.items()]
    return out


@contextlib.contextmanager
def injected_noise(noises):
    """The reference draws torch.randn_like once per p_sample (utils/sampler.py:199); feed it a
    fixed list instead so CPU/GPU runs can replay the same stream."""
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: next(it).to(x)
    try:
        yield
    finally:
        torch.randn_like = orig


def gen_schedule():
    _, Diffusion, SpacedSampler, _ = ref_import.import_reference()
    from model.util import timestep_embedding
    from utils.sampler import space_timesteps
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    sampler = SpacedSampler(diffusion.betas)
    out = {"betas": diffusion.betas,
           "q_sqrt_ac": diffusion.sqrt_alphas_cumprod.numpy(),
           "q_sqrt_1mac": diffusion.sqrt_one_minus_alphas_cumprod.numpy()}
    names = ["sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    sampler.make_schedule(4, USED_TIMESTEPS)
    for n in names:
        out["s4_" + n] = getattr(sampler, n).numpy().copy()
    out["s4_timesteps"] = sampler.timesteps.copy()
    sampler.make_schedule(50)
    for n in names:
        out["s50_" + n] = getattr(sampler, n).numpy().copy()
    out["s50_timesteps"] = sampler.timesteps.copy()
    out["space_1000_50"] = np.array(sorted(space_timesteps(1000, "50")), dtype=np.int32)
    out["space_1000_10_15_20"] = np.array(sorted(space_timesteps(300, [10, 15, 20])), dtype=np.int32)
    out["space_ddim25"] = np.array(sorted(space_timesteps(1000, "ddim25")), dtype=np.int32)
    t = torch.tensor([50, 100, 150, 200, 999], dtype=torch.int64)
    out["temb_320"] = timestep_embedding(t, 320).numpy()
    out["temb_64"] = timestep_embedding(t, 64).numpy()
    # elementwise sampler algebra on a small tensor, all 4 indices
    x = synth.synth_normal("sched_x", (4, 4, 8, 8))
    eps = synth.synth_normal("sched_eps", (4, 4, 8, 8))
    noise = synth.synth_normal("sched_noise", (4, 4, 8, 8))
    sampler.make_schedule(4, USED_TIMESTEPS)
    index = torch.tensor([0, 1, 2, 3], dtype=torch.int64)

    class _Fixed(torch.nn.Module):
        def forward(self, x_, t_, cond_):
            return eps

    with injected_noise([noise]):
        x_prev, pred_x0 = sampler.p_sample(_Fixed(), x, torch.tensor([50, 100, 150, 200]), index, None, None, 1.0)
    out["p_sample_x_prev"] = x_prev.numpy()
    out["p_sample_pred_x0"] = pred_x0.numpy()
    out["q_sample_200"] = diffusion.q_sample(x, torch.full((4,), 200, dtype=torch.int64), noise).numpy()
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)
    print("schedule.npz written")


def run_pipeline(cldm, cfg, B, H, W, tag, Diffusion, SpacedSampler, ref_common, store_controls=True):
    ctx_dim = cfg["unet_cfg"]["context_dim"]
    pre_res = synth.synth_input(f"{tag}:pre_res", (B, 3, H, W), 0.0, 1.0)
    c_txt = synth.synth_input(f"{tag}:c_txt", (B, 77, ctx_dim), -1.0, 1.0)
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    sampler = SpacedSampler(diffusion.betas)
    out = {}
    eps_list, ctrl_list = [], []
    h1 = cldm.register_forward_hook(lambda m, i, o: eps_list.append(o.detach().clone()))
    h2 = cldm.controlnet.register_forward_hook(lambda m, i, o: ctrl_list.append([c.detach().clone() for c in o]))
    with torch.no_grad():
        t0 = time.time()
        z_pre = cldm.vae_encode(pre_res * 2 - 1, sample=False)
        t_enc = time.time() - t0
        noises = [synth.synth_normal(f"{tag}:noise{i}", tuple(z_pre.shape)) for i in range(5)]
        x_T = diffusion.q_sample(z_pre, torch.full((B,), 200, dtype=torch.int64), noises[0])
        cond = {"c_txt": c_txt, "c_img": z_pre}
        t0 = time.time()
        with injected_noise(noises[1:]):
            z, inter = sampler.manual_sample_with_timesteps(
                model=cldm, device="cpu", x_T=x_T, steps=4, used_timesteps=USED_TIMESTEPS, batch_size=B,
                cond=cond, uncond=None, cfg_scale=1.0, progress=False, return_intermediates=True)
        t_smp = time.time() - t0
        t0 = time.time()
        img = cldm.vae_decode(z)
        t_dec = time.time() - t0
        res = ref_common.wavelet_reconstruction((img + 1) / 2, pre_res)
    h1.remove()
    h2.remove()
    out["z_pre"] = z_pre.numpy()
    out["x_T"] = x_T.numpy()
    for i, e in enumerate(eps_list):
        out[f"eps{i}"] = e.numpy()
    for i, p in enumerate(inter):
        out[f"pred_x0_{i}"] = p.numpy()
    out["z"] = z.numpy()
    out["img"] = img.numpy()
    out["res_wavelet"] = res.numpy()
    ctrl0 = ctrl_list[0]
    out["ctrl_stats"] = np.array([[float(c.mean()), float(c.abs().mean()), float(c.abs().max())] for c in ctrl0],
                                 dtype=np.float64)
    if store_controls:
        for i in (0, 3, 6, 12):
            out[f"ctrl{i}"] = ctrl0[i].numpy().astype(np.float16)
    out["timing_s"] = np.array([t_enc, t_smp, t_dec])
    return out


def gen_tiny():
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    cldm, cfg = build_reference_cldm("tiny")
    with open(os.path.join(GOLD, "manifest_tiny.json"), "w") as f:
        json.dump(manifest(cldm), f)
    out = run_pipeline(cldm, cfg, 2, 128, 128, "tiny", Diffusion, SpacedSampler, ref_common)
    np.savez_compressed(os.path.join(GOLD, "tiny_pipeline.npz"), **out)
    print("tiny_pipeline.npz written; timings", out["timing_s"])
    # non-square, B=1 (seg-style 72x96-like latent in miniature: 192x128 image -> 24x16 latent)
    out = run_pipeline(cldm, cfg, 1, 192, 128, "tinyrect", Diffusion, SpacedSampler, ref_common, store_controls=False)
    np.savez_compressed(os.path.join(GOLD, "tiny_pipeline_rect.npz"), **out)
    print("tiny_pipeline_rect.npz written")


def gen_sd21():
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    t0 = time.time()
    cldm, cfg = build_reference_cldm("sd21")
    print(f"sd21 reference built + synthetic weights in {time.time() - t0:.1f}s")
    with open(os.path.join(GOLD, "manifest_sd21.json"), "w") as f:
        json.dump(manifest(cldm), f)
    out = {}
    with torch.no_grad():
        # one ControlLDM.forward at the true hot shape (latent 64x64), B=1, t=200
        x = synth.synth_normal("sd21:x", (1, 4, 64, 64))
        c_img = synth.synth_normal("sd21:c_img", (1, 4, 64, 64))
        c_txt = synth.synth_input("sd21:c_txt", (1, 77, 1024), -1.0, 1.0)
        t = torch.tensor([200], dtype=torch.int64)
        ctrl = cldm.controlnet(x=x, hint=c_img, timesteps=t, context=c_txt)
        out["ctrl_stats"] = np.array([[float(c.mean()), float(c.abs().mean()), float(c.abs().max())] for c in ctrl],
                                     dtype=np.float64)
        out["ctrl12"] = ctrl[12].numpy()
        out["ctrl0_f16"] = ctrl[0].numpy().astype(np.float16)
        t0 = time.time()
        eps = cldm(x, t, {"c_txt": c_txt, "c_img": c_img})
        print(f"sd21 cldm forward {time.time() - t0:.1f}s")
        out["eps"] = eps.numpy()
        # VAE at full width on a 256x256 image (latent 32x32)
        img = synth.synth_input("sd21:img", (1, 3, 256, 256), -1.0, 1.0)
        z = cldm.vae_encode(img, sample=False)
        out["vae_z"] = z.numpy()
        zin = synth.synth_normal("sd21:zdec", (1, 4, 32, 32))
        dec = cldm.vae_decode(zin)
        out["vae_dec"] = dec.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "sd21_blocks.npz"), **out)
    print("sd21_blocks.npz written")


def gen_moderate():
    """The same fixtures on the MODERATE-outlier weight set (edtr_amd.synth.synth_param_moderate: x8 rows, +-3 gains, x1.3 q / k,
    x3 biases — the constants MODERATE_ROW / MODERATE_GAIN / MODERATE_QK there are authoritative): the robustness set off the smooth
    one; which precision modes hold the north-star 1e-3 on it is pinned by tests/test_gpu_heavy.py (tests/golden/moderate.npz)."""
    gen_heavy(weights="moderate", tag="moderate", fname="moderate.npz")


def gen_heavy(weights="heavy", tag="heavy", fname="heavy.npz"):
    """Range-robustness fixtures: the reference on the HEAVY-TAILED weight set (edtr_amd.synth.synth_param_heavy: outlier
    channels x50, norm gains +-10, sharp attention) — the tiny end-to-end pipeline, and one denoise step + VAE at SD-2.1
    widths on small grids (latent 32x32; 128x128 image; 16x16 latent)."""
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    cldm, cfg = build_reference_cldm("tiny", weights)
    out = run_pipeline(cldm, cfg, 2, 128, 128, tag, Diffusion, SpacedSampler, ref_common, store_controls=False)
    out = {k: v for k, v in out.items() if k in ("z_pre", "eps0", "eps3", "z", "img", "ctrl_stats")}
    del cldm
    cldm, cfg = build_reference_cldm("sd21", weights)
    with torch.no_grad():
        x = synth.synth_normal(f"{tag}:x", (1, 4, 32, 32))
        c_img = synth.synth_normal(f"{tag}:c_img", (1, 4, 32, 32))
        c_txt = synth.synth_input(f"{tag}:c_txt", (1, 77, 1024), -1.0, 1.0)
        t = torch.tensor([200], dtype=torch.int64)
        acts = {}
        # activation magnitudes the fixture exercises (max |x| of the residual stream at the ends of the UNet encoder)
        h = cldm.unet.middle_block.register_forward_hook(lambda m, i, o: acts.__setitem__("mid_absmax", float(o.abs().max())))
        out["sd21_eps"] = cldm(x, t, {"c_txt": c_txt, "c_img": c_img}).numpy()
        h.remove()
        out["sd21_mid_absmax"] = np.array([acts["mid_absmax"]])
        out["sd21_vae_z"] = cldm.vae_encode(synth.synth_input(f"{tag}:img", (1, 3, 128, 128), -1.0, 1.0), sample=False).numpy()
        out["sd21_vae_dec"] = cldm.vae_decode(synth.synth_normal(f"{tag}:zdec", (1, 4, 16, 16))).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, "written; |eps| max", float(np.abs(out["sd21_eps"]).max()), "mid absmax", acts["mid_absmax"],
          "tiny img absmax", float(np.abs(out["img"]).max()))


def gen_tiled():
    """cfg-4 style paths on the tiny config: latent-tiled ControlLDM step (tile 8 / stride 4 latent px
    on a 16x24 latent) and the tile geometry helpers."""
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    out = {}
    out["gauss_64"] = ref_common.gaussian_weights(64, 64)
    out["gauss_8x8"] = ref_common.gaussian_weights(8, 8)
    out["win_128_128_64_32"] = np.array(ref_common.sliding_windows(128, 128, 64, 32), dtype=np.int32)
    out["win_72_96_64_32"] = np.array(ref_common.sliding_windows(72, 96, 64, 32), dtype=np.int32)
    out["win_16_24_8_4"] = np.array(ref_common.sliding_windows(16, 24, 8, 4), dtype=np.int32)
    cldm, cfg = build_reference_cldm("tiny")
    B, h, w = 1, 16, 24
    ctx_dim = cfg["unet_cfg"]["context_dim"]
    x_T = synth.synth_normal("tiled:x_T", (B, 4, h, w))
    c_img = synth.synth_normal("tiled:c_img", (B, 4, h, w))
    c_txt = synth.synth_input("tiled:c_txt", (B, 77, ctx_dim), -1.0, 1.0)
    noises = [synth.synth_normal(f"tiled:noise{i}", (B, 4, h, w)) for i in range(4)]
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    sampler = SpacedSampler(diffusion.betas)
    with torch.no_grad(), injected_noise(noises):
        z = sampler.manual_sample_with_timesteps(
            model=cldm, device="cpu", x_T=x_T, steps=4, used_timesteps=USED_TIMESTEPS, batch_size=B,
            cond={"c_txt": c_txt, "c_img": c_img}, uncond=None, cfg_scale=1.0,
            tiled=True, tile_size=8, tile_stride=4, progress=False)
    out["z_tiled"] = z.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiled.npz"), **out)
    print("tiled.npz written")


def gen_tiledvae():
    """VAEHook paths (reference utils/tilevae/tilevae.py) on the tiny config: tiled encode (tile 64 px, pad 32) of a
    192x256 image and tiled decode (tile 8 latent px, pad 11) of a 32x40 latent, plus tile geometry tables."""
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    from utils.tilevae import VAEHook
    cldm, cfg = build_reference_cldm("tiny")
    out = {}
    for name, (h, w, ts, dec) in {"enc_1024_256": (1024, 1024, 256, False), "enc_192x256_64": (192, 256, 64, False),
                                  "dec_128_64": (128, 128, 64, True), "dec_32x40_8": (32, 40, 8, True),
                                  "enc_560x760_256": (560, 760, 256, False)}.items():
        hook = VAEHook(None, ts, dec, False, False, True)
        with contextlib.redirect_stdout(io.StringIO()):
            ib, ob = hook.split_tiles(h, w)
        out[f"bbox_in_{name}"] = np.array(ib, dtype=np.int32)
        out[f"bbox_out_{name}"] = np.array(ob, dtype=np.int32)
    img = synth.synth_input("tvae:img", (1, 3, 192, 256), -1.0, 1.0)
    zin = synth.synth_normal("tvae:z", (1, 4, 32, 40))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        out["z_tiled"] = cldm.vae_encode(img, sample=False, tiled=True, tile_size=64).numpy()
        out["z_plain"] = cldm.vae_encode(img, sample=False).numpy()
        out["dec_tiled"] = cldm.vae_decode(zin, tiled=True, tile_size=8).numpy()
        out["dec_plain"] = cldm.vae_decode(zin).numpy()
    np.savez_compressed(os.path.join(GOLD, "tiled_vae.npz"), **out)
    print("tiled_vae.npz written; tiled vs plain rel diff: enc",
          float(np.linalg.norm(out["z_tiled"] - out["z_plain"]) / np.linalg.norm(out["z_plain"])), "dec",
          float(np.linalg.norm(out["dec_tiled"] - out["dec_plain"]) / np.linalg.norm(out["dec_plain"])))


def gen_vaesample():
    """ControlLDM.vae_encode with its DEFAULT sample=True (model/cldm.py:107-134, model/distributions.py:38-41) on the tiny
    config: the reference draws torch.randn(mean.shape) on the host generator, so a seeded call is reproducible anywhere."""
    cldm, cfg = build_reference_cldm("tiny")
    img = synth.synth_input("vsample:img", (2, 3, 64, 96), -1.0, 1.0)
    out = {"seed": np.array([1234])}
    with torch.no_grad():
        torch.manual_seed(1234)
        out["z_sample"] = cldm.vae_encode(img).numpy()                       # sample=True is the signature's default
        out["z_mode"] = cldm.vae_encode(img, sample=False).numpy()
        moments = cldm.vae.quant_conv(cldm.vae.encoder(img))
        out["logvar_minmax"] = np.array([float(moments[:, 4:].min()), float(moments[:, 4:].max())])
    np.savez_compressed(os.path.join(GOLD, "vae_sample.npz"), **out)
    print("vae_sample.npz written; |sample - mode| / |mode| =",
          float(np.linalg.norm(out["z_sample"] - out["z_mode"]) / np.linalg.norm(out["z_mode"])), "logvar range", out["logvar_minmax"])


def gen_wavelet():
    _, _, _, ref_common = ref_import.import_reference()
    a = synth.synth_input("wav:content", (2, 3, 96, 80), 0.0, 1.0)
    b = synth.synth_input("wav:style", (2, 3, 96, 80), 0.0, 1.0)
    out = {"recon": ref_common.wavelet_reconstruction(a, b).numpy()}
    hf, lf = ref_common.wavelet_decomposition(a)
    out["high"] = hf.numpy()
    out["low"] = lf.numpy()
    np.savez_compressed(os.path.join(GOLD, "wavelet.npz"), **out)
    print("wavelet.npz written")


def gen_clip():
    """The reference FrozenOpenCLIPEmbedder (model/clip.py) on synthetic weights: a small tower (head width 64) and the
    full ViT-H text tower of configs/det/demo.yaml (24 layers, width 1024, penultimate layer), plus its key manifest."""
    ref_import.install_stubs()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    from model.clip import FrozenOpenCLIPEmbedder
    out = {}
    tokens = synth.clip_test_tokens()
    out["tokens"] = tokens.numpy()
    man = {}
    for tag, cfg in (("small", synth.clip_small_config()), ("vith", synth.sd21_config()["clip_cfg"])):
        with contextlib.redirect_stdout(io.StringIO()):
            m = FrozenOpenCLIPEmbedder(**cfg).eval()
        with torch.no_grad():
            for key, val in m.state_dict().items():
                val.copy_(synth.synth_param(f"clip{tag}." + key, tuple(val.shape)))
            z = m(tokens)
        man[tag] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        out[f"z_{tag}"] = z.numpy().astype(np.float32) if tag == "small" else z[:2].numpy().astype(np.float16)
        out[f"stats_{tag}"] = np.array([float(z.mean()), float(z.abs().mean()), float(z.abs().max())])
        print(tag, out[f"stats_{tag}"])
    np.savez_compressed(os.path.join(GOLD, "clip_text.npz"), **out)
    with open(os.path.join(GOLD, "manifest_clip.json"), "w") as f:
        json.dump(man, f)
    print("clip_text.npz written")


def gen_psnr():
    """PSNR / YCbCr of the reference's utils/common.py on a synthetic image pair (tests/golden/psnr.npz)."""
    _, _, _, ref_common = ref_import.import_reference()
    a = synth.synth_input("psnr:a", (3, 3, 40, 56), 0.0, 1.0)
    b = (a + 0.05 * synth.synth_normal("psnr:n", (3, 3, 40, 56))).clamp(0, 1)
    np.savez_compressed(os.path.join(GOLD, "psnr.npz"), psnr_0=ref_common.calculate_psnr_pt(a, b, 0, False).numpy(),
                        psnr_4y=ref_common.calculate_psnr_pt(a, b, 4, True).numpy(), ycbcr=ref_common.rgb2ycbcr_pt(a).numpy())
    print("psnr.npz written")


def build_reference_swinir(tag: str, cfg: dict):
    ref_import.install_stubs()
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    from model.swinir import SwinIR
    with contextlib.redirect_stdout(io.StringIO()):
        m = SwinIR(**cfg).eval()
    with torch.no_grad():
        for key, val in m.state_dict().items():
            if val.dtype.is_floating_point and not key.endswith("attn_mask"):     # buffers keep the reference's own values
                val.copy_(synth.synth_param(f"swinir{tag}." + key, tuple(val.shape)))
    return m


def gen_swinir():
    """The reference SwinIR (model/swinir.py) on synthetic weights: a 2x2-layer network on a non-square input (every
    window / shift-mask case), the shipped 8x6-layer network at 256^2 (full output) and 512^2 (statistics + samples),
    its relative-position index and shift masks, and the state-dict key manifest."""
    out, man = {}, {}
    small = build_reference_swinir("small", synth.swinir_small_config())
    man["small"] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in small.state_dict().items()]
    x = synth.synth_input("swinir:small", (2, 3, 128, 192), 0.0, 1.0)
    with torch.no_grad():
        out["y_small"] = small(x).numpy()
    # an input that is not a multiple of the window: check_image_size reflect-pads it in image space and the final crop
    # `[:H*sf, :W*sf]` is a no-op for the pixel-unshuffle configuration, i.e. the PADDED size comes back (model/swinir.py:834-839,894)
    with torch.no_grad():
        out["y_small_padded"] = small(synth.synth_input("swinir:odd", (1, 3, 60, 124), 0.0, 1.0)).numpy()
    blk = small.layers[0].residual_group.blocks[1]
    out["rel_index"] = blk.attn.relative_position_index.numpy().astype(np.int16)
    out["mask_64x64"] = np.packbits(blk.attn_mask.numpy() != 0)
    out["mask_16x24"] = np.packbits(blk.calculate_mask((16, 24)).numpy() != 0)
    out["mask_value"] = np.array([float(blk.attn_mask.min())])
    full = build_reference_swinir("full", synth.swinir_config())
    man["full"] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in full.state_dict().items()]
    with torch.no_grad():
        t0 = time.time()
        y256 = full(synth.synth_input("swinir:256", (1, 3, 256, 256), 0.0, 1.0))
        y512 = full(synth.synth_input("swinir:512", (1, 3, 512, 512), 0.0, 1.0))
        print(f"full SwinIR 256^2 + 512^2 on CPU: {time.time() - t0:.1f} s")
    out["y_256"] = y256.numpy().astype(np.float16)
    out["y_512_stride8"] = y512[:, :, 3::8, 5::8].numpy()
    out["y_512_stats"] = np.array([float(y512.mean()), float(y512.abs().mean()), float(y512.abs().max()), float(y512.std())])
    print("small", float(out["y_small"].mean()), float(np.abs(out["y_small"]).max()), "full stats", out["y_512_stats"])
    np.savez_compressed(os.path.join(GOLD, "swinir.npz"), **out)
    with open(os.path.join(GOLD, "manifest_swinir.json"), "w") as f:
        json.dump(man, f)
    print("swinir.npz written")


def gen_demo():
    """The demo flow, demo.py:84-131 + the crop of :165, in miniature through the REFERENCE's own functions: a 150 x 100 low-quality
    image -> pad_if_smaller(128) (512 in the demo) -> pad_to_multiples_of(64) -> SwinIR (2 x 2-layer) -> vae_encode(pre * 2 - 1) ->
    q_sample(t = 200) -> 4 spaced steps -> vae_decode -> (x + 1) / 2 -> wavelet_reconstruction(res, pre) -> [:, :h0, :w0].
    The golden of edtr_amd.evalutil.restore_dataset(pad_mode="demo") (VERDICT r04 item 9 / SURVEY §8 row f4)."""
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    cldm, cfg = build_reference_cldm("tiny")
    swinir = build_reference_swinir("small", synth.swinir_small_config())
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    sampler = SpacedSampler(diffusion.betas)
    img = synth.synth_input("demo:lq", (1, 3, 150, 100), 0.0, 1.0)
    h0, w0 = img.shape[2:]
    x = ref_common.pad_to_multiples_of(ref_common.pad_if_smaller(img, size=128), multiple=64)
    c_txt = synth.synth_input("demo:c_txt", (1, 77, cfg["unet_cfg"]["context_dim"]), -1.0, 1.0)
    out = {"padded_shape": np.array(x.shape)}
    with torch.no_grad():
        pre = swinir(x)
        z_pre = cldm.vae_encode(pre * 2 - 1, sample=False)
        noises = [synth.synth_normal(f"demo:noise{i}", tuple(z_pre.shape)) for i in range(5)]
        with injected_noise(noises):
            noise = torch.randn_like(z_pre)                      # demo.py:109
            z_partial = diffusion.q_sample(x_start=z_pre, t=torch.tensor([200], dtype=torch.int64), noise=noise)
            z = sampler.manual_sample_with_timesteps(model=cldm, device="cpu", x_T=z_partial, steps=4, used_timesteps=USED_TIMESTEPS,
                                                     batch_size=1, cond=dict(c_txt=c_txt, c_img=z_pre), uncond=None, cfg_scale=1.0,
                                                     progress=False)
        res = (cldm.vae_decode(z) + 1) / 2
        res = ref_common.wavelet_reconstruction(res, pre)[0]
    out.update(pre=pre.numpy().astype(np.float16), z_pre=z_pre.numpy(), z=z.numpy(), res=res[:, :h0, :w0].numpy())
    np.savez_compressed(os.path.join(GOLD, "demo_flow.npz"), **out)
    print("demo_flow.npz written: padded", tuple(x.shape), "restored", tuple(out["res"].shape), "range", float(res.min()), float(res.max()))


TOKEN_PROMPTS = [
    "", "a cat", "A photo of a DOG, running fast!", "remove dense noise", "high quality, 8k, ultra-detailed",
    "it's the artist's best work; they've said so", "  multiple   spaces\tand\nnewlines  ", "naïve café — déjà vu",
    "日本語のテキスト", "emoji 😀 test", "numbers 12345 and 3.14159", "under_score and-hyphen/slash",
    "&amp;lt;escaped&amp;gt; html &quot;entities&quot;", "<start_of_text> literal marker <end_of_text>", "ALL CAPS SHOUTING",
    "supercalifragilisticexpialidocious antidisestablishmentarianism", "don't won't can't I'll we'd she'm",
    "a " * 100, "x" * 300, "mixed123abc456 !!! ??? ...", "Ünïcödé ẞharp ǅ",
]


def gen_tokens():
    """Token ids of the reference tokenizer (model/open_clip/tokenizer.py) for prompts covering contractions, unicode, html
    entities, digits, long words and truncation -> tests/golden/clip_tokens.json."""
    ref_import.install_stubs()
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    from model.open_clip import tokenizer as T
    toks = T.tokenize(TOKEN_PROMPTS)
    tok = T.SimpleTokenizer()
    raw = [tok.encode(p) for p in TOKEN_PROMPTS]
    with open(os.path.join(GOLD, "clip_tokens.json"), "w") as f:
        json.dump({"prompts": TOKEN_PROMPTS, "tokenize_77": toks.tolist(), "encode": raw}, f)
    print("clip_tokens.json written:", [len(r) for r in raw])


def _img_digest(img: torch.Tensor) -> dict:
    """Compact pin of a large image tensor: stride-4 samples (fp16) + global statistics."""
    return {"samples": img[:, :, 1::4, 2::4].numpy().astype(np.float32),
            "stats": np.array([float(img.mean()), float(img.abs().mean()), float(img.abs().max()), float(img.std())])}


def gen_full():
    """The BASELINE.json configurations at FULL size on the reference (SD-2.1 widths, CPU fp32), on exactly the synthetic
    inputs bench.py feeds its workloads (`bench:*` names), so that the -m gpu tests compare bench.py's own code paths:
      * configs[1] det512: images 3 and 7 of the batch of 8 (512x512, 4 steps)               -> full_det512.npz
      * configs[3] seg1024tiled: one 1024x1024 image, tiled VAE encoder (256-px tiles), latent-tiled sampler (64/32),
        untiled decoder (demo.py:96-124)                                                      -> full_seg1024.npz
      * configs[4] det512s50: image 0, 50-step `sample` from pure noise (utils/sampler.py:206-265) -> full_s50.npz"""
    ControlLDM, Diffusion, SpacedSampler, ref_common = ref_import.import_reference()
    t0 = time.time()
    cldm, cfg = build_reference_cldm("sd21")
    print(f"sd21 reference built + synthetic weights in {time.time() - t0:.1f}s", flush=True)
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    c_txt = synth.synth_normal("bench:c_txt", (1, 77, 1024))
    eps_list = []
    hook = cldm.register_forward_hook(lambda m, i, o: eps_list.append(o.detach().clone()))

    only = os.environ.get("EDTR_GOLD_FULL_ONLY", "det512,s50,seg1024").split(",")
    if "det512" in only:
        # ---- configs[1]: images 3 and 7 of the bench batch
        GB, S, h = 8, 512, 64
        sel = [3, 7]
        pre = synth.synth_input("bench:pre_res", (GB, 3, S, S), 0.0, 1.0)[sel]
        noises = [synth.synth_normal(f"bench:noise{i}", (GB, 4, h, h))[sel] for i in range(5)]
        out = {"images": np.array(sel)}
        with torch.no_grad():
            t0 = time.time()
            z_pre = cldm.vae_encode(pre * 2 - 1, sample=False)
            x_T = diffusion.q_sample(z_pre, torch.full((len(sel),), 200, dtype=torch.int64), noises[0])
            sampler = SpacedSampler(diffusion.betas)
            eps_list.clear()
            with injected_noise(noises[1:]):
                z = sampler.manual_sample_with_timesteps(
                    model=cldm, device="cpu", x_T=x_T, steps=4, used_timesteps=USED_TIMESTEPS, batch_size=len(sel),
                    cond={"c_txt": c_txt.expand(len(sel), -1, -1), "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False)
            img = cldm.vae_decode(z)
            print(f"det512 x{len(sel)}: {time.time() - t0:.1f}s", flush=True)
        out.update(z_pre=z_pre.numpy(), z=z.numpy(), img_samples=_img_digest(img)["samples"], img_stats=_img_digest(img)["stats"])
        for i, e in enumerate(eps_list):
            out[f"eps{i}"] = e.numpy()
        np.savez_compressed(os.path.join(GOLD, "full_det512.npz"), **out)
        print("full_det512.npz written", flush=True)

    if "s50" in only:
        # ---- configs[4]: image 0 of the batch of 4, 50 spaced steps from pure noise
        GB = 4
        pre = synth.synth_input("bench:pre_res", (GB, 3, S, S), 0.0, 1.0)[:1]
        x_T = synth.synth_normal("bench:noise0", (GB, 4, h, h))[:1]
        step_noise = [synth.synth_normal(f"bench:s50noise{i}", (GB, 4, h, h))[:1] for i in range(50)]
        out = {}
        with torch.no_grad():
            t0 = time.time()
            z_pre = cldm.vae_encode(pre * 2 - 1, sample=False)
            sampler = SpacedSampler(diffusion.betas)
            eps_list.clear()
            with injected_noise(step_noise):
                z, inter = sampler.sample(model=cldm, device="cpu", steps=50, batch_size=1, x_size=(4, h, h),
                                          cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, x_T=x_T,
                                          progress=False, return_intermediates=True)
            img = cldm.vae_decode(z)
            print(f"det512s50 x1: {time.time() - t0:.1f}s", flush=True)
        out.update(z_pre=z_pre.numpy(), z=z.numpy(), img_samples=_img_digest(img)["samples"], img_stats=_img_digest(img)["stats"])
        for i in (0, 9, 24, 39, 49):
            out[f"pred_x0_{i}"] = inter[i].numpy()
            out[f"eps{i}"] = eps_list[i].numpy()
        np.savez_compressed(os.path.join(GOLD, "full_s50.npz"), **out)
        print("full_s50.npz written", flush=True)

    if "seg1024" in only:
        # ---- configs[3]: 1024x1024, tiled encoder / latent-tiled sampler / untiled decoder
        S, h = 1024, 128
        pre = synth.synth_input("bench:pre_res", (1, 3, S, S), 0.0, 1.0)
        noises = [synth.synth_normal(f"bench:noise{i}", (1, 4, h, h)) for i in range(5)]
        out = {}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            t0 = time.time()
            z_pre = cldm.vae_encode(pre * 2 - 1, sample=False, tiled=True, tile_size=256)
            t_enc = time.time() - t0
            x_T = diffusion.q_sample(z_pre, torch.full((1,), 200, dtype=torch.int64), noises[0])
            sampler = SpacedSampler(diffusion.betas)
            eps_list.clear()
            with injected_noise(noises[1:]):
                z = sampler.manual_sample_with_timesteps(
                    model=cldm, device="cpu", x_T=x_T, steps=4, used_timesteps=USED_TIMESTEPS, batch_size=1,
                    cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False,
                    tiled=True, tile_size=64, tile_stride=32)
            t_smp = time.time() - t0 - t_enc
            img = cldm.vae_decode(z)
        print(f"seg1024tiled: enc {t_enc:.1f}s sampler {t_smp:.1f}s total {time.time() - t0:.1f}s", flush=True)
        out.update(z_pre=z_pre.numpy(), z=z.numpy(), img_samples=_img_digest(img)["samples"], img_stats=_img_digest(img)["stats"])
        hook.remove()
        np.savez_compressed(os.path.join(GOLD, "full_seg1024.npz"), **out)
        print("full_seg1024.npz written", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="schedule,tiny,sd21,tiled,tiledvae,vaesample,heavy,wavelet,clip,psnr,swinir,full,tokens")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    todo = args.only.split(",")
    for name in todo:
        {"schedule": gen_schedule, "tiny": gen_tiny, "sd21": gen_sd21, "tiled": gen_tiled, "tiledvae": gen_tiledvae,
         "vaesample": gen_vaesample, "heavy": gen_heavy, "wavelet": gen_wavelet, "clip": gen_clip, "psnr": gen_psnr, "swinir": gen_swinir, "full": gen_full, "tokens": gen_tokens, "demo": gen_demo, "moderate": gen_moderate}[name]()


if __name__ == "__main__":
    main()
