"""End-to-end parity on the MI355X (pytest -m gpu): the reference-API flow of demo.py:102-124 /
main/det/test_edtr.py:121-135 run through edtr_amd (HIP kernels via the C ABI) against the committed goldens that
the REFERENCE produced on CPU fp32 (tests/golden/tiny_pipeline*.npz) with identical weights, inputs and noise.

Stated tolerances, relative L2 error vs the fp32 reference (16-bit activations, fp32 accumulation; four denoise
steps through 2 x 25 residual blocks and 23 transformer blocks each):
    fp16 storage: eps per step 4e-3, final latent 4e-3, decoded image 6e-3
    bf16 storage: eps per step 3e-2, final latent 3e-2, decoded image 4e-2"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

USED = [50, 100, 150, 200]
# <= 1.5 x the measured values (fp16: z_pre 1.47e-3, eps 2.43e-3, z 8.5e-4, img 1.49e-3; bf16: 1.18e-2, 1.96e-2, 6.7e-3, 1.18e-2)
TOL = {torch.float16: dict(z_pre=2.2e-3, eps=3.6e-3, z=1.3e-3, img=2.2e-3),
       torch.bfloat16: dict(z_pre=1.75e-2, eps=2.9e-2, z=1.0e-2, img=1.77e-2)}


def _run(golden_dir, name, tag, B, H, W, dtype):
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, name))
    cfg = synth.tiny_config()
    cldm = build_synthetic_cldm(cfg, dev, dtype)
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)
    pre_res = synth.synth_input(f"{tag}:pre_res", (B, 3, H, W), 0.0, 1.0).to(dev)
    c_txt = synth.synth_input(f"{tag}:c_txt", (B, 77, 64), -1.0, 1.0).to(dev)
    noises = [synth.synth_normal(f"{tag}:noise{i}", (B, 4, H // 8, W // 8)).to(dev) for i in range(5)]
    z_pre = cldm.vae_encode(pre_res * 2 - 1, sample=False)
    x_T = diffusion.q_sample(z_pre, torch.full((B,), 200, dtype=torch.int64, device=dev), noises[0])
    eps_log = []
    orig_forward = cldm.forward

    def logging_forward(x, t, cond, woSD=False):
        e = orig_forward(x, t, cond)
        eps_log.append(e.clone())
        return e

    cldm.forward = logging_forward
    with injected_noise(noises[1:]):
        z, inter = sampler.manual_sample_with_timesteps(
            model=cldm, device=dev, x_T=x_T, steps=4, used_timesteps=USED, batch_size=B,
            cond={"c_txt": c_txt, "c_img": z_pre}, uncond=None, cfg_scale=1.0, progress=False, return_intermediates=True)
    img = cldm.vae_decode(z)
    torch.cuda.synchronize()
    errs = {"z_pre": rel_err(z_pre, g["z_pre"]), "x_T": rel_err(x_T, g["x_T"]), "z": rel_err(z, g["z"]),
            "img": rel_err(img, g["img"])}
    for i in range(4):
        errs[f"eps{i}"] = rel_err(eps_log[i], g[f"eps{i}"])
        errs[f"pred_x0_{i}"] = rel_err(inter[i], g[f"pred_x0_{i}"])
    print(f"\n[{name} {dtype}] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    return errs


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,tag,B,H,W", [("tiny_pipeline.npz", "tiny", 2, 128, 128),
                                            ("tiny_pipeline_rect.npz", "tinyrect", 1, 192, 128)])
def test_tiny_pipeline_vs_reference_golden(golden_dir, name, tag, B, H, W, dtype):
    errs = _run(golden_dir, name, tag, B, H, W, dtype)
    tol = TOL[dtype]
    assert errs["z_pre"] < tol["z_pre"]
    assert errs["x_T"] < tol["z_pre"]
    for i in range(4):
        assert errs[f"eps{i}"] < tol["eps"], (i, errs)
    assert errs["z"] < tol["z"]
    assert errs["img"] < tol["img"]


def test_cpu_module_fails_loudly():
    """No silent fallback: a CPU-resident model refuses to run."""
    from edtr_amd import synth
    from edtr_amd.model import ControlLDM
    m = ControlLDM(**synth.tiny_config())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.vae_encode(torch.zeros(1, 3, 64, 64), sample=False)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_latent_tiled_sampler_vs_reference_golden(golden_dir, dtype):
    """cfg-4 style latent tiling (tile 8 / stride 4 on a 16x24 latent, 15 tiles per step) through the reference-shaped
    sampler API; the golden is the REFERENCE's tiled output (tiling is a different function, not an optimisation)."""
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "tiled.npz"))
    cldm = build_synthetic_cldm(synth.tiny_config(), dev, dtype)
    sampler = SpacedSampler(Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).betas)
    B, h, w = 1, 16, 24
    x_T = synth.synth_normal("tiled:x_T", (B, 4, h, w)).to(dev)
    c_img = synth.synth_normal("tiled:c_img", (B, 4, h, w)).to(dev)
    c_txt = synth.synth_input("tiled:c_txt", (B, 77, 64), -1.0, 1.0).to(dev)
    noises = [synth.synth_normal(f"tiled:noise{i}", (B, 4, h, w)).to(dev) for i in range(4)]
    with injected_noise(noises):
        z = sampler.manual_sample_with_timesteps(
            model=cldm, device=dev, x_T=x_T, steps=4, used_timesteps=USED, batch_size=B,
            cond={"c_txt": c_txt, "c_img": c_img}, uncond=None, cfg_scale=1.0, tiled=True, tile_size=8, tile_stride=4,
            progress=False)
    torch.cuda.synchronize()
    err = rel_err(z, g["z_tiled"])
    print(f"\n[tiled sampler {dtype}] z={err:.2e}")
    assert err < TOL[dtype]["z"]
    assert "forward" in cldm.__dict__          # like the reference, the patched forward is never restored


def test_sample_api_and_cfg_vs_oracle():
    """DiffBIR-style `sample(steps=N)` (respaced from 1000, reference utils/sampler.py:206-265) against the CPU oracle,
    plus the classifier-free-guidance mix of predict_noise (:178-181)."""
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler, space_timesteps
    from edtr_amd.testing import build_synthetic_cldm, injected_noise, rel_err, synthetic_state_dicts
    from oracle import edtr_oracle as O
    from oracle import flat_sd as flat_oracle_sd
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cfg = synth.tiny_config()
    sds = synthetic_state_dicts(cfg)
    cldm = build_synthetic_cldm(cfg, dev, torch.float16, sds)
    betas = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).betas
    sampler = SpacedSampler(betas)
    B, h, w, steps = 2, 8, 8, 5
    x_T = synth.synth_normal("sample:x_T", (B, 4, h, w))
    c_img = synth.synth_normal("sample:c_img", (B, 4, h, w))
    c_txt = synth.synth_input("sample:c_txt", (B, 77, 64), -1.0, 1.0)
    noises = [synth.synth_normal(f"sample:noise{i}", (B, 4, h, w)) for i in range(steps)]
    with torch.no_grad():
        ref = O.sample(flat_oracle_sd(sds), cfg, O.make_betas(), x_T, sorted(space_timesteps(1000, str(steps))),
                       {"c_txt": c_txt, "c_img": c_img}, noises)
    with injected_noise(noises):
        z = sampler.sample(model=cldm, device=dev, steps=steps, batch_size=B, x_size=(4, h, w),
                           cond={"c_txt": c_txt.to(dev), "c_img": c_img.to(dev)}, uncond=None, cfg_scale=1.0,
                           x_T=x_T.to(dev), progress=False)
    torch.cuda.synchronize()
    assert rel_err(z, ref) < 5e-3
    # CFG algebra with a stub model: uncond + s (cond - uncond)
    e_c, e_u = synth.synth_normal("cfg:c", (B, 4, h, w)).to(dev), synth.synth_normal("cfg:u", (B, 4, h, w)).to(dev)

    def stub(x, t, cond):
        return e_c if cond == "c" else e_u

    out = sampler.predict_noise(stub, x_T.to(dev), None, "c", "u", 2.5)
    torch.cuda.synchronize()
    assert rel_err(out, e_u + 2.5 * (e_c - e_u)) < 1e-6


def test_wavelet_colour_fix_vs_reference_golden(golden_dir):
    from edtr_amd import synth
    from edtr_amd.testing import rel_err
    from edtr_amd.wavelet import wavelet_decomposition, wavelet_reconstruction
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "wavelet.npz"))
    a = synth.synth_input("wav:content", (2, 3, 96, 80), 0.0, 1.0).to(dev)
    b = synth.synth_input("wav:style", (2, 3, 96, 80), 0.0, 1.0).to(dev)
    hi, lo = wavelet_decomposition(a)
    rec = wavelet_reconstruction(a, b)
    torch.cuda.synchronize()
    assert rel_err(lo, g["low"]) < 1e-6
    assert rel_err(hi, g["high"]) < 1e-5
    assert rel_err(rec, g["recon"]) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tiled_vae_vs_reference_golden(golden_dir, dtype):
    """VAEHook paths (a21): tiled encode (tile 64 px -> 6 padded tiles) and tiled decode (tile 8 latent px -> 6 tiles) with
    GroupNorm statistics pooled across tiles, against the REFERENCE's tiled outputs (which differ from the untiled ones by
    6 % / 2 %, so this is not satisfied by the plain VAE)."""
    from edtr_amd import synth
    from edtr_amd.testing import build_synthetic_cldm, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "tiled_vae.npz"))
    cldm = build_synthetic_cldm(synth.tiny_config(), dev, dtype)
    img = synth.synth_input("tvae:img", (1, 3, 192, 256), -1.0, 1.0).to(dev)
    zin = synth.synth_normal("tvae:z", (1, 4, 32, 40)).to(dev)
    z_t = cldm.vae_encode(img, sample=False, tiled=True, tile_size=64)
    d_t = cldm.vae_decode(zin, tiled=True, tile_size=8)
    z_p = cldm.vae_encode(img, sample=False)
    small = cldm.vae_encode(img[:, :, :128, :128].contiguous(), sample=False, tiled=True, tile_size=64)   # falls back to untiled
    small_p = cldm.vae_encode(img[:, :, :128, :128].contiguous(), sample=False)
    torch.cuda.synchronize()
    e_enc, e_dec, e_plain = rel_err(z_t, g["z_tiled"]), rel_err(d_t, g["dec_tiled"]), rel_err(z_p, g["z_plain"])
    print(f"\n[tiled vae {dtype}] enc={e_enc:.2e} dec={e_dec:.2e} plain_enc={e_plain:.2e} "
          f"(reference tiled-vs-plain gap: enc {rel_err(g['z_tiled'], g['z_plain']):.2e})")
    tol = TOL[dtype]
    assert e_enc < tol["z_pre"] and e_plain < tol["z_pre"]
    assert e_dec < tol["img"]
    assert rel_err(small, small_p) == 0.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sd21_width_networks_vs_reference_golden(golden_dir, dtype):
    """The FULL SD-2.1-width ControlNet + UNet (one denoise step at the true hot shape, latent 64x64, every real channel
    count: 320/640/1280, 5/10/20 heads, 1024-wide context) and the full-width VAE encoder / decoder against outputs of the
    REFERENCE modules on CPU fp32 (tests/golden/sd21_blocks.npz, tools/make_goldens.py).  Exercises the automatic tile
    choice (128x128, 128x160, 256x256 kernels), split-K, fused GroupNorm statistics and both attention shapes."""
    from edtr_amd import synth
    from edtr_amd.testing import build_synthetic_cldm, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "sd21_blocks.npz"))
    cldm = build_synthetic_cldm(synth.sd21_config(), dev, dtype)
    x = synth.synth_normal("sd21:x", (1, 4, 64, 64)).to(dev)
    c_img = synth.synth_normal("sd21:c_img", (1, 4, 64, 64)).to(dev)
    c_txt = synth.synth_input("sd21:c_txt", (1, 77, 1024), -1.0, 1.0).to(dev)
    t = torch.tensor([200], device=dev)
    eps = cldm.forward(x, t, {"c_txt": c_txt, "c_img": c_img})
    img = synth.synth_input("sd21:img", (1, 3, 256, 256), -1.0, 1.0).to(dev)
    z = cldm.vae_encode(img, sample=False)
    zin = synth.synth_normal("sd21:zdec", (1, 4, 32, 32)).to(dev)
    dec = cldm.vae_decode(zin)
    torch.cuda.synchronize()
    errs = {"eps": rel_err(eps, g["eps"]), "vae_z": rel_err(z, g["vae_z"]), "vae_dec": rel_err(dec, g["vae_dec"])}
    print(f"\n[sd21 widths {dtype}] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    tol = TOL[dtype]
    assert errs["eps"] < tol["eps"] and errs["vae_z"] < tol["z_pre"] and errs["vae_dec"] < tol["img"]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tag", ["small", "vith"])
def test_clip_text_tower_vs_reference_golden(golden_dir, tag, dtype):
    """FrozenOpenCLIPEmbedder on the GPU (edtr_embed_tokens, layernorm, igemm with exact-GELU epilogue, causal
    edtr_flash_attn64) against the reference module's output on the same synthetic weights: empty prompt, short prompt,
    a full 77-token row and a half-length row (causal mask + ragged last key tile)."""
    from edtr_amd import synth
    from edtr_amd.model.clip import FrozenOpenCLIPEmbedder, clip_text_param_spec
    from edtr_amd.model.params import skip_init
    from edtr_amd.testing import rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "clip_text.npz"))
    cfg = synth.clip_small_config() if tag == "small" else synth.sd21_config()["clip_cfg"]
    with skip_init():
        m = FrozenOpenCLIPEmbedder(**cfg)
    m.load_state_dict({k: synth.synth_param(f"clip{tag}." + k, shp)
                       for k, shp in clip_text_param_spec(cfg["embed_dim"], cfg["text_cfg"])}, strict=True)
    m = m.eval().to(dev)
    m.compute_dtype = dtype
    tokens = torch.from_numpy(g["tokens"]).to(dev)
    ref = g["z_small"] if tag == "small" else g["z_vith"].astype(np.float32)
    z = m(tokens if tag == "small" else tokens[:2])
    torch.cuda.synchronize()
    err = rel_err(z, ref)
    print(f"\n[clip {tag} {dtype}] rel err {err:.2e}")
    assert err < (4e-3 if dtype == torch.float16 else 3e-2)
    # encode([""]) goes through the vocabulary-free tokenizer path and equals row 0
    if tag == "small":
        z0 = m.encode([""])
        assert rel_err(z0[0], z[0]) < 1e-6


def test_restore_dataset_driver_shapes_and_psnr():
    """evalutil.restore_dataset (accelerate-free driver of main/det/test_edtr.py:121-135): ragged images are padded to the
    model size, restored in batches, colour-fixed, cropped back and scored."""
    from edtr_amd import evalutil, synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cldm = build_synthetic_cldm(synth.tiny_config(), dev, torch.float16)
    cldm.clip.set_embedding(synth.synth_input("drv:c_txt", (1, 77, 64), -1.0, 1.0).to(dev))
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)
    imgs = [synth.synth_input(f"drv:img{i}", (3, h, w), 0.0, 1.0) for i, (h, w) in enumerate([(128, 128), (96, 120), (64, 40)])]
    torch.manual_seed(0)
    outs, psnr = evalutil.restore_dataset(cldm, diffusion, sampler, imgs, gts=imgs, img_size=128, batch_size=2)
    assert [tuple(o.shape) for o in outs] == [tuple(i.shape) for i in imgs]
    assert all(torch.isfinite(o).all() and float(o.min()) >= 0.0 and float(o.max()) <= 1.0 for o in outs)
    assert torch.isfinite(psnr) and 0.0 < float(psnr) < 60.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_demo_flow_through_the_harness_vs_reference_golden(golden_dir, dtype):
    """SURVEY §8 row f4 with VALUES: the demo flow of demo.py:84-131,165 — a 150 x 100 image,
    pad_if_smaller -> pad_to_multiples_of(64) -> SwinIR -> vae_encode -> q_sample(200) -> 4 steps -> vae_decode ->
    wavelet_reconstruction -> crop — through evalutil.restore_dataset(pad_mode="demo") on the GPU against what the REFERENCE's own
    functions produced on the same weights, input and noise (tools/make_goldens.py gen_demo -> tests/golden/demo_flow.npz).
    Stated tolerance on the restored image (relative L2): fp16 4e-3, bf16 3e-2 (<= 1.5 x measured)."""
    from edtr_amd import evalutil, synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.model.swinir import SwinIR
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "demo_flow.npz"))
    cfg = synth.tiny_config()
    cldm = build_synthetic_cldm(cfg, dev, dtype)
    cldm.clip.set_embedding(synth.synth_input("demo:c_txt", (1, 77, cfg["unet_cfg"]["context_dim"]), -1.0, 1.0).to(dev))
    sw = SwinIR(**synth.swinir_small_config())
    sw.load_state_dict({k: (synth.synth_param("swinirsmall." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                        for k, v in sw.state_dict().items()}, strict=True)
    sw = sw.eval().to(dev)
    sw.compute_dtype = dtype
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)
    img = synth.synth_input("demo:lq", (3, 150, 100), 0.0, 1.0)
    noises = [synth.synth_normal(f"demo:noise{i}", tuple(g["z_pre"].shape)) for i in range(5)]
    with injected_noise(noises):
        outs, psnr = evalutil.restore_dataset(cldm, diffusion, sampler, [img], gts=[img], img_size=128, swinir=sw, pad_mode="demo",
                                              multiple=64, clamp=False)
    torch.cuda.synchronize()
    assert len(outs) == 1 and tuple(outs[0].shape) == (3, 150, 100) == tuple(g["res"].shape)
    err = rel_err(outs[0], g["res"])
    print(f"\n[demo flow {dtype}] restored image rel err {err:.2e}, PSNR vs the input {float(psnr):.2f} dB")
    assert err < (4e-3 if dtype == torch.float16 else 3e-2)
    # the clamped form is what the harness returns by default; its PSNR against the input is the reference's own number
    want = evalutil.calculate_psnr_pt(torch.from_numpy(g["res"])[None].clamp(0, 1), img[None], crop_border=0)[0]
    with injected_noise(noises):
        outs_c, psnr_c = evalutil.restore_dataset(cldm, diffusion, sampler, [img], gts=[img], img_size=128, swinir=sw, pad_mode="demo")
    assert float(outs_c[0].min()) >= 0.0 and float(outs_c[0].max()) <= 1.0
    assert abs(float(psnr_c) - float(want)) < (0.05 if dtype == torch.float16 else 0.3)


def test_full_size_batch_invariance_and_determinism():
    """Size-independent properties at the BASELINE workload's full size (SD-2.1 widths, 512x512 images, 4 steps): the
    restoration of an image does not depend on the batch it travels in (batch 3 vs batch 1: different tile choices and
    workgroup counts, same arithmetic up to 16-bit rounding), and a replay of the same batch is bit-identical."""
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cldm = build_synthetic_cldm(synth.sd21_config(), dev, torch.bfloat16)
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)
    B, S = 3, 512
    pre = synth.synth_input("full:pre_res", (B, 3, S, S), 0.0, 1.0).to(dev)
    c_txt = synth.synth_normal("full:c_txt", (1, 77, 1024)).to(dev)
    noises = [synth.synth_normal(f"full:noise{i}", (B, 4, S // 8, S // 8)).to(dev) for i in range(5)]

    def run(sel):
        n = len(sel)
        z_pre = cldm.vae_encode(pre[sel] * 2 - 1, sample=False)
        x_T = diffusion.q_sample(z_pre, torch.full((n,), 200, dtype=torch.int64), noises[0][sel])
        with injected_noise([nz[sel] for nz in noises[1:]]):
            z = sampler.manual_sample_with_timesteps(model=cldm, device=dev, x_T=x_T, steps=4, used_timesteps=USED,
                                                     batch_size=n, cond={"c_txt": c_txt.expand(n, -1, -1).contiguous(), "c_img": z_pre},
                                                     uncond=None, cfg_scale=1.0, progress=False)
        return z, cldm.vae_decode(z)

    z3, img3 = run([0, 1, 2])
    z3b, img3b = run([0, 1, 2])
    assert torch.equal(z3, z3b) and torch.equal(img3, img3b)                    # deterministic replay
    z1, img1 = run([1])
    ez, ei = rel_err(z3[1:2], z1), rel_err(img3[1:2], img1)
    print(f"\n[full size] batch-3 vs batch-1: latent {ez:.2e}, image {ei:.2e}")
    assert ez < 8e-3 and ei < 1.85e-2          # measured 5.2e-3 / 1.23e-2 (bf16 rounding: tile choice / split-K depend on M)
    assert torch.isfinite(img3).all() and float(img3.abs().max()) < 50.0


@pytest.mark.parametrize("precision,dtype", [("fast", torch.bfloat16), ("mixed", None)])
def test_batch_invariant_mode_is_bit_exact_across_batch_sizes(monkeypatch, precision, dtype):
    """EDTR_AMD_BATCH_INVARIANT=1: tile geometry / split-K / statistics fusion are chosen from the layer's per-image shape, never
    from the row count, so an image's restoration is BIT-identical whatever batch it travels in (and therefore however a data
    set is sharded over GPUs).  SD-2.1 widths, 256x256 images (32x32 latents: all four UNet levels), 4 steps."""
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, injected_noise
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    monkeypatch.setenv("EDTR_AMD_BATCH_INVARIANT", "1")
    dev = torch.device("cuda:0")
    cldm = build_synthetic_cldm(synth.sd21_config(), dev, dtype, precision=precision)
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)
    B, S = 5, 256
    pre = synth.synth_input("inv:pre_res", (B, 3, S, S), 0.0, 1.0).to(dev)
    c_txt = synth.synth_normal("inv:c_txt", (1, 77, 1024)).to(dev)
    noises = [synth.synth_normal(f"inv:noise{i}", (B, 4, S // 8, S // 8)).to(dev) for i in range(5)]

    def run(sel):
        n = len(sel)
        z_pre = cldm.vae_encode(pre[sel] * 2 - 1, sample=False)
        x_T = diffusion.q_sample(z_pre, torch.full((n,), 200, dtype=torch.int64), noises[0][sel])
        with injected_noise([nz[sel] for nz in noises[1:]]):
            z = sampler.manual_sample_with_timesteps(model=cldm, device=dev, x_T=x_T, steps=4, used_timesteps=USED,
                                                     batch_size=n, cond={"c_txt": c_txt.expand(n, -1, -1).contiguous(), "c_img": z_pre},
                                                     uncond=None, cfg_scale=1.0, progress=False)
        return z_pre, z, cldm.vae_decode(z)

    zp5, z5, img5 = run([0, 1, 2, 3, 4])
    zp1, z1, img1 = run([3])
    zp2, z2, img2 = run([3, 0])
    assert torch.isfinite(img5).all()
    assert torch.equal(zp5[3:4], zp1) and torch.equal(z5[3:4], z1) and torch.equal(img5[3:4], img1)
    assert torch.equal(zp2[0:1], zp1) and torch.equal(z2[0:1], z1) and torch.equal(img2[0:1], img1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tag", ["small", "full"])
def test_swinir_vs_reference_golden(golden_dir, tag, dtype):
    """SwinIR pre-restoration (edtr_pixel_unshuffle, padded edtr_layernorm, edtr_igemm incl. the LeakyReLU epilogue and the
    fused nearest-x2 convs, edtr_window_attn) against the reference class's output on the same synthetic weights: a 2 x 2-layer
    network on a non-square batch (eager launch list AND hipGraph replay), and the shipped 8 x 6-layer network at 256^2 / 512^2.
    Stated tolerance (relative L2 of the output image): fp16 3e-3, bf16 2e-2 — measured 7e-4 / 5e-3."""
    from edtr_amd import synth
    from edtr_amd.model.swinir import SwinIR
    from edtr_amd.testing import rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "swinir.npz"))
    cfg = synth.swinir_small_config() if tag == "small" else synth.swinir_config()
    m = SwinIR(**cfg)
    sd = m.state_dict()
    m.load_state_dict({k: (synth.synth_param(f"swinir{tag}." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                       for k, v in sd.items()}, strict=True)
    m = m.eval().to(dev)
    m.compute_dtype = dtype
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    if tag == "small":
        x = synth.synth_input("swinir:small", (2, 3, 128, 192), 0.0, 1.0).to(dev)
        outs = []
        for graph in (False, True):
            m.use_graph = graph
            m._engines.clear()
            y = m(x)
            torch.cuda.synchronize()
            outs.append(y)
            err = rel_err(y, g["y_small"])
            print(f"\n[swinir small {dtype} graph={graph}] rel err {err:.2e}")
            assert y.shape == (2, 3, 128, 192) and err < tol
        assert torch.equal(outs[0], outs[1])                       # graph replay == eager launch list, bit for bit
        # an image of a batch does not depend on its batch mates
        y1 = m(x[1:])
        assert rel_err(y1, outs[1][1:]) < 1e-6
        # input that is not a multiple of the window: reflect pad, padded size returned (reference quirk, model/swinir.py:834-839,894)
        yp = m(synth.synth_input("swinir:odd", (1, 3, 60, 124), 0.0, 1.0).to(dev))
        assert yp.shape == (1, 3, 64, 128) and rel_err(yp, g["y_small_padded"]) < tol
    else:
        y = m(synth.synth_input("swinir:256", (1, 3, 256, 256), 0.0, 1.0).to(dev))
        err = rel_err(y, g["y_256"].astype(np.float32))
        y5 = m(synth.synth_input("swinir:512", (1, 3, 512, 512), 0.0, 1.0).to(dev))
        err5 = rel_err(y5[:, :, 3::8, 5::8], g["y_512_stride8"])
        print(f"\n[swinir full {dtype}] rel err 256^2 {err:.2e}, 512^2 (stride-8 samples) {err5:.2e}")
        assert err < tol and err5 < tol
        np.testing.assert_allclose(float(y5.mean()), g["y_512_stats"][0], rtol=5e-3)


def test_restore_dataset_with_pre_restoration():
    """The whole demo.py:89-124 chain on the device: low-quality images -> SwinIR -> vae_encode -> q_sample -> 4 steps ->
    vae_decode -> colour fix, through evalutil.restore_dataset(swinir=...).  The pre-restoration stage must be what the
    driver feeds the path: the result equals running SwinIR by hand and passing its output as `pre_restored`."""
    from edtr_amd import evalutil, synth
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.model.swinir import SwinIR
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import build_synthetic_cldm, rel_err
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cldm = build_synthetic_cldm(synth.tiny_config(), dev, torch.float16)
    cldm.clip.set_embedding(synth.synth_input("drv:c_txt", (1, 77, 64), -1.0, 1.0).to(dev))
    swinir = SwinIR(**synth.swinir_small_config())
    sd = swinir.state_dict()
    swinir.load_state_dict({k: (synth.synth_param("swinirsmall." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                            for k, v in sd.items()}, strict=True)
    swinir = swinir.eval().to(dev)
    swinir.compute_dtype = torch.float16
    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)
    imgs = [synth.synth_input(f"drv2:img{i}", (3, 128, 128), 0.0, 1.0) for i in range(3)]
    torch.manual_seed(0)
    outs, _ = evalutil.restore_dataset(cldm, diffusion, sampler, imgs, img_size=128, batch_size=2, swinir=swinir)
    by_hand = [swinir(i[None].to(dev))[0] for i in imgs]
    torch.manual_seed(0)
    outs2, _ = evalutil.restore_dataset(cldm, diffusion, sampler, by_hand, img_size=128, batch_size=2)
    assert all(tuple(o.shape) == (3, 128, 128) and torch.isfinite(o).all() for o in outs)
    assert all(rel_err(a, b) < 2e-3 for a, b in zip(outs, outs2))


def test_tiled_pre_restoration():
    """demo.py:96-98 (`--pre-res-tiled`): make_tiled_fn(swinir, size, stride) = SwinIR per sliding window + Gaussian-weighted
    overlap-add, against the same overlap-add done by hand in torch from the per-window outputs."""
    from edtr_amd import synth
    from edtr_amd.model.swinir import SwinIR
    from edtr_amd.testing import rel_err
    from edtr_amd.tiling import gaussian_weights, make_tiled_fn, sliding_windows
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    swinir = SwinIR(**synth.swinir_small_config())
    sd = swinir.state_dict()
    swinir.load_state_dict({k: (synth.synth_param("swinirsmall." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                            for k, v in sd.items()}, strict=True)
    swinir = swinir.eval().to(dev)
    swinir.compute_dtype = torch.float16
    x = synth.synth_input("swinir:tiled", (1, 3, 128, 192), 0.0, 1.0).to(dev)
    y = make_tiled_fn(swinir, size=64, stride=32)(x)
    torch.cuda.synchronize()
    w = torch.tensor(gaussian_weights(64, 64), dtype=torch.float32)
    num, den = torch.zeros(1, 3, 128, 192), torch.zeros(1, 3, 128, 192)
    wins = sliding_windows(128, 192, 64, 32)
    assert len(wins) == 15
    for hi, he, wi, we in wins:
        num[..., hi:he, wi:we] += swinir(x[..., hi:he, wi:we].contiguous()).cpu() * w
        den[..., hi:he, wi:we] += w
    assert y.shape == x.shape and rel_err(y, num / den) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["stride2", "splitk", "fusable"])
def test_deferred_groupnorm_falls_back_to_its_apply_launch(case):
    """`group_norm(..., conv_n=N)` defers the apply to the consuming convolution on the caller's promise of what that
    convolution is; when another one arrives (stride 2; a shape whose split-K the halo tile does not take) `Emitter.conv` emits the apply
    launch itself instead of raising — same numbers as the undeferred program (reference model/vae.py:103-114: norm -> swish -> conv)."""
    from edtr_amd.engine import Act, Arena, Emitter, Program, WeightStore
    dev = torch.device("cuda:0")
    B, H, C, N = 2, 80, 128, 128          # 50 units of 16 x 16 pixels: a shape the halo tile's fused GroupNorm takes
    g = torch.Generator().manual_seed(3)
    params = {"n.weight": torch.rand(C, generator=g) + 0.5, "n.bias": torch.randn(C, generator=g) * 0.1,
              "c.weight": torch.randn(N, C, 3, 3, generator=g) / (3 * C ** 0.5), "c.bias": torch.randn(N, generator=g) * 0.1}
    x0 = torch.randn(B * H * H, C, generator=g).to(dev, torch.bfloat16)
    outs = []
    for defer in (False, True):
        prog, arena = Program("t"), Arena(dev, 1 << 26)
        em = Emitter(prog, arena, WeightStore(params, torch.bfloat16, dev), torch.bfloat16)
        xa = Act(arena.alloc((B * H * H, C), torch.bfloat16), B, H, H, C)
        xa.t.copy_(x0)
        kw = dict(stride=2, pad_tl=0) if case == "stride2" else {}
        h = em.group_norm(xa, "n.", 1e-6, True, conv_n=N if defer else 0)
        if defer:
            assert h.gn_in is not None and h.gn_apply is not None
            if case == "splitk":         # the promise breaks after the fact: the tensor the convolution sees is not 32-bit addressable
                h = Act(h.t, h.B, h.H, h.W, h.C, h.gnp, gn_in=h.gn_in, gn_silu=h.gn_silu, owns=h.owns, gn_apply=h.gn_apply)
                import edtr_amd.ops as ops_mod
                real = ops_mod.gn_in_conv_ok
                ops_mod.gn_in_conv_ok = lambda *a, **k: False
        try:
            y = em.conv(h, "c.", name="res.conv1", **kw)
        finally:
            if defer and case == "splitk":
                ops_mod.gn_in_conv_ok = real
        n_apply = sum(1 for r in prog.recs if r.name == "gn.apply")
        assert n_apply == (1 if (not defer or case != "fusable") else 0), (case, defer, [r.name for r in prog.recs])
        prog.run()
        torch.cuda.synchronize()
        outs.append(y.t.float().cpu())
    err = (outs[0] - outs[1]).norm() / outs[0].norm()
    assert err < (2e-2 if case == "fusable" else 1e-6), (case, float(err))
