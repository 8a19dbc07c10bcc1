"""Pin the oracle (oracle/edtr_oracle.py) to the reference: every fixture here was produced by
tools/make_goldens.py running /root/reference on CPU fp32 with edtr_amd.synth weights/inputs."""
import json
import os

import numpy as np
import pytest
import torch

from edtr_amd import synth
from oracle import edtr_oracle as O

USED = [50, 100, 150, 200]


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def synth_sd(manifest_path, parts=("unet", "controlnet", "vae")):
    with open(manifest_path) as f:
        man = json.load(f)
    sd = {}
    for part in parts:
        for key, shape in man[part]:
            full = f"{part}.{key}"
            sd[full] = synth.synth_param(full, tuple(shape))
    return sd


def test_schedule_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    betas = O.make_betas()
    np.testing.assert_array_equal(betas, g["betas"])
    t4 = O.schedule_tables(betas, USED)
    t50 = O.schedule_tables(betas, O.space_timesteps(1000, "50"))
    for name in ["sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                 "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]:
        np.testing.assert_array_equal(t4[name], g["s4_" + name])
        np.testing.assert_array_equal(t50[name], g["s50_" + name])
    np.testing.assert_array_equal(t4["timesteps"], g["s4_timesteps"])
    np.testing.assert_array_equal(t50["timesteps"], g["s50_timesteps"])
    # values quoted in SURVEY.md §8(a) row a1
    np.testing.assert_allclose(t4["sqrt_recip_alphas_cumprod"], [1.0251279, 1.0574917, 1.0990925, 1.1518688], rtol=1e-6)
    np.testing.assert_allclose(t4["posterior_mean_coef2"], [0, 0.44377658, 0.59105539, 0.6670472], rtol=1e-6)
    sa, sb = O.q_sample_coefs(betas)
    np.testing.assert_array_equal(sa, g["q_sqrt_ac"])
    np.testing.assert_array_equal(sb, g["q_sqrt_1mac"])
    np.testing.assert_allclose([sa[200], sb[200]], [0.86815441, 0.49629423], rtol=1e-6)


def test_space_timesteps(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    np.testing.assert_array_equal(O.space_timesteps(1000, "50"), g["space_1000_50"])
    np.testing.assert_array_equal(O.space_timesteps(300, [10, 15, 20]), g["space_1000_10_15_20"])
    np.testing.assert_array_equal(O.space_timesteps(1000, "ddim25"), g["space_ddim25"])
    with pytest.raises(ValueError):
        O.space_timesteps(10, [20])
    with pytest.raises(ValueError):
        O.space_timesteps(1000, "ddim999")


def test_timestep_embedding_and_updates(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    t = torch.tensor([50, 100, 150, 200, 999])
    assert rel_err(O.timestep_embedding(t, 320), g["temb_320"]) < 1e-6
    assert rel_err(O.timestep_embedding(t, 64), g["temb_64"]) < 1e-6
    x = synth.synth_normal("sched_x", (4, 4, 8, 8))
    eps = synth.synth_normal("sched_eps", (4, 4, 8, 8))
    noise = synth.synth_normal("sched_noise", (4, 4, 8, 8))
    tabs = O.schedule_tables(O.make_betas(), USED)
    x_prev, pred_x0 = O.p_sample_update(tabs, x, eps, noise, torch.tensor([0, 1, 2, 3]))
    np.testing.assert_allclose(x_prev.numpy(), g["p_sample_x_prev"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pred_x0.numpy(), g["p_sample_pred_x0"], rtol=1e-6, atol=1e-6)
    q = O.q_sample(O.make_betas(), x, torch.full((4,), 200), noise)
    np.testing.assert_allclose(q.numpy(), g["q_sample_200"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name,tag,B,H,W", [("tiny_pipeline.npz", "tiny", 2, 128, 128),
                                            ("tiny_pipeline_rect.npz", "tinyrect", 1, 192, 128)])
def test_tiny_pipeline(golden_dir, name, tag, B, H, W):
    g = np.load(os.path.join(golden_dir, name))
    cfg = synth.tiny_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_tiny.json"))
    pre_res = synth.synth_input(f"{tag}:pre_res", (B, 3, H, W), 0.0, 1.0)
    c_txt = synth.synth_input(f"{tag}:c_txt", (B, 77, 64), -1.0, 1.0)
    noises = [synth.synth_normal(f"{tag}:noise{i}", (B, 4, H // 8, W // 8)) for i in range(5)]
    with torch.no_grad():
        img, tr = O.restore(sd, cfg, O.make_betas(), pre_res, c_txt, noises, USED, 200, return_trace=True)
        res = O.wavelet_reconstruction((img + 1) / 2, pre_res)
    assert rel_err(tr["z_pre"], g["z_pre"]) < 1e-5
    assert rel_err(tr["x_T"], g["x_T"]) < 1e-5
    for i in range(4):
        assert rel_err(tr["eps"][i], g[f"eps{i}"]) < 2e-5, i
        assert rel_err(tr["pred_x0"][i], g[f"pred_x0_{i}"]) < 2e-5, i
    assert rel_err(tr["z"], g["z"]) < 2e-5
    assert rel_err(img, g["img"]) < 2e-5
    assert rel_err(res, g["res_wavelet"]) < 2e-5


def test_tiny_controls(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_pipeline.npz"))
    cfg = synth.tiny_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_tiny.json"), parts=("controlnet",))
    x_T = torch.from_numpy(g["x_T"])
    z_pre = torch.from_numpy(g["z_pre"])
    c_txt = synth.synth_input("tiny:c_txt", (2, 77, 64), -1.0, 1.0)
    with torch.no_grad():
        ctrl = O.controlnet_forward(sd, cfg["controlnet_cfg"], x_T, z_pre, torch.full((2,), 200), c_txt, p="controlnet.")
    assert len(ctrl) == 13
    stats = np.array([[float(c.mean()), float(c.abs().mean()), float(c.abs().max())] for c in ctrl])
    np.testing.assert_allclose(stats, g["ctrl_stats"], rtol=2e-4, atol=1e-5)
    for i in (0, 3, 6, 12):
        assert rel_err(ctrl[i], g[f"ctrl{i}"].astype(np.float32)) < 2e-3  # stored as fp16


def test_tiled_paths(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiled.npz"))
    np.testing.assert_allclose(O.gaussian_weights(64, 64), g["gauss_64"], rtol=1e-12)
    np.testing.assert_allclose(O.gaussian_weights(8, 8), g["gauss_8x8"], rtol=1e-12)
    np.testing.assert_array_equal(np.array(O.sliding_windows(128, 128, 64, 32)), g["win_128_128_64_32"])
    np.testing.assert_array_equal(np.array(O.sliding_windows(72, 96, 64, 32)), g["win_72_96_64_32"])
    np.testing.assert_array_equal(np.array(O.sliding_windows(16, 24, 8, 4)), g["win_16_24_8_4"])
    cfg = synth.tiny_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_tiny.json"), parts=("unet", "controlnet"))
    B, h, w = 1, 16, 24
    x_T = synth.synth_normal("tiled:x_T", (B, 4, h, w))
    c_img = synth.synth_normal("tiled:c_img", (B, 4, h, w))
    c_txt = synth.synth_input("tiled:c_txt", (B, 77, 64), -1.0, 1.0)
    noises = [synth.synth_normal(f"tiled:noise{i}", (B, 4, h, w)) for i in range(4)]
    with torch.no_grad():
        z = O.sample(sd, cfg, O.make_betas(), x_T, USED, {"c_txt": c_txt, "c_img": c_img}, noises,
                     tiled=True, tile_size=8, tile_stride=4)
    assert rel_err(z, g["z_tiled"]) < 2e-5


def test_wavelet(golden_dir):
    g = np.load(os.path.join(golden_dir, "wavelet.npz"))
    a = synth.synth_input("wav:content", (2, 3, 96, 80), 0.0, 1.0)
    b = synth.synth_input("wav:style", (2, 3, 96, 80), 0.0, 1.0)
    assert rel_err(O.wavelet_reconstruction(a, b), g["recon"]) < 1e-6
    hi, lo = O.wavelet_decomposition(a)
    assert rel_err(hi, g["high"]) < 1e-5
    assert rel_err(lo, g["low"]) < 1e-6


@pytest.mark.slow
def test_sd21_blocks(golden_dir):
    """Full SD-2.1-width networks at the true hot shape (latent 64x64): pins every real channel count."""
    g = np.load(os.path.join(golden_dir, "sd21_blocks.npz"))
    cfg = synth.sd21_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_sd21.json"))
    x = synth.synth_normal("sd21:x", (1, 4, 64, 64))
    c_img = synth.synth_normal("sd21:c_img", (1, 4, 64, 64))
    c_txt = synth.synth_input("sd21:c_txt", (1, 77, 1024), -1.0, 1.0)
    t = torch.tensor([200])
    with torch.no_grad():
        ctrl = O.controlnet_forward(sd, cfg["controlnet_cfg"], x, c_img, t, c_txt, p="controlnet.")
        stats = np.array([[float(c.mean()), float(c.abs().mean()), float(c.abs().max())] for c in ctrl])
        np.testing.assert_allclose(stats, g["ctrl_stats"], rtol=5e-4, atol=2e-5)
        assert rel_err(ctrl[12], g["ctrl12"]) < 5e-5
        eps = O.unet_forward(sd, cfg["unet_cfg"], x, t, c_txt, ctrl, p="unet.")
        assert rel_err(eps, g["eps"]) < 5e-5
        img = synth.synth_input("sd21:img", (1, 3, 256, 256), -1.0, 1.0)
        assert rel_err(O.vae_encode(sd, cfg, img), g["vae_z"]) < 5e-5
        zin = synth.synth_normal("sd21:zdec", (1, 4, 32, 32))
        assert rel_err(O.vae_decode(sd, cfg, zin), g["vae_dec"]) < 5e-5


def test_tiled_vae(golden_dir):
    """VAEHook restatement: tile geometry and the cross-tile GroupNorm pooling vs the reference's tiled outputs."""
    g = np.load(os.path.join(golden_dir, "tiled_vae.npz"))
    for name, (h, w, ts, dec) in {"enc_1024_256": (1024, 1024, 256, False), "enc_192x256_64": (192, 256, 64, False),
                                  "dec_128_64": (128, 128, 64, True), "dec_32x40_8": (32, 40, 8, True),
                                  "enc_560x760_256": (560, 760, 256, False)}.items():
        ib, ob = O.split_tiles(h, w, ts, dec)
        np.testing.assert_array_equal(np.array(ib), g[f"bbox_in_{name}"])
        np.testing.assert_array_equal(np.array(ob), g[f"bbox_out_{name}"])
    cfg = synth.tiny_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_tiny.json"), parts=("vae",))
    img = synth.synth_input("tvae:img", (1, 3, 192, 256), -1.0, 1.0)
    zin = synth.synth_normal("tvae:z", (1, 4, 32, 40))
    with torch.no_grad():
        assert rel_err(O.vae_encode_tiled(sd, cfg, img, 64), g["z_tiled"]) < 2e-5
        assert rel_err(O.vae_decode_tiled(sd, cfg, zin, 8), g["dec_tiled"]) < 2e-5
        assert rel_err(O.vae_encode(sd, cfg, img), g["z_plain"]) < 2e-5
        # small inputs fall back to the plain network (tilevae.py:317-323)
        small = synth.synth_input("tvae:small", (1, 3, 128, 128), -1.0, 1.0)
        assert rel_err(O.vae_encode_tiled(sd, cfg, small, 64), O.vae_encode(sd, cfg, small)) == 0.0


def test_vae_encode_sample_default(golden_dir):
    """vae_encode's DEFAULT path (sample=True): the oracle's restatement of DiagonalGaussianDistribution.sample() against the
    reference's own seeded call (tools/make_goldens.py gen_vaesample; the reference draws on the host generator)."""
    g = np.load(os.path.join(golden_dir, "vae_sample.npz"))
    cfg = synth.tiny_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_tiny.json"), parts=("vae",))
    img = synth.synth_input("vsample:img", (2, 3, 64, 96), -1.0, 1.0)
    torch.manual_seed(int(g["seed"][0]))
    noise = torch.randn(tuple(g["z_sample"].shape))
    with torch.no_grad():
        assert rel_err(O.vae_encode_sample(sd, cfg, img, noise), g["z_sample"]) < 2e-5
        assert rel_err(O.vae_encode(sd, cfg, img), g["z_mode"]) < 2e-5


def test_tiny_pipeline_heavy_tailed_weights(golden_dir):
    """The oracle against the reference on the HEAVY-TAILED weight set (outlier channels x12, norm gains +-4, sharper attention:
    edtr_amd.synth.synth_param_heavy) — the fixture the GPU range-robustness tests compare with (tests/test_gpu_heavy.py)."""
    from edtr_amd.testing import synthetic_state_dicts
    from oracle import flat_sd
    g = np.load(os.path.join(golden_dir, "heavy.npz"))
    cfg = synth.tiny_config()
    sd = flat_sd(synthetic_state_dicts(cfg, None, "heavy"))
    B, H, W = 2, 128, 128
    pre_res = synth.synth_input("heavy:pre_res", (B, 3, H, W), 0.0, 1.0)
    c_txt = synth.synth_input("heavy:c_txt", (B, 77, 64), -1.0, 1.0)
    noises = [synth.synth_normal(f"heavy:noise{i}", (B, 4, H // 8, W // 8)) for i in range(5)]
    with torch.no_grad():
        img, tr = O.restore(sd, cfg, O.make_betas(), pre_res, c_txt, noises, USED, 200, return_trace=True)
    errs = {"z_pre": rel_err(tr["z_pre"], g["z_pre"]), "eps0": rel_err(tr["eps"][0], g["eps0"]), "eps3": rel_err(tr["eps"][3], g["eps3"]),
            "z": rel_err(tr["z"], g["z"]), "img": rel_err(img, g["img"])}
    assert all(v < 2e-4 for v in errs.values()), errs      # (measured <= 6.6e-5: this net amplifies fp32 rounding ~10x more than the smooth set)


def _clip_sd(tag, cfg):
    from edtr_amd.model.clip import clip_text_param_spec
    return {"clip." + k: synth.synth_param(f"clip{tag}." + k, shp) for k, shp in clip_text_param_spec(cfg["embed_dim"], cfg["text_cfg"])}


def test_clip_text_small(golden_dir):
    """CLIP text tower restatement (causal pre-LN transformer, penultimate layer, ln_final) vs the reference's
    FrozenOpenCLIPEmbedder on a small tower with ViT-H's head width."""
    g = np.load(os.path.join(golden_dir, "clip_text.npz"))
    cfg = synth.clip_small_config()
    tokens = synth.clip_test_tokens()
    np.testing.assert_array_equal(tokens.numpy(), g["tokens"])
    with torch.no_grad():
        z = O.clip_text_forward(_clip_sd("small", cfg), cfg["text_cfg"], tokens, layer_idx=1)
    assert rel_err(z, g["z_small"]) < 2e-5


@pytest.mark.slow
def test_clip_text_vith(golden_dir):
    """The full ViT-H text tower (24 layers, width 1024, 16 heads) of configs/det/demo.yaml."""
    g = np.load(os.path.join(golden_dir, "clip_text.npz"))
    cfg = synth.sd21_config()["clip_cfg"]
    tokens = synth.clip_test_tokens()
    with torch.no_grad():
        z = O.clip_text_forward(_clip_sd("vith", cfg), cfg["text_cfg"], tokens[:2], layer_idx=1)
    assert rel_err(z, g["z_vith"].astype(np.float32)) < 1e-3          # golden stored as fp16
    np.testing.assert_allclose([float(z.abs().mean())], [g["stats_vith"][1]], rtol=2e-2)


def _swinir_sd(tag, cfg):
    from edtr_amd.model.swinir import swinir_state_spec
    return {"swinir." + k: synth.synth_param(f"swinir{tag}." + k, shp) for k, shp, kind in swinir_state_spec(cfg) if kind == "param"}


def test_swinir_small(golden_dir):
    """SwinIR restatement vs the reference class on a 2x2-layer network, non-square input (2 x 3 windows: every shifted-window
    mask case), plus the relative-position index and shift masks against the reference's own buffers."""
    g = np.load(os.path.join(golden_dir, "swinir.npz"))
    cfg = synth.swinir_small_config()
    np.testing.assert_array_equal(O.swin_relative_index(8), g["rel_index"].astype(np.int64))
    for name, (h, w) in (("mask_64x64", (64, 64)), ("mask_16x24", (16, 24))):
        m = O.swin_shift_mask(h, w, 8, 4)
        np.testing.assert_array_equal(np.packbits(m != 0), g[name])
        assert set(np.unique(m)) == {0.0, float(g["mask_value"][0])}
    x = synth.synth_input("swinir:small", (2, 3, 128, 192), 0.0, 1.0)
    with torch.no_grad():
        y = O.swinir_forward(_swinir_sd("small", cfg), cfg, x)
    assert y.shape == (2, 3, 128, 192)
    assert rel_err(y, g["y_small"]) < 2e-5
    # not a multiple of the window: reflect pad in image space, and the PADDED size comes back (model/swinir.py:834-839,894)
    with torch.no_grad():
        yp = O.swinir_forward(_swinir_sd("small", cfg), cfg, synth.synth_input("swinir:odd", (1, 3, 60, 124), 0.0, 1.0))
    assert yp.shape == (1, 3, 64, 128) and rel_err(yp, g["y_small_padded"]) < 2e-5


def test_swinir_full(golden_dir):
    """The shipped 8 x 6-layer, 180-channel network of configs/det/demo.yaml at 256^2 (full output) and 512^2 (samples)."""
    g = np.load(os.path.join(golden_dir, "swinir.npz"))
    cfg = synth.swinir_config()
    sd = _swinir_sd("full", cfg)
    with torch.no_grad():
        y256 = O.swinir_forward(sd, cfg, synth.synth_input("swinir:256", (1, 3, 256, 256), 0.0, 1.0))
        y512 = O.swinir_forward(sd, cfg, synth.synth_input("swinir:512", (1, 3, 512, 512), 0.0, 1.0))
    assert rel_err(y256, g["y_256"].astype(np.float32)) < 1e-3            # golden stored as fp16
    assert rel_err(y512[:, :, 3::8, 5::8], g["y_512_stride8"]) < 5e-5
    np.testing.assert_allclose([float(y512.mean()), float(y512.abs().mean()), float(y512.abs().max()), float(y512.std())],
                               g["y_512_stats"], rtol=1e-4)


@pytest.mark.slow
def test_full_size_det512_image_vs_reference(golden_dir):
    """BASELINE configs[1] at full size (SD-2.1 widths, 512x512, 4 steps): the oracle on image 7 of bench.py's batch against the
    reference's own run on the same inputs (tests/golden/full_det512.npz) — this is the oracle bench.py times as `cpu_baseline`
    and checks the GPU result against, at the size it is used."""
    g = np.load(os.path.join(golden_dir, "full_det512.npz"))
    cfg = synth.sd21_config()
    sd = synth_sd(os.path.join(golden_dir, "manifest_sd21.json"))
    k = int(g["images"][1])
    pre = synth.synth_input("bench:pre_res", (8, 3, 512, 512), 0.0, 1.0)[k:k + 1]
    c_txt = synth.synth_normal("bench:c_txt", (1, 77, 1024))
    noises = [synth.synth_normal(f"bench:noise{i}", (8, 4, 64, 64))[k:k + 1] for i in range(5)]
    with torch.no_grad():
        img, tr = O.restore(sd, cfg, O.make_betas(), pre, c_txt, noises, USED, 200, return_trace=True)
    assert rel_err(tr["z_pre"], g["z_pre"][1:2]) < 2e-5
    assert rel_err(tr["z"], g["z"][1:2]) < 5e-5
    assert rel_err(img[:, :, 1::4, 2::4], g["img_samples"][1:2]) < 5e-5
