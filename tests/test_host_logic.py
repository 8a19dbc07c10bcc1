"""Host-side logic that needs no GPU: schedules, respacing, tile geometry, state-dict compatibility with the
reference manifests, weight packing, arena, split-K heuristic, error behaviour."""
import json
import os

import numpy as np
import pytest
import torch

from edtr_amd import arch, ops, synth
from edtr_amd.diffusion import Diffusion, make_beta_schedule
from edtr_amd.engine import Arena
from edtr_amd.model import ControlLDM
from edtr_amd.model.params import skip_init
from edtr_amd.parallel import bucketize, shard_slice
from edtr_amd.sampler import SpacedSampler, space_timesteps
from edtr_amd.tiling import gaussian_weights, sliding_windows

TABLES = ["sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
          "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]


def test_diffusion_and_sampler_tables_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    d = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    np.testing.assert_array_equal(d.betas, g["betas"])
    np.testing.assert_array_equal(d.sqrt_alphas_cumprod.numpy(), g["q_sqrt_ac"])
    np.testing.assert_array_equal(d.sqrt_one_minus_alphas_cumprod.numpy(), g["q_sqrt_1mac"])
    s = SpacedSampler(d.betas)
    s.make_schedule(4, [50, 100, 150, 200])
    for n in TABLES:
        np.testing.assert_array_equal(getattr(s, n).numpy(), g["s4_" + n])
    np.testing.assert_array_equal(s.timesteps, g["s4_timesteps"])
    a, b, c1, c2, sigma = s._coefs(0)
    assert sigma == 0.0 and c1 == 1.0 and c2 == 0.0           # last step: x_prev = pred_x0, noise masked
    s.make_schedule(50)
    for n in TABLES:
        np.testing.assert_array_equal(getattr(s, n).numpy(), g["s50_" + n])
    np.testing.assert_array_equal(s.timesteps, g["s50_timesteps"])
    s.make_schedule(1, [200])
    assert float(s.posterior_log_variance_clipped[0]) == -10.0


def test_space_timesteps_and_errors(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    np.testing.assert_array_equal(sorted(space_timesteps(1000, "50")), g["space_1000_50"])
    np.testing.assert_array_equal(sorted(space_timesteps(300, [10, 15, 20])), g["space_1000_10_15_20"])
    np.testing.assert_array_equal(sorted(space_timesteps(1000, "ddim25")), g["space_ddim25"])
    with pytest.raises(ValueError):
        space_timesteps(10, [20])
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(ValueError):
        make_beta_schedule("nope", 10)


def test_vae_tile_geometry(golden_dir):
    from edtr_amd.nets import split_tiles
    g = np.load(os.path.join(golden_dir, "tiled_vae.npz"))
    for name, (h, w, ts, dec) in {"enc_1024_256": (1024, 1024, 256, False), "enc_192x256_64": (192, 256, 64, False),
                                  "dec_128_64": (128, 128, 64, True), "dec_32x40_8": (32, 40, 8, True),
                                  "enc_560x760_256": (560, 760, 256, False)}.items():
        ib, ob = split_tiles(h, w, ts, dec)
        np.testing.assert_array_equal(np.array(ib), g[f"bbox_in_{name}"])
        np.testing.assert_array_equal(np.array(ob), g[f"bbox_out_{name}"])
    assert len(split_tiles(1024, 1024, 256, False)[0]) == 16          # BASELINE config 4: 16 encoder tiles


def test_tiling_geometry(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiled.npz"))
    np.testing.assert_allclose(gaussian_weights(64, 64), g["gauss_64"], rtol=1e-12)
    np.testing.assert_array_equal(np.array(sliding_windows(128, 128, 64, 32)), g["win_128_128_64_32"])
    np.testing.assert_array_equal(np.array(sliding_windows(72, 96, 64, 32)), g["win_72_96_64_32"])
    np.testing.assert_array_equal(np.array(sliding_windows(16, 24, 8, 4)), g["win_16_24_8_4"])
    assert sliding_windows(64, 64, 64, 32) == [(0, 64, 0, 64)]          # single window when the tile covers the latent


@pytest.mark.parametrize("name", ["tiny", "sd21"])
def test_state_dict_spec_equals_reference_manifest(golden_dir, name):
    man = json.load(open(os.path.join(golden_dir, f"manifest_{name}.json")))
    cfg = synth.CONFIGS[name]()
    specs = {"unet": arch.unet_param_spec(arch.unet_arch(cfg["unet_cfg"])),
             "controlnet": arch.unet_param_spec(arch.unet_arch(cfg["controlnet_cfg"], controlnet=True)),
             "vae": arch.vae_param_spec(cfg["vae_cfg"])}
    for part, spec in specs.items():
        assert [[k, list(s)] for k, s in spec] == man[part], part       # names, shapes AND order


def test_strict_checkpoint_loading_tiny():
    cfg = synth.tiny_config()
    with skip_init():
        m = ControlLDM(**cfg)
    from edtr_amd.testing import synthetic_state_dicts
    sds = synthetic_state_dicts(cfg)
    sd_ckpt = {f"model.diffusion_model.{k}": v for k, v in sds["unet"].items()}
    sd_ckpt.update({f"first_stage_model.{k}": v for k, v in sds["vae"].items()})
    clip_sd = {k: torch.zeros_like(v) for k, v in m.clip.state_dict().items()}
    clip_sd["model.positional_embedding"] = torch.full((77, 64), 0.25)
    sd_ckpt.update({f"cond_stage_model.{k}": v for k, v in clip_sd.items()})
    sd_ckpt["model_ema.decay"] = torch.zeros(())
    unused = m.load_pretrained_sd(sd_ckpt)
    assert unused == {"model_ema.decay"}                                 # unet / vae / clip keys all consumed (cldm.py:46-78)
    assert float(m.clip.state_dict()["model.positional_embedding"][3, 5]) == 0.25
    missing = dict(sd_ckpt)
    missing.pop("cond_stage_model.model.ln_final.bias")
    with pytest.raises(KeyError):
        m.load_pretrained_sd(missing)
    assert torch.equal(m.unet.state_dict()["out.2.weight"], sds["unet"]["out.2.weight"])
    m.load_controlnet_from_ckpt(sds["controlnet"])
    bad = dict(sds["controlnet"])
    bad.pop("zero_convs.0.0.weight")
    with pytest.raises(RuntimeError):
        m.load_controlnet_from_ckpt(bad)                                # strict=True, as the reference
    zero_keys, scratch_keys = m.load_controlnet_from_unet()
    assert zero_keys == {"input_blocks.0.0.weight"}
    w = m.controlnet.state_dict()["input_blocks.0.0.weight"]
    assert w.shape[1] == 8 and float(w[:, 4:].abs().max()) == 0.0       # hint half zero-initialised
    assert all(k.startswith(("zero_convs.", "middle_block_out.")) for k in scratch_keys)
    m.vae.decoder.load_state_dict({k[len("decoder."):]: v for k, v in sds["vae"].items() if k.startswith("decoder.")},
                                  strict=True)                          # demo.py:53 call shape


def test_unsupported_configs_and_cpu_inputs_raise():
    cfg = synth.tiny_config()
    with pytest.raises(NotImplementedError):
        arch.unet_arch(dict(cfg["unet_cfg"], num_head_channels=32))
    with pytest.raises(NotImplementedError):
        ControlLDM(**cfg, tail_block=True)
    with skip_init():
        m = ControlLDM(**cfg)
    x = torch.zeros(1, 4, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.zeros(1, dtype=torch.long), {"c_txt": torch.zeros(1, 77, 64), "c_img": x})
    with pytest.raises(RuntimeError, match="no CPU fallback"):            # the CLIP text tower, too, is GPU-only
        m.clip.encode([""])
    m.clip.set_embedding(torch.ones(1, 77, 64))                          # optional precomputed-embedding override
    assert m.clip.encode(["", ""]).shape == (2, 77, 64)


def test_weight_packing():
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = ops.pack_conv_weight(w, torch.float16, cin_pad=8)
    assert p.shape == (8, 72)
    assert float(p[1, (1 * 3 + 2) * 8 + 2]) == float(w[1, 2, 1, 2])      # [cout][ky][kx][cin]
    assert float(p[:, 3:8].abs().max()) == 0.0 and float(p[2:].abs().max()) == 0.0
    perm = ops.geglu_perm(64)
    assert perm[:32].tolist() == list(range(32)) and perm[32:64].tolist() == list(range(64, 96))
    assert perm[64:96].tolist() == list(range(32, 64)) and sorted(perm.tolist()) == list(range(128))


def test_splitk_heuristic_and_arena():
    assert ops.choose_splitk(32768, 320, 2880) == (0, 1)                 # fills the chip already
    tile, s = ops.choose_splitk(512, 1280, 11520)
    assert tile == 0 and 4 <= s <= 8
    assert ops.choose_splitk(512, 1280, 11520, act=1) == (0, 1)          # GEGLU epilogue cannot be split
    a = Arena(torch.device("cpu"), chunk_bytes=1 << 16)
    t1 = a.alloc((100, 8), torch.float32)
    t2 = a.alloc((64,), torch.float16)
    p1 = t1.data_ptr()
    a.free(t1[:, :4])            # a view never frees its owner
    assert p1 in a.live
    a.free(t1)
    t3 = a.alloc((50, 8), torch.float32)
    assert t3.data_ptr() == p1   # first fit reuses the hole
    a.free(t2), a.free(t3)
    assert a.in_use == 0 and len(a.free_lists[0]) == 1


def test_batch_invariant_launch_choices_do_not_depend_on_the_batch(monkeypatch):
    """EDTR_AMD_BATCH_INVARIANT=1 (engine.Emitter): tile geometry comes from the operand layout, the fused-statistics
    eligibility from the per-image shape — never from the row count M = B * H * W."""
    from edtr_amd import ops
    monkeypatch.delenv("EDTR_AMD_BATCH_INVARIANT", raising=False)
    assert not ops.batch_invariant()
    monkeypatch.setenv("EDTR_AMD_BATCH_INVARIANT", "1")
    assert ops.batch_invariant()
    assert ops.invariant_tile(320, 0) == 3 and ops.invariant_tile(4, 0) == 1 and ops.invariant_tile(320, 320) == 1
    for hw, N, C1 in [(4096, 320, 320), (1024, 640, 8), (256, 1280, 1280), (16, 1280, 1280), (262144, 128, 3)]:
        got = {ops.gn_fusable(B * hw, N, C1, hw, invariant=True) for B in (1, 2, 3, 8)}
        assert len(got) == 1, (hw, N, C1, got)
    # the default path DOES look at M for the register-staged shapes (documented: tolerance-level batch dependence)
    assert ops.gn_fusable(8 * 1024, 640, 8, 1024) != ops.gn_fusable(1 * 1024, 640, 8, 1024)


def test_batch_sharding_helpers():
    cover = []
    for r in range(8):
        s = shard_slice(r, 8, 64)
        cover += list(range(64))[s]
    assert cover == list(range(64))
    sizes = [len(range(10)[shard_slice(r, 4, 10)]) for r in range(4)]
    assert sizes == [3, 3, 2, 2]
    with pytest.raises(ValueError):
        shard_slice(4, 4, 8)
    ts = [torch.zeros(n) for n in (10, 20, 30, 40)]
    assert [len(b) for b in bucketize(ts, 100)] == [2, 1, 1]      # 40+80 B, 120 B, 160 B


def test_clip_state_dict_spec_equals_reference_manifest(golden_dir):
    """FrozenOpenCLIPEmbedder keys / shapes / order equal the reference module's state_dict (strict loading of
    `cond_stage_model.*`), for the small tower and for ViT-H."""
    import json
    from edtr_amd import synth
    from edtr_amd.model.clip import clip_text_param_spec
    with open(os.path.join(golden_dir, "manifest_clip.json")) as f:
        man = json.load(f)
    for tag, cfg in (("small", synth.clip_small_config()), ("vith", synth.sd21_config()["clip_cfg"])):
        spec = [[k, list(shp)] for k, shp in clip_text_param_spec(cfg["embed_dim"], cfg["text_cfg"])]
        assert spec == man[tag]


def test_clip_tokenizer_empty_prompt_and_truncation():
    from edtr_amd.model.clip import tokenize
    t = tokenize(["", "  "])
    assert t.shape == (2, 77) and t.dtype == torch.int64
    assert t[0, :3].tolist() == [49406, 49407, 0] and int(t[0].sum()) == 49406 + 49407 and bool((t[0] == t[1]).all())
    with pytest.raises(RuntimeError):
        tokenize(["a cat"], bpe_path="/nonexistent/bpe.gz")


def test_eval_helpers_vs_reference_golden(golden_dir):
    """PSNR / YCbCr (utils/common.py:194-247) against the reference's numbers; list_to_batch / batch_to_list round trip."""
    from edtr_amd import evalutil
    g = np.load(os.path.join(golden_dir, "psnr.npz"))
    a = synth.synth_input("psnr:a", (3, 3, 40, 56), 0.0, 1.0)
    b = (a + 0.05 * synth.synth_normal("psnr:n", (3, 3, 40, 56))).clamp(0, 1)
    np.testing.assert_array_equal(evalutil.calculate_psnr_pt(a, b, 0, False).numpy(), g["psnr_0"])
    np.testing.assert_array_equal(evalutil.calculate_psnr_pt(a, b, 4, True).numpy(), g["psnr_4y"])
    np.testing.assert_array_equal(evalutil.rgb2ycbcr_pt(a).numpy(), g["ycbcr"])
    imgs = [a[0, :, :30, :56], a[1, :, :40, :17], a[2]]
    batch = evalutil.list_to_batch(imgs, 64, "cpu")
    assert batch.shape == (3, 3, 64, 64) and float(batch[0, :, 30:, :].abs().max()) == 0.0 and float(batch[1, :, :, 17:].abs().max()) == 0.0
    back = evalutil.batch_to_list(batch, imgs)
    assert all(torch.equal(x, y) for x, y in zip(back, imgs))


def test_swinir_state_dict_spec_equals_reference_manifest(golden_dir):
    """Keys, shapes, dtypes AND order of the reference SwinIR state dict (parameters and the two registered buffers), so that
    `load_state_dict(torch.load("swinir_last.pt"), strict=True)` (main/det/test_edtr.py:45) works on edtr_amd's class."""
    from edtr_amd.model.swinir import SwinIR, swinir_state_spec
    with open(os.path.join(golden_dir, "manifest_swinir.json")) as f:
        man = json.load(f)
    for tag, cfg in (("small", synth.swinir_small_config()), ("full", synth.swinir_config())):
        kinds = {"param": "float32", "mask": "float32", "index": "int64"}
        spec = [[k, list(s), kinds[kind]] for k, s, kind in swinir_state_spec(cfg)]
        assert spec == man[tag]
    m = SwinIR(**synth.swinir_small_config())
    sd = m.state_dict()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()] == man["small"]
    # strict load of a reference-shaped checkpoint, buffers included
    ck = {k: (synth.synth_param("swinirsmall." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v.clone())
          for k, v in sd.items()}
    m.load_state_dict(ck, strict=True)
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in ck.items() if not k.endswith("conv_last.bias")}, strict=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(NotImplementedError):
        SwinIR(**{**synth.swinir_small_config(), "upsampler": "pixelshuffle"})


def test_swinir_packing_helpers(golden_dir):
    """Host-side tables of the device path: the expanded bias, the packed qkv projection (heads widened to 32 columns) and the
    region labels reproduce the reference's buffers / a plain fp32 evaluation."""
    from edtr_amd.model import swinir as S
    g = np.load(os.path.join(golden_dir, "swinir.npz"))
    np.testing.assert_array_equal(S.relative_position_index(8), g["rel_index"].astype(np.int64))
    np.testing.assert_array_equal(np.packbits(S.shift_mask(64, 64, 8, 4) != 0), g["mask_64x64"])
    np.testing.assert_array_equal(np.packbits(S.shift_mask(16, 24, 8, 4) != 0), g["mask_16x24"])
    lab = S.region_labels(16, 24, 8, 4)
    win = lab.reshape(2, 8, 3, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    np.testing.assert_array_equal((win[:, None, :] != win[:, :, None]), S.shift_mask(16, 24, 8, 4) != 0)
    heads, d, C, cp = 6, 30, 180, 192
    w, b = synth.synth_param("t.qkv.weight", (3 * C, C)), synth.synth_param("t.qkv.bias", (3 * C,))
    wp, bp = S.pack_qkv(w, b, heads, cp)
    assert wp.shape == (3 * heads * 32, cp) and bp.shape == (3 * heads * 32,)
    x = synth.synth_input("t.x", (5, C))
    xp = torch.zeros(5, cp)
    xp[:, :C] = x
    got = (xp @ wp.T + bp).reshape(5, 3, heads, 32)
    ref = (x @ w.T + b).reshape(5, 3, heads, d)
    assert torch.allclose(got[..., :d], ref, atol=1e-6) and float(got[..., d:].abs().max()) == 0.0
    table = synth.synth_param("t.table", (225, heads))
    bias = S.expand_bias(table, 8)
    idx = torch.from_numpy(S.relative_position_index(8))
    assert bias.shape == (heads, 64, 64) and float(bias[3, 10, 50]) == float(table[idx[10, 50], 3])


def test_swin_mlp_weight_images():
    """pack_swin_mlp_weights writes the LDS images edtr_swin_mlp documents (include/edtr_hip.h): chunk c of row r at slot
    c ^ key(r), element for element, and every MFMA operand read of the kernel (16 lanes = 16 rows, one chunk) touches 16
    different 16-byte bank slots."""
    from edtr_amd import ops
    C, HID = ops.SWIN_MLP_C, ops.SWIN_MLP_HIDDEN
    w1 = (torch.arange(HID * C, dtype=torch.float32).reshape(HID, C) % 2039) / 8.0          # exactly representable in fp16
    w2 = (torch.arange(C * HID, dtype=torch.float32).reshape(C, HID) % 2029) / 8.0
    i1, i2 = ops.pack_swin_mlp_weights(w1, w2, torch.float16)
    assert i1.numel() == HID * C and i2.numel() == C * HID
    b1, b2 = i1.view(torch.uint8).numpy(), i2.view(torch.uint8).numpy()
    h1, h2 = w1.to(torch.float16).numpy(), w2.to(torch.float16).numpy()
    rng = np.random.default_rng(0)
    for _ in range(400):
        t, r, c, j = rng.integers(HID // 32), rng.integers(32), rng.integers(C // 8), rng.integers(8)
        off = t * 12288 + r * 384 + ((c ^ ((r >> 1) & 7)) << 4) + 2 * j
        assert b1[off:off + 2].view(np.float16)[0] == h1[32 * t + r, 8 * c + j]
        r2, c2 = rng.integers(C), rng.integers(4)
        off = t * 12288 + r2 * 64 + ((c2 ^ ((r2 >> 2) & 3)) << 4) + 2 * j
        assert b2[off:off + 2].view(np.float16)[0] == h2[r2, 32 * t + 8 * c2 + j]
    for c in range(C // 8):          # W1 tile: the 16 rows a lane group reads (any permutation of 16 consecutive rows), chunk c
        for r0 in (0, 16):
            assert len({((r * 384 + ((c ^ ((r >> 1) & 7)) << 4)) >> 4) & 15 for r in range(r0, r0 + 16)}) == 16
    for c in range(4):               # W2 slice
        for r0 in range(0, C, 16):
            assert len({((r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) >> 4) & 15 for r in range(r0, r0 + 16)}) == 16


def test_round4_weight_images_and_tables():
    """The LDS images and tables of the round-4 kernels, element for element against the layouts include/edtr_hip.h documents:
    edtr_swin_attn (per-head q / k / v tiles, proj slices, the key-group-major bias table), edtr_conv64 and edtr_conv128_out (nine tap
    matrices); and every ds_read_b128 lane group of the kernels' operand reads lands on 16 different 16-byte bank slots."""
    from edtr_amd import ops
    from edtr_amd.model import swinir as S
    rng = np.random.default_rng(1)
    H, CP = ops.SWIN_ATTN_HEADS, ops.SWIN_MLP_C
    wq = (torch.arange(3 * H * 32 * CP, dtype=torch.float32).reshape(3 * H * 32, CP) % 2027) / 8.0
    wp = (torch.arange(CP * H * 32, dtype=torch.float32).reshape(CP, H * 32) % 2017) / 8.0
    iq, ip = ops.pack_swin_attn_weights(wq, wp, torch.float16)
    bq, bp = iq.view(torch.uint8).numpy(), ip.view(torch.uint8).numpy()
    hq, hp = wq.to(torch.float16).numpy(), wp.to(torch.float16).numpy()
    for _ in range(300):
        h, s_, r, c, j = rng.integers(H), rng.integers(3), rng.integers(32), rng.integers(CP // 8), rng.integers(8)
        off = (3 * h + s_) * 12288 + r * 384 + ((c ^ ((r >> 1) & 7)) << 4) + 2 * j
        assert bq[off:off + 2].view(np.float16)[0] == hq[s_ * H * 32 + h * 32 + r, 8 * c + j]        # pack_qkv's row order is (s, h, e)
        r2, c2 = rng.integers(CP), rng.integers(4)
        off = h * 12288 + r2 * 64 + ((c2 ^ ((r2 >> 2) & 3)) << 4) + 2 * j
        assert bp[off:off + 2].view(np.float16)[0] == hp[r2, 32 * h + 8 * c2 + j]
    bias = torch.arange(H * 64 * 64, dtype=torch.float32).reshape(H, 64, 64)
    bt = ops.swin_attn_bias(bias)
    assert bt.shape == (H, 16, 64, 4) and float(bt[3, 5, 17, 2]) == float(bias[3, 17, 4 * 5 + 2])
    # edtr_conv64: [64 n][64 k] per tap, chunk c of row n in slot c ^ ((n >> 1) & 7)
    w64 = (torch.arange(48 * 64 * 9, dtype=torch.float32).reshape(48, 64, 3, 3) % 2003) / 8.0
    b64 = ops.pack_conv64_weight(w64, torch.float16).view(torch.uint8).numpy()
    h64 = w64.to(torch.float16).numpy()
    for _ in range(300):
        t, n, c, j = rng.integers(9), rng.integers(64), rng.integers(8), rng.integers(8)
        off = t * 8192 + n * 128 + ((c ^ ((n >> 1) & 7)) << 4) + 2 * j
        want = h64[n, 8 * c + j, t // 3, t % 3] if n < 48 else 0.0
        assert b64[off:off + 2].view(np.float16)[0] == want
    # edtr_conv128_out: [32 n][128 k] per tap, chunk c of row n in slot c ^ (n & 15)
    w128 = (torch.arange(3 * 128 * 9, dtype=torch.float32).reshape(3, 128, 3, 3) % 1999) / 8.0
    b128 = ops.pack_conv128_out_weight(w128, torch.float16).view(torch.uint8).numpy()
    h128 = w128.to(torch.float16).numpy()
    for _ in range(300):
        t, n, c, j = rng.integers(9), rng.integers(32), rng.integers(16), rng.integers(8)
        off = t * 8192 + n * 256 + ((c ^ (n & 15)) << 4) + 2 * j
        want = h128[n, 8 * c + j, t // 3, t % 3] if n < 3 else 0.0
        assert b128[off:off + 2].view(np.float16)[0] == want
    # bank slots (64 banks x 4 B = 16 slots of 16 B per bank row) of the lane groups of a ds_read_b128: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
    swap = lambda v: (v & ~12) | ((v & 4) << 1) | ((v & 8) >> 1)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for grp in groups:
        rows = [(l & 16) | swap(l & 15) for l in grp]                     # the weight row lane l feeds (bits 2, 3 swapped)
        for c in range(24):               # 384-byte rows (swin weight tiles, token tiles): key (r >> 1) & 7
            assert len({((r * 384 + ((c ^ ((r >> 1) & 7)) << 4)) >> 4) & 15 for r in rows}) == 16
        for c in range(4):                # 64-byte rows (swin weight slices): key (r >> 2) & 3
            assert len({((r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) >> 4) & 15 for r in rows}) == 16
        for c in range(8):                # conv64 weights: 128-byte rows, key (n >> 1) & 7
            assert len({((r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) >> 4) & 15 for r in rows}) == 16
        for c in range(16):               # conv128_out weights: 256-byte rows, key n & 15
            assert len({((r * 256 + ((c ^ (r & 15)) << 4)) >> 4) & 15 for r in rows}) == 16
        for kx in range(3):               # patch reads: lane l -> pixel (y + (l >> 4), (l & 15) + kx) of an 18-pixel-wide patch
            px = [((l >> 4) * 18 + (l & 15) + kx, (l & 15) + kx) for l in grp]
            for c in range(8):            # conv64: 128-byte pixels, key (x >> 1) & 7
                assert len({((p_ * 128 + ((c ^ ((x >> 1) & 7)) << 4)) >> 4) & 15 for p_, x in px}) == 16
            for c in range(16):           # conv128_out: 256-byte pixels, key x & 15
                assert len({((p_ * 256 + ((c ^ (x & 15)) << 4)) >> 4) & 15 for p_, x in px}) == 16


def test_engine_cache_lru():
    """Shape-keyed engine cache: hits refresh recency, the least recently used entry is evicted and released past capacity."""
    from edtr_amd.engine import EngineCache
    released, built = [], []
    cache = EngineCache(release=released.append, capacity=2)

    def build(tag):
        built.append(tag)
        return tag
    assert cache.fetch("a", lambda: build("A")) == "A"
    assert cache.fetch("b", lambda: build("B")) == "B"
    assert cache.fetch("a", lambda: build("A2")) == "A" and built == ["A", "B"]        # hit: nothing rebuilt
    assert cache.fetch("c", lambda: build("C")) == "C"
    assert list(cache) == ["a", "c"] and released == ["B"]                              # b was the least recently used
    cache.drop_all()
    assert len(cache) == 0 and sorted(released) == ["A", "B", "C"]


@pytest.mark.skipif(not os.path.exists("/root/reference/model/open_clip/bpe_simple_vocab_16e6.txt.gz"),
                    reason="needs the OpenCLIP BPE vocabulary of the reference checkout (build container only)")
def test_clip_tokenizer_token_ids_vs_reference(golden_dir):
    """Non-empty prompts: token ids equal the reference tokenizer's (tests/golden/clip_tokens.json, tools/make_goldens.py
    gen_tokens) — contractions, unicode letters / digits, html entities, special markers, long words, truncation at 77."""
    import json
    from edtr_amd.model.clip import SimpleTokenizer, tokenize
    path = "/root/reference/model/open_clip/bpe_simple_vocab_16e6.txt.gz"
    with open(os.path.join(golden_dir, "clip_tokens.json")) as f:
        g = json.load(f)
    tok = SimpleTokenizer(path)
    for prompt, want in zip(g["prompts"], g["encode"]):
        assert tok.encode(prompt) == want, prompt
    got = tokenize(g["prompts"], 77, bpe_path=path)
    assert got.tolist() == g["tokenize_77"]


def test_subpixel_weight_packing_equals_upsample_then_conv():
    """ops.pack_conv_weight_subpixel (include/edtr_hip.h: w_phase_stride): four 2x2 convolutions of the source image with the
    pre-summed weights reproduce `F.interpolate(x, 2, "nearest")` + 3x3 conv (reference model/unet.py:70-79, model/vae.py:35-39)
    exactly in fp32 arithmetic — the index algebra the halo kernel's sub-pixel geometry relies on, incl. the image borders."""
    import torch.nn.functional as F
    from edtr_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, H, W, N = 2, 8, 5, 7, 16
    x = torch.randn((B, C, H, W), generator=g, dtype=torch.float64)
    w = torch.randn((N, C, 3, 3), generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    packed = ops.pack_conv_weight_subpixel(w.float(), ops.MIXED, cin_pad=C, parts=3)           # [4 N][2][2][hi | hi | lo] fp16
    assert packed.shape == (4 * N, 4 * 3 * C)
    p5 = packed.reshape(4, N, 2, 2, 3, C).double()
    wsum = p5[..., 0, :] + p5[..., 2, :]                                                   # hi + lo = the fp32 sums to ~22 bits
    out = torch.zeros_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))                                                            # source pixel (s, r) at xp[s + 1, r + 1]
    for py in (0, 1):
        for px in (0, 1):
            acc = torch.zeros((B, N, H, W), dtype=torch.float64)
            for dy in (0, 1):
                for dx in (0, 1):
                    src = xp[:, :, py + dy:py + dy + H, px + dx:px + dx + W]               # src[s + py - 1 + dy, r + px - 1 + dx]
                    acc += torch.einsum("bchw,nc->bnhw", src, wsum[2 * py + px, :, dy, dx])
            out[:, :, py::2, px::2] = acc
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-6
    # one-part packing keeps the layout and rounds the SUMS once
    p1 = ops.pack_conv_weight_subpixel(w.float(), torch.bfloat16, cin_pad=C).reshape(4, N, 2, 2, C)
    want = (w[:, :, 1, 1] + w[:, :, 1, 2] + w[:, :, 2, 1] + w[:, :, 2, 2]).float().to(torch.bfloat16)   # phase (0, 0), tap (1, 1)
    assert torch.equal(p1[0, :, 1, 1], want)
    assert ops.subpixel_ok(64, 64, 512, 512, 8) and not ops.subpixel_ok(8, 8, 1280, 1280, 8) and not ops.subpixel_ok(64, 64, 320, 320, 8)


def test_params_fingerprint_sees_a_replaced_parameter_object():
    """replacing a Parameter that is NOT the first one (`sub.weight = nn.Parameter(...)`, load_state_dict(assign=True)
    on a sub-module) must change the fingerprint that invalidates packed weights / programs / hipGraphs."""
    from edtr_amd.model.params import ParamTree, params_fingerprint
    t = ParamTree([("a.weight", (4, 4)), ("a.bias", (4,)), ("b.c.weight", (2, 4))])
    fp0 = params_fingerprint(t)
    assert params_fingerprint(t) == fp0
    t.b.c.weight = torch.nn.Parameter(torch.ones(2, 4), requires_grad=False)          # same shape, same version counter (0)
    fp1 = params_fingerprint(t)
    assert fp1 != fp0
    t.a.load_state_dict({"weight": torch.zeros(4, 4), "bias": torch.zeros(4)}, assign=True)
    fp2 = params_fingerprint(t)
    assert fp2 != fp1
    with torch.no_grad():
        t.a.bias.add_(1.0)                                                             # in-place write: the version counter
    assert params_fingerprint(t) != fp2
    other = ParamTree([("x.weight", (2, 2))])                                          # another tree: this one's fingerprint is unchanged
    assert other is not None and params_fingerprint(t) == params_fingerprint(t)


def test_emitted_programs_of_the_precision_modes():
    """The launch programs the three precision modes emit for one ControlNet + UNet evaluation (tiny config, built on the CPU: no
    launch happens) carry the round-4 structure: fast = no operand-formation launches at all; mixed = the fp32 stream's one-part
    consumers (zero / down / upsample convolutions) read fp16 mirrors written by their producers' epilogues (` m16`), the 1x1 skip
    convolutions run the weights-exact two-part product on the mirror (` 2w`), conv1 -> GroupNorm -> conv2 and ff.out -> proj_out
    stay 16-bit, and only the few multi-part stream carriers (conv_in, time path) still form operands by a launch; high = every
    attention runs on hi + lo operand pairs (`attn.split` launches, ` split pv` kernels)."""
    import collections
    from edtr_amd import synth
    from edtr_amd.model.cldm import CldmEngine
    from edtr_amd.testing import build_synthetic_cldm
    cfg = synth.tiny_config()
    progs = {}
    for prec in ("fast", "mixed", "high"):
        cldm = build_synthetic_cldm(cfg, "cpu", torch.bfloat16, precision=prec)
        eng = CldmEngine(cldm, 2, 32, 32, 77)
        progs[prec] = eng.step_prog.recs
    names = {k: collections.Counter(r.name for r in v) for k, v in progs.items()}
    by = lambda prec, name: [r for r in progs[prec] if r.name == name]
    # fast: nothing but the network's own launches
    assert names["fast"]["split_operand"] == 0 and names["fast"]["attn.split"] == 0
    assert not any(" m16" in r.tag or " 2w" in r.tag for r in progs["fast"])
    # mixed
    assert by("mixed", "res.skip1x1") and all(" 2w" in r.tag for r in by("mixed", "res.skip1x1"))
    assert names["mixed"]["split_operand"] <= 8 < names["high"]["split_operand"]           # conv_in x 2 + the time path
    assert sum(" m16" in r.tag for r in progs["mixed"]) >= len(by("mixed", "zero_conv"))   # every block output that a zero conv taps
    assert all(" f32" not in r.tag for r in by("mixed", "res.conv1")), "conv1's output is branch-internal: fp16"
    assert all(" f32" in r.tag for r in by("mixed", "res.conv2")), "conv2 writes the fp32 residual stream"
    assert all(" f32" not in r.tag for r in by("mixed", "ff.out")) and all(" f32" in r.tag for r in by("mixed", "st.proj_out"))
    assert names["mixed"]["attn.split"] == 0
    # high: three launches cut q / k / v^T of a self-attention, one cuts q of a cross-attention (k / v^T of the context once per prompt)
    n_attn = len(by("high", "flash_attn64"))
    assert n_attn and names["high"]["attn.split"] == n_attn // 2 * 3 + n_attn // 2
    assert all(" split pv" in r.tag for r in by("high", "flash_attn64"))
    assert len(progs["fast"]) < len(progs["mixed"]) < len(progs["high"])


def test_max_norm_gate_catches_a_localised_defect(golden_dir):
    """VERDICT r03 missing 2 / weak 2: the relative L2 norm alone lets a localised defect through (a wrong halo column, one bad tile
    seam).  edtr_amd.testing.err_stats adds max|a-b| / max|b| and its 99.99th percentile; bench.golden_parity asserts the max-norm
    against MAX_OVER_L2 x the L2 tolerance.  Here: the golden itself passes, a 2 x 2 latent patch off by 0.5 (L2 error still inside
    the bf16 tolerance) fails on the max-norm alone."""
    import importlib.util
    from edtr_amd.testing import err_stats, rel_err
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(golden_dir.rstrip("/")), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = np.load(os.path.join(golden_dir, "full_det512.npz"))
    sel = [int(k) for k in g["images"]]
    z = torch.zeros((8, 4, 64, 64))
    img = torch.zeros((8, 3, 512, 512))
    z[sel] = torch.from_numpy(g["z"])
    img[sel, :, 1::4, 2::4] = torch.from_numpy(g["img_samples"].astype(np.float32))
    ok = bench.golden_parity("det512", img, z, rel_err, "bf16")["parity_vs_reference_golden"]
    assert ok["ok"] and ok["max_norm_ok"] and ok["rel_err_latent"] == 0.0 and ok["max_err_latent"] == 0.0
    bad = z.clone()
    bad[sel[0], 1, 10:12, 10:12] += 0.5
    res = bench.golden_parity("det512", img, bad, rel_err, "bf16")["parity_vs_reference_golden"]
    assert res["rel_err_latent"] < res["tolerance"]["latent"], "the defect hides inside the L2 tolerance ..."
    assert not res["max_norm_ok"] and not res["ok"], "... and is caught by the max-norm bound"
    st = err_stats(bad[sel], g["z"])
    assert st["max"] > 20 * st["l2"] and st["p9999"] > 10 * st["l2"]
    assert err_stats(torch.ones(10), torch.ones(10)) == {"l2": 0.0, "max": 0.0, "p9999": 0.0}


def test_host_launch_predicates_imply_the_librarys_own(monkeypatch):
    """ops.gn_in_conv_ok / ops.subpixel_ok decide in Python that the halo tile WILL take a convolution (the
    emitter then has no edtr_gn_apply launch to fall back to).  They must therefore sit inside edtr_igemm's own rules, including the
    two they used to omit: 32-bit addressability of the operand and split-K over whole 64-channel chunks.  edtr_igemm_plan answers
    with the tile the library would run (no launch, no GPU)."""
    import ctypes as C
    from edtr_amd import lib as L
    lib = L.load()
    for k in ("EDTR_GN_IN_CONV", "EDTR_IGEMM_HALO", "EDTR_SUBPIXEL", "EDTR_GN_IN_CONV_MAXN", "EDTR_IGEMM_HALO512"):
        monkeypatch.delenv(k, raising=False)
    dummy = C.create_string_buffer(64)
    base = C.addressof(dummy) & ~15

    def plan(B, H, W, cin, N, *, ld=None, splitk=1, a_gn=False, ups=0, ldw=None, K=None, phase=0):
        p = L.IgemmParams()
        OH, OW = (2 * H, 2 * W) if ups else (H, W)
        p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = 0, 9, B * OH * OW, N, K or 9 * cin, 1, 1
        p.a1, p.C1, p.ld1, p.w, p.ldw = base, cin, ld or cin, base, ldw or 9 * cin
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l, p.upsample2x = H, W, OH, OW, 1, 1, 1, ups
        p.alpha, p.out, p.ldc, p.tile, p.splitk = 1.0, base, N, 0, splitk
        if splitk > 1:
            p.workspace, p.workspace_bytes = base, splitk * p.M * N * 4
        if a_gn:
            p.a_gn, p.a_gn_silu = base, 1
        p.w_phase_stride = phase
        return lib.edtr_igemm_plan(C.byref(p))

    seen = {16: 0, 17: 0}
    shapes = [(8, 512, 512, 128, 128), (8, 1024, 1024, 128, 128), (8, 1024, 1024, 256, 128), (1, 64, 64, 64, 128), (4, 32, 32, 128, 128), (8, 16, 16, 1280, 128),
              (8, 256, 256, 256, 256), (2, 16, 16, 192, 128), (3, 64, 64, 64, 128), (8, 64, 48, 128, 128), (8, 2048, 1024, 128, 128)]
    for (B, H, W, cin, N) in shapes:
        for splitk in (1, 2, 3, 6):
            for ld in (cin, 2 * cin):
                if ops.gn_in_conv_ok(B, H, W, cin, N, splitk, ld):
                    t = plan(B, H, W, cin, N, ld=ld, splitk=splitk, a_gn=True)
                    assert t in (16, 17), (B, H, W, cin, N, splitk, ld, t)
                    seen[t] += 1
    assert seen[16] and seen[17]
    # the cases the old predicates got wrong are now refused by the predicate (and by the library, which is why they must be)
    assert not ops.gn_in_conv_ok(8, 1024, 1024, 256, 128, 1, 256) and plan(8, 1024, 1024, 256, 128, a_gn=True) < 0       # 4.29-GB operand
    assert not ops.gn_in_conv_ok(8, 16, 16, 128, 128, 6, 128)                                                           # 6 splits of 2 chunks
    n_sub = 0
    for (B, H, W, Ce, N) in [(8, 256, 256, 256, 256), (8, 128, 128, 512, 512), (8, 32, 32, 640, 640), (2, 16, 16, 64, 128), (8, 512, 512, 768, 128),
                             (8, 1024, 1024, 384, 128)]:
        for ld in (Ce, 3 * Ce):
            if ops.subpixel_ok(H, W, Ce, N, B, ld):
                assert plan(B, H, W, Ce, N, ld=ld, ups=2, ldw=4 * Ce, phase=N * 4 * Ce) == 16, (B, H, W, Ce, N, ld)
                n_sub += 1
    assert n_sub >= 4 and not ops.subpixel_ok(1024, 1024, 384, 128, 8, 384)


def test_demo_padding_helpers_vs_reference_golden(golden_dir):
    """evalutil.pad_if_smaller / pad_to_multiples_of (utils/common.py:337-348): the padded size the REFERENCE produced for the demo
    golden's 150 x 100 input, zeros at the bottom / right only, and a no-op (clone) on aligned sizes."""
    from edtr_amd import evalutil
    g = np.load(os.path.join(golden_dir, "demo_flow.npz"))
    img = synth.synth_input("demo:lq", (1, 3, 150, 100), 0.0, 1.0)
    x = evalutil.pad_to_multiples_of(evalutil.pad_if_smaller(img, size=128), multiple=64)
    assert tuple(x.shape) == tuple(g["padded_shape"]) == (1, 3, 192, 128)
    assert torch.equal(x[:, :, :150, :100], img) and float(x[:, :, 150:].abs().max()) == 0.0 and float(x[:, :, :, 100:].abs().max()) == 0.0
    y = evalutil.pad_to_multiples_of(x, 64)
    assert torch.equal(y, x) and y.data_ptr() != x.data_ptr()
    assert tuple(evalutil.pad_if_smaller(torch.zeros(1, 3, 600, 40), 512).shape) == (1, 3, 600, 512)


def test_weight_store_sibling_formats_share_the_freeze_and_the_tensor_list():
    """The hybrid precision mode packs the same parameters in two storage formats (WeightStore.for_dtype): everything that walks "the"
    store — the packed broadcast, its checksum, the freeze of a receiving rank — must see the sibling formats too."""
    from edtr_amd import ops
    from edtr_amd.engine import WeightStore
    params = {"a.weight": torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64) / 4096.0, "a.bias": torch.ones(64)}
    root = WeightStore(params, torch.float16, torch.device("cpu"))
    sib = root.for_dtype(ops.MIXED)
    assert root.for_dtype(torch.float16) is root and sib.for_dtype(torch.float16) is root and root.for_dtype(ops.MIXED) is sib
    w16, _ = root.linear(["a.weight"], ["a.bias"])
    wmx, _ = sib.linear(["a.weight"], ["a.bias"])
    t1, t3 = w16.get(1), wmx.get(3)
    assert t1.shape[1] == 64 and t3.shape[1] == 3 * 64
    ptrs = {t.data_ptr() for t in root.tensors()}
    assert t1.data_ptr() in ptrs and t3.data_ptr() in ptrs, "a root lists its sibling formats' tensors"
    root.frozen = "received from rank 0"
    assert sib.frozen == "received from rank 0"
    with pytest.raises(RuntimeError, match="frozen"):
        sib.vec("a.bias", 128)
    sib.frozen = None
    assert root.frozen is None
    root.invalidate()
    assert not sib.cache and not root.cache


def test_precision_sections_and_policies():
    """precision="hybrid" / "robust" resolve to per-section modes and policies (edtr_amd/model/cldm.py, edtr_amd/precision.py)."""
    import json
    from edtr_amd import ops
    from edtr_amd.model import cldm as M
    from edtr_amd.precision import mixed_policy, robust_policy
    assert M.section_mode("hybrid", torch.bfloat16, "cldm") == ("fast", torch.float16)
    assert M.section_mode("hybrid", torch.bfloat16, "vae.encode") == ("fast", torch.float16)
    assert M.section_mode("hybrid", torch.bfloat16, "vae.decode")[0] == "mixed"
    assert M.section_mode("robust", torch.bfloat16, "cldm")[0] == "mixed" and M.section_mode("mixed", torch.bfloat16, "vae.decode")[0] == "mixed"
    assert M._store_dtype("hybrid", torch.bfloat16) == torch.float16 and M._store_dtype("hybrid", torch.bfloat16, "vae.decode") == ops.MIXED
    os.environ["EDTR_AMD_HYBRID"] = json.dumps({"vae.encode": "mixed"})
    try:
        assert M.section_mode("hybrid", torch.bfloat16, "vae.encode")[0] == "mixed"
        os.environ["EDTR_AMD_HYBRID"] = json.dumps({"vae.encode": "nonsense"})
        with pytest.raises(ValueError):
            M.hybrid_sections()
    finally:
        del os.environ["EDTR_AMD_HYBRID"]
    rp, mp = robust_policy(), mixed_policy()
    assert rp.attn_split == 1 and mp.attn_split is None and rp.key() != mp.key()
    assert rp.parts("res.conv1") == 3 and rp.parts("ff.geglu") == 3 and rp.parts("vae.conv1") == 1 and rp.parts("vae.upsample.conv") == 3
    assert mp.parts("res.conv1") == 1


def test_ffn_host_packing_matches_the_header():
    """ops.pack_ffn_w2 / pack_ffn_constants write what include/edtr_hip.h documents for edtr_ffn (column permutation inside groups of 16;
    per-chunk constant order with halved gate entries); ffn_ok is the host's launch rule."""
    from edtr_amd import ops
    w2 = torch.arange(320 * 1280, dtype=torch.float32).reshape(320, 1280)
    p2 = ops.pack_ffn_w2(w2, torch.float32)
    perm = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]
    for g in (0, 5, 79):
        for i in range(16):
            assert float(p2[7, 16 * g + i]) == float(w2[7, 16 * g + perm[i]])
    c2b = torch.arange(2560, dtype=torch.float32)
    cst = ops.pack_ffn_constants(c2b).reshape(20, 2, 4, 2, 2, 4)
    for (c, hh, q, vg, lh, e) in [(0, 0, 0, 0, 0, 0), (3, 1, 2, 1, 1, 3), (19, 1, 3, 0, 1, 2)]:
        R = 128 * c + 64 * hh + 32 * vg + e + 8 * q + 4 * lh
        assert float(cst[c, hh, q, vg, lh, e]) == (0.5 if vg else 1.0) * R
    assert ops.ffn_ok(32768, 320, 1280) and ops.ffn_ok(16384, 320, 1280)
    assert not ops.ffn_ok(4096, 320, 1280) and not ops.ffn_ok(32768, 640, 2560) and not ops.ffn_ok(16400, 320, 1280)


def test_tile21_is_opt_in_and_refuses_what_it_cannot_hold():
    """Tile 21 (the persistent form of tile 17 with deferred stores, round 6) runs only when asked for: 16-bit output, no residual, no
    time-embedding row, an even number of 32-channel chunks (edtr_igemm_plan: every check of the launch, no HIP call)."""
    import ctypes as C
    from edtr_amd import lib as L
    lib = L.load()
    buf = (C.c_char * 4096)()
    base = C.addressof(buf) & ~15

    def plan(tile, cin=64, N=128, residual=False, out_f32=False, rowvec=False, H=512, W=512):
        p = L.IgemmParams()
        p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = 0, 9, H * W, N, 9 * cin, 1, 1
        p.a1, p.C1, p.ld1, p.w, p.ldw = base, cin, cin, base, 9 * cin
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l = H, W, H, W, 1, 1, 1
        p.alpha, p.out, p.ldc, p.tile, p.splitk, p.out_f32 = 1.0, base, N, tile, 1, int(out_f32)
        if residual:
            p.residual, p.ldr = base, N
        if rowvec:
            p.rowvec, p.rowvec_ld, p.rows_per_image = base, N, H * W
        return lib.edtr_igemm_plan(C.byref(p))

    assert plan(21) == 21 and plan(21, cin=128, N=256) == 21
    assert plan(0) == 17                                            # the automatic choice stays tile 17 (EDTR_IGEMM_HALO512P is the A/B switch)
    assert plan(21, residual=True) < 0 and plan(21, out_f32=True) < 0 and plan(21, rowvec=True) < 0
    assert plan(21, cin=96) < 0                                     # three chunks
    assert plan(21, W=528) < 0 and plan(17, W=528) < 0              # tile 17's own shape rules apply



def test_lin320_host_packing_and_launch_rules():
    """ops.pack_lin320_w writes the fragment order include/edtr_hip.h documents for edtr_lin320; WeightStore.lin320 folds the LayerNorm (gamma into
    the columns, alpha W beta + bias into the additive row, vt_alpha on the V rows); edtr_lin320_plan answers every launch check without HIP."""
    import ctypes as C
    from edtr_amd import lib as L
    from edtr_amd import ops
    from edtr_amd.engine import WeightStore
    w = torch.arange(96 * 320, dtype=torch.float32).reshape(96, 320)
    pk = ops.pack_lin320_w(w, torch.float32).reshape(3, 20, 64, 8)
    for (c, s, lane, j) in [(0, 0, 0, 0), (1, 7, 37, 5), (2, 19, 63, 7), (0, 3, 31, 2)]:
        assert float(pk[c, s, lane, j]) == float(w[32 * c + (lane & 31), 16 * s + 8 * (lane >> 5) + j])
    g = torch.Generator().manual_seed(5)
    params = {"n.weight": 1 + 0.1 * torch.randn(320, generator=g), "n.bias": 0.1 * torch.randn(320, generator=g),
              "q.weight": torch.randn(320, 320, generator=g) / 18, "k.weight": torch.randn(320, 320, generator=g) / 18,
              "v.weight": torch.randn(320, 320, generator=g) / 18, "o.weight": torch.randn(320, 320, generator=g) / 18, "o.bias": torch.randn(320, generator=g)}
    store = WeightStore(params, torch.float32, torch.device("cpu"))      # (fp32 "storage": the packing is exact, the fold can be checked to rounding)
    wq, cq = store.lin320(["q.weight", "k.weight", "v.weight"], None, "n.", 0.5, vt_col0=640, vt_alpha=1.0)
    wcat = torch.cat([params["q.weight"], params["k.weight"], params["v.weight"]])
    assert torch.equal(wq, ops.pack_lin320_w(wcat * params["n.weight"][None, :], torch.float32))
    shift = wcat @ params["n.bias"]
    assert torch.allclose(cq[:640], 0.5 * shift[:640], atol=1e-6) and torch.allclose(cq[640:], shift[640:], atol=1e-6)
    wo, co = store.lin320(["o.weight"], ["o.bias"], None, 1.0)
    assert torch.equal(wo, ops.pack_lin320_w(params["o.weight"], torch.float32)) and torch.equal(co, params["o.bias"])
    assert store.lin320(["o.weight"], None, None, 1.0)[1] is None

    lib = L.load()
    buf = (C.c_char * 4096)()
    base = C.addressof(buf) & ~15

    def plan(M=256, N=320, K=320, ln=0, res=False, vt=False, gn=False, rpi=0, same=False, ldo=None):
        p = L.Lin320Params()
        p.dtype, p.M, p.N, p.K, p.ln, p.eps, p.alpha = 0, M, N, K, ln, 1e-5, 1.0
        p.x, p.ldx, p.w, p.out, p.ldo = base, 320, base + 1024, base if same else base + 2048, ldo or (640 if vt else N)
        if res:
            p.residual, p.ldr = base + 512, N
        if vt:
            p.vt_out, p.vt_col0, p.vt_ld, p.vt_alpha, p.rows_per_image = base + 3072, 640, rpi or 128, 1.0, rpi or 128
        if gn:
            p.gn_table, p.rows_per_image = base + 3584, rpi or 128
        return lib.edtr_lin320_plan(C.byref(p))

    assert plan() == 0 and plan(ln=1, res=True) == 0 and plan(N=960, ln=1, vt=True) == 0 and plan(gn=True) == 0 and plan(gn=True, res=True) == 0
    assert plan(M=192) < 0 and plan(N=96) < 0 and plan(K=640) < 0 and plan(same=True) < 0 and plan(N=2048) < 0
    assert plan(N=960, vt=True, res=True) < 0 and plan(N=960, vt=True, rpi=48) < 0 and plan(N=960, vt=True, ldo=320) < 0
    assert plan(gn=True, ln=1) < 0 and plan(gn=True, rpi=192) < 0 and plan(M=384, gn=True, rpi=256) < 0
