"""Deterministic synthetic weights / inputs and the named model configurations.

No real EDTR / SD-2.1 checkpoints exist offline, so every parity
fixture and every bench run uses parameters produced by a closed-form integer hash
keyed by ``(state-dict key, flat index)``.  The hash is pure int64 arithmetic, so
the reference side (tools/make_goldens.py), the oracle and the HIP engine all
regenerate bit-identical fp32 values without relying on any torch RNG stream.

The reference initialises 69 tensors to zero (``zero_module``: reference
model/unet.py:177,260,678, model/controlnet.py:261, model/attention.py:274,280),
which makes every control tensor and eps exactly 0; the generator therefore
overwrites *all* parameters, including those.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_M32 = (1 << 32) - 1


def _hash_u32(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """murmur3-style finaliser over (index, seed); int64 in, values in [0, 2^32)."""
    x = (idx * 2654435761 + seed) & _M32
    x = x ^ (x >> 16)
    x = (x * 2246822519) & _M32
    x = x ^ (x >> 13)
    x = (x * 3266489917) & _M32
    x = x ^ (x >> 16)
    return x


def hashed_uniform(key: str, numel: int, chunk: int = 1 << 24, device=None) -> torch.Tensor:
    """fp32 tensor of ``numel`` values, uniform in [-1, 1), fully determined by ``key``.  ``device``: where the integer hash
    runs — int64 arithmetic and exactly representable fp32 results, so a GPU produces the same bits as the CPU (the 1.3 G
    parameters of the SD-2.1 size take about a minute on the host cores and about a second on the device)."""
    seed = zlib.crc32(key.encode("utf-8")) & _M32
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for s in range(0, numel, chunk):
        e = min(numel, s + chunk)
        idx = torch.arange(s, e, dtype=torch.int64, device=device)
        h = _hash_u32(idx, seed)
        # 24 random bits -> exactly representable fp32 in [-1, 1)
        out[s:e] = ((h >> 8).to(torch.float32) * (2.0 / (1 << 24))) - 1.0
    return out


def synth_param(key: str, shape: Tuple[int, ...], device=None) -> torch.Tensor:
    """Synthetic value for one state-dict entry.

    * norm scales (GroupNorm / LayerNorm ``weight``, 1-D, not a bias):  1 + 0.1 u
    * biases (1-D ``bias``):                                            0.05 u
    * conv / linear weights (>= 2-D):  u * sqrt(3 / fan_in)  (unit-variance preserving)
    """
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    u = hashed_uniform(key, numel, device=device).reshape(shape)
    if len(shape) <= 1:
        if key.endswith("bias"):
            return 0.05 * u
        return 1.0 + 0.1 * u
    fan_in = int(np.prod(shape[1:]))
    return u * float(np.sqrt(3.0 / fan_in))


HEAVY_ROW, HEAVY_GAIN, HEAVY_QK = 12.0, 4.0, 1.5


def synth_param_heavy(key: str, shape: Tuple[int, ...], device=None) -> torch.Tensor:
    """The "heavy-tailed" weight set: the statistics real SD-2.1 / ControlNet checkpoints have and the
    smooth set above lacks — outlier channels, large norm gains, sharp attention.  Same hash, then:

    * conv / linear weights: one output channel in ~128 is an OUTLIER, its whole row x HEAVY_ROW;
    * GroupNorm / LayerNorm gains: one channel in ~64 has gain +-HEAVY_GAIN instead of 1 +- 0.1;
    * attention `to_q` / `to_k` (and the VAE's `q` / `k`): x HEAVY_QK each (logit std ~2.3, x12 more on outlier rows).  Sharper
      than that and the NETWORK turns chaotic — at x4 a 1e-5 perturbation of the input image changes eps by 40 % after four
      steps in fp32 on the CPU (argmax-like attention), so no arithmetic could be compared with any other; x1.5 keeps the
      amplification of a perturbation at <= 20 (the smooth set: ~1).  Logits of +-1e4 are driven through the attention
      kernels directly (tests/test_gpu_heavy.py::test_flash_attention_with_extreme_logits);
    * biases x 4.
    Used for range-robustness tests (fp16 storage / fp16 operands must stay finite and inside their envelopes).  The factors
    are calibrated so that the residual stream peaks at ~1e4 (tests/golden/heavy.npz: sd21_mid_absmax), the edge of what fp16
    can hold at all — the reference's own GPU path is fp16 autocast (main/det/test_edtr.py:95-96), so a checkpoint whose
    activations exceed 65504 does not run there either; the first calibration (x50 rows, +-10 gains) reached 6.5e6."""
    base = synth_param(key, shape, device=device)
    shape = tuple(int(s) for s in shape)
    n0 = shape[0] if len(shape) else 1
    seed = zlib.crc32(("heavy:" + key).encode("utf-8")) & _M32
    ch = _hash_u32(torch.arange(n0, dtype=torch.int64, device=device), seed)
    if len(shape) <= 1:
        if key.endswith("bias"):
            return 4.0 * base
        big = (ch % 64) == 0
        sign = torch.where((ch >> 7) % 2 == 0, 1.0, -1.0)
        return torch.where(big, HEAVY_GAIN * sign, base)
    scale = torch.where((ch % 128) == 0, HEAVY_ROW, 1.0).to(torch.float32)
    if any(key.endswith(sfx) for sfx in ("to_q.weight", "to_k.weight", ".q.weight", ".k.weight")):
        scale = scale * HEAVY_QK
    return base * scale.reshape((n0,) + (1,) * (len(shape) - 1))


MODERATE_ROW, MODERATE_GAIN, MODERATE_QK = 8.0, 3.0, 1.3


def synth_param_moderate(key: str, shape: Tuple[int, ...], device=None) -> torch.Tensor:
    """The "moderate-outlier" weight set: the heavy set's STRUCTURE — one outlier output channel in
    ~128, one large norm gain in ~64, sharper attention, larger biases — at factors (x8 rows, +-3 gains, x1.3 q / k, x3 biases)
    calibrated so that the residual stream stays below ~6e3, i.e. inside what the reference's own fp16-autocast GPU path runs with
    headroom, and the network's amplification of a rounding stays within a small factor of the smooth set's (the heavy set's 10 -
    30 x makes its 4-step pipeline a poor gate: two builds of the same arithmetic moved its fp16 error 2 x).  What the tests pin
    on this set (tests/test_gpu_heavy.py against the reference's outputs in tests/golden/moderate.npz): precision="mixed" does NOT
    meet the north-star 1e-3 here (it measures 1.6 - 1.9e-3 and is bounded at <= 1.5 x its measured value); the modes asserted
    against 1e-3 itself are the ones named there (`high`, and any faster mode that test lists)."""
    base = synth_param(key, shape, device=device)
    shape = tuple(int(s) for s in shape)
    n0 = shape[0] if len(shape) else 1
    seed = zlib.crc32(("moderate:" + key).encode("utf-8")) & _M32
    ch = _hash_u32(torch.arange(n0, dtype=torch.int64, device=device), seed)
    if len(shape) <= 1:
        if key.endswith("bias"):
            return 3.0 * base
        big = (ch % 64) == 0
        sign = torch.where((ch >> 7) % 2 == 0, 1.0, -1.0)
        return torch.where(big, MODERATE_GAIN * sign, base)
    scale = torch.where((ch % 128) == 0, MODERATE_ROW, 1.0).to(torch.float32)
    if any(key.endswith(sfx) for sfx in ("to_q.weight", "to_k.weight", ".q.weight", ".k.weight")):
        scale = scale * MODERATE_QK
    return base * scale.reshape((n0,) + (1,) * (len(shape) - 1))


WEIGHT_SETS = {"smooth": synth_param, "heavy": synth_param_heavy, "moderate": synth_param_moderate}


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], prefix: str = "") -> Dict[str, torch.Tensor]:
    """``spec`` yields (key, shape); the hash key is ``prefix + key`` so that the unet and
    the controlnet (which share key names) get different values."""
    return {k: synth_param(prefix + k, shp) for k, shp in spec}


def synth_input(name: str, shape: Tuple[int, ...], lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    u = hashed_uniform("input:" + name, int(np.prod(shape))).reshape(shape)
    return (u + 1.0) * (0.5 * (hi - lo)) + lo


def synth_normal(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """Approximately N(0,1) (sum of 4 uniforms, variance-normalised): the explicit noise
    tensors injected into q_sample / p_sample so CPU and GPU runs see the same stream."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    acc = torch.zeros(n, dtype=torch.float32)
    for j in range(4):
        acc += hashed_uniform(f"noise:{name}:{j}", n)
    return (acc * float(np.sqrt(3.0 / 4.0))).reshape(shape)


# ----------------------------------------------------------------------------------------------
# Named configurations (constructor kwargs of ControlLDM; reference configs/det/demo.yaml:20-86)
# ----------------------------------------------------------------------------------------------

def sd21_config() -> dict:
    """The SD-2.1 / EDTR configuration every shipped YAML uses (configs/det/demo.yaml:24-86)."""
    unet = dict(
        use_checkpoint=True, image_size=32, in_channels=4, out_channels=4, model_channels=320,
        attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
        num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
        transformer_depth=1, context_dim=1024, legacy=False,
    )
    cnet = dict(unet)
    cnet.pop("out_channels")
    cnet["hint_channels"] = 4
    vae = dict(
        train_decoder=True, embed_dim=4,
        ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                      ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0),
    )
    clip = dict(
        embed_dim=1024,
        vision_cfg=dict(image_size=224, layers=32, width=1280, head_width=80, patch_size=14),
        text_cfg=dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24),
        layer="penultimate",
    )
    return dict(unet_cfg=unet, vae_cfg=vae, clip_cfg=clip, controlnet_cfg=cnet, latent_scale_factor=0.18215)


def tiny_config() -> dict:
    """Reduced configuration for end-to-end goldens: same topology (4 levels, 2 res blocks,
    transformers at the first three levels, head width 64), 1/5 of the channels."""
    cfg = sd21_config()
    for k in ("unet_cfg", "controlnet_cfg"):
        cfg[k] = dict(cfg[k], model_channels=64, context_dim=64)
    cfg["vae_cfg"] = dict(cfg["vae_cfg"], ddconfig=dict(cfg["vae_cfg"]["ddconfig"], ch=32, resolution=64))
    cfg["clip_cfg"] = dict(
        embed_dim=64,
        vision_cfg=dict(image_size=32, layers=1, width=64, head_width=32, patch_size=16),
        text_cfg=dict(context_length=77, vocab_size=49408, width=64, heads=2, layers=2),
        layer="penultimate",
    )
    return cfg


def swinir_config() -> dict:
    """SwinIR pre-restoration network of reference configs/det/demo.yaml:2-18."""
    return dict(img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8,
                mlp_ratio=2, sf=8, img_range=1.0, upsampler="nearest+conv", resi_connection="1conv", unshuffle=True,
                unshuffle_scale=8)


def swinir_small_config() -> dict:
    """Same structure, 2 groups of 2 layers, 2 heads of the SAME head width (30) as the shipped network."""
    cfg = swinir_config()
    cfg.update(embed_dim=60, depths=[2, 2], num_heads=[2, 2])
    return cfg


def clip_small_config() -> dict:
    """Reduced text tower for the CLIP goldens: head width 64 like ViT-H (2 heads x 64), 3 layers."""
    return dict(embed_dim=128, vision_cfg=dict(image_size=32, layers=1, width=64, head_width=32, patch_size=16),
                text_cfg=dict(context_length=77, vocab_size=49408, width=128, heads=2, layers=3), layer="penultimate")


def clip_test_tokens() -> "torch.Tensor":
    """[4, 77] int64: the empty prompt, a short prompt, a full-length row of hashed ids, a row ending early."""
    t = torch.zeros((4, 77), dtype=torch.int64)
    t[0, :2] = torch.tensor([49406, 49407])
    t[1, :7] = torch.tensor([49406, 320, 1125, 539, 320, 2368, 49407])
    ids = (hashed_uniform("clip:tokens", 75) * 0.5 + 0.5).mul(49000).long().clamp(1, 49000)
    t[2, 0], t[2, 1:76], t[2, 76] = 49406, ids, 49407
    t[3, 0], t[3, 1:40], t[3, 40] = 49406, ids[:39].flip(0), 49407
    return t


CONFIGS = {"sd21": sd21_config, "tiny": tiny_config}
