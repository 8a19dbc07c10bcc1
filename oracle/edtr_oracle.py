"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU fp32 restatement (plain torch functional ops + numpy float64 for the schedules) of the
reference's ControlLDM / SD-2.1 restoration hot path.  Only tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg may import this file, and only as the *checker*; the
product path (edtr_amd/) never routes through it and fails loudly without its HIP library.

Parity pin: tests/test_oracle_golden.py checks every function below against fixtures in
tests/golden/ that tools/make_goldens.py produced by running the real reference
(/root/reference, CPU fp32) on the same synthetic weights/inputs (edtr_amd/synth.py).
Parity with the *released* checkpoints is unpinned (no weights exist offline).

The restatement is functional: every network is a pure function of a flat ``{key: tensor}``
state dict that uses the reference's parameter names, so the same dict can be loaded into the
reference modules.  Each function cites the reference lines (relative to /root/reference) it follows.
The only third-party arithmetic is ATen (conv2d, linear, group_norm, layer_norm, softmax, gelu(erf),
nearest interpolation) — the same library the reference calls.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------
# a1  schedules
# ------------------------------------------------------------------------------------------

def make_betas(n_timestep: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.0120) -> np.ndarray:
    """SD "linear" schedule = linspace in sqrt space, squared.  model/gaussian_diffusion.py:9-13."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


def space_timesteps(num_timesteps: int, section_counts) -> List[int]:
    """IDDPM respacing.  utils/sampler.py:14-64.  Returns the sorted used timesteps."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                steps = list(range(0, num_timesteps, stride))
                if len(steps) == want:
                    return steps
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(s) for s in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    start, used = 0, []
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            used.append(start + round(pos))
            pos += stride
        start += size
    return sorted(set(used))


def schedule_tables(betas: np.ndarray, used_timesteps: Sequence[int]) -> Dict[str, np.ndarray]:
    """Re-spaced posterior tables, float64 math then fp32.  utils/sampler.py:85-133."""
    ac_full = np.cumprod(1.0 - betas, axis=0)
    used = sorted(set(int(t) for t in used_timesteps))
    new_betas, last = [], 1.0
    for t in used:
        new_betas.append(1.0 - ac_full[t] / last)
        last = ac_full[t]
    b = np.array(new_betas, dtype=np.float64)
    a = 1.0 - b
    ac = np.cumprod(a, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = b * (1.0 - ac_prev) / (1.0 - ac)
    if len(used) == 1:
        post_logvar = np.array([-10.0])
    else:
        post_logvar = np.log(np.append(post_var[1], post_var[1:]))
    tabs = {
        "timesteps": np.array(used, dtype=np.int32),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1.0),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": post_logvar,
        "posterior_mean_coef1": b * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(a) / (1.0 - ac),
    }
    return {k: (v if k == "timesteps" else v.astype(np.float32)) for k, v in tabs.items()}


def q_sample_coefs(betas: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """model/gaussian_diffusion.py:63-75 (fp32 buffers of sqrt(ac), sqrt(1-ac))."""
    ac = np.cumprod(1.0 - betas, axis=0)
    return np.sqrt(ac).astype(np.float32), np.sqrt(1.0 - ac).astype(np.float32)


def _bcast(table: np.ndarray, idx: torch.Tensor, ndim: int) -> torch.Tensor:
    """extract_into_tensor, model/gaussian_diffusion.py:34-37."""
    v = torch.from_numpy(np.asarray(table, dtype=np.float32))[idx.long()]
    return v.reshape(-1, *([1] * (ndim - 1)))


def q_sample(betas: np.ndarray, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """model/gaussian_diffusion.py:80-84."""
    sa, sb = q_sample_coefs(betas)
    return _bcast(sa, t, x0.dim()) * x0 + _bcast(sb, t, x0.dim()) * noise


def p_sample_update(tabs, x, eps, noise, index) -> Tuple[torch.Tensor, torch.Tensor]:
    """x0 from eps, posterior mean, masked noise.  utils/sampler.py:160-164, 135-158, 196-203."""
    n = x.dim()
    pred_x0 = _bcast(tabs["sqrt_recip_alphas_cumprod"], index, n) * x \
        - _bcast(tabs["sqrt_recipm1_alphas_cumprod"], index, n) * eps
    mean = _bcast(tabs["posterior_mean_coef1"], index, n) * pred_x0 \
        + _bcast(tabs["posterior_mean_coef2"], index, n) * x
    var = _bcast(tabs["posterior_variance"], index, n)
    mask = (index != 0).float().reshape(-1, *([1] * (n - 1)))
    return mean + mask * torch.sqrt(var) * noise, pred_x0


# ------------------------------------------------------------------------------------------
# a8  timestep embedding
# ------------------------------------------------------------------------------------------

def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[cos | sin] sinusoid, fp32.  model/util.py:98-118."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def time_embed(sd: SD, p: str, t: torch.Tensor, model_channels: int) -> torch.Tensor:
    """Linear-SiLU-Linear on the sinusoid.  model/unet.py:475-480, model/controlnet.py:128-133."""
    e = timestep_embedding(t, model_channels)
    e = F.linear(e, sd[p + "time_embed.0.weight"], sd[p + "time_embed.0.bias"])
    return F.linear(F.silu(e), sd[p + "time_embed.2.weight"], sd[p + "time_embed.2.bias"])


# ------------------------------------------------------------------------------------------
# a9-a15  UNet / ControlNet building blocks
# ------------------------------------------------------------------------------------------

def gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    """32-group GroupNorm in fp32.  model/util.py:146-163 (eps 1e-5), model/attention.py:50-51 and
    model/vae.py:22-23 (eps 1e-6)."""
    return F.group_norm(x.float(), 32, sd[p + "weight"], sd[p + "bias"], eps)


def conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=padding)


def resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """GN-SiLU-conv, + Linear(SiLU(emb)), GN-SiLU-conv, + skip.  model/unet.py:203-223 (non-updown,
    no scale-shift norm, dropout p=0)."""
    h = conv(sd, p + "in_layers.2.", F.silu(gn(sd, p + "in_layers.0.", x, 1e-5)))
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = conv(sd, p + "out_layers.3.", F.silu(gn(sd, p + "out_layers.0.", h, 1e-5)))
    if (p + "skip_connection.weight") in sd:
        x = conv(sd, p + "skip_connection.", x, padding=0)
    return x + h


def attention(sd: SD, p: str, x: torch.Tensor, ctx: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """q/k/v projections (no bias), per-head softmax(q k^T / sqrt(d)) v, output Linear (+bias).
    model/attention.py:176-203 (SDP variant: default scale, no mask, no dropout)."""
    ctx = x if ctx is None else ctx
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(ctx, sd[p + "to_k.weight"])
    v = F.linear(ctx, sd[p + "to_v.weight"])
    b, n, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    w = torch.softmax((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(d)), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def transformer_block(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """LN-selfattn, LN-crossattn, LN-GEGLU-FF, each residual.  model/attention.py:230-234, 20-47."""
    c = x.shape[-1]

    def ln(name, t):
        return F.layer_norm(t, (c,), sd[p + name + ".weight"], sd[p + name + ".bias"], 1e-5)

    x = attention(sd, p + "attn1.", ln("norm1", x), None, heads) + x
    x = attention(sd, p + "attn2.", ln("norm2", x), ctx, heads) + x
    return feed_forward(sd, p, x)


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """x + ff(norm3(x)): LayerNorm (eps 1e-5), the GEGLU projection (value * gelu(gate), exact GELU), the output projection.
    model/attention.py:233 with FeedForward :30-47 and GEGLU :20-27 — what edtr_ffn computes in one launch."""
    c = x.shape[-1]
    h = F.linear(F.layer_norm(x, (c,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5), sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    val, gate = h.chunk(2, dim=-1)
    h = val * F.gelu(gate)
    return F.linear(h, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"]) + x


def norm_linear(sd: SD, ln: Optional[str], w: str, b: Optional[str], x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Linear(LayerNorm(x)) (+ residual): the pairs `to_q(norm2(x))` (model/attention.py:171 behind :231), `to_out[0](o) + x` (:195 behind
    :230-231), `proj_in` / `proj_out` (:283-302) taken on their own — what one edtr_lin320 launch computes."""
    if ln is not None:
        x = F.layer_norm(x, (x.shape[-1],), sd[ln + "weight"], sd[ln + "bias"], 1e-5)
    y = F.linear(x, sd[w], sd[b] if b else None)
    return y if residual is None else y + residual


def spatial_transformer(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """GN(1e-6) -> tokens -> proj_in -> block -> proj_out -> image, + input.
    model/attention.py:283-302 (use_linear=True, depth 1)."""
    b, c, hh, ww = x.shape
    t = gn(sd, p + "norm.", x, 1e-6).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    t = F.linear(t, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    t = transformer_block(sd, p + "transformer_blocks.0.", t, ctx, heads)
    t = F.linear(t, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return t.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + x


def encoder_layout(cfg: dict) -> List[List[Tuple[str, int]]]:
    """Kinds of the layers inside each of the 12 input blocks, with the channel count after the
    block.  Mirrors the constructor loops of model/unet.py:505-573 / model/controlnet.py:146-214."""
    mc, mult = cfg["model_channels"], cfg["channel_mult"]
    nres = cfg["num_res_blocks"]
    blocks = [[("conv", mc)]]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            layers = [("res", m * mc)]
            if ds in cfg["attention_resolutions"]:
                layers.append(("attn", m * mc))
            blocks.append(layers)
        if level != len(mult) - 1:
            blocks.append([("down", m * mc)])
            ds *= 2
    return blocks


def _run_block(sd, p, layers, h, emb, ctx, head_dim):
    for j, (kind, ch) in enumerate(layers):
        q = f"{p}{j}."
        if kind == "conv":
            h = conv(sd, q, h)
        elif kind == "res":
            h = resblock(sd, q, h, emb)
        elif kind == "attn":
            h = spatial_transformer(sd, q, h, ctx, ch // head_dim)
        elif kind == "down":
            h = conv(sd, q + "op.", h, stride=2)  # model/unet.py:99-108
        elif kind == "up":
            h = conv(sd, q + "conv.", F.interpolate(h, scale_factor=2, mode="nearest"))  # unet.py:70-79
    return h


def controlnet_forward(sd: SD, cfg: dict, x, hint, t, ctx, p: str = "") -> List[torch.Tensor]:
    """ControlNet: 12 encoder blocks each tapped by a 1x1 conv, middle block + tap -> 13 tensors.
    model/controlnet.py:263-277."""
    emb = time_embed(sd, p, t, cfg["model_channels"])
    h = torch.cat((x, hint), dim=1).float()
    hd = cfg["num_head_channels"]
    outs = []
    layout = encoder_layout(cfg)
    for i, layers in enumerate(layout):
        h = _run_block(sd, f"{p}input_blocks.{i}.", layers, h, emb, ctx, hd)
        outs.append(conv(sd, f"{p}zero_convs.{i}.0.", h, padding=0))
    ch = layout[-1][-1][1]
    h = _run_block(sd, f"{p}middle_block.", [("res", ch), ("attn", ch), ("res", ch)], h, emb, ctx, hd)
    outs.append(conv(sd, f"{p}middle_block_out.0.", h, padding=0))
    return outs


def decoder_layout(cfg: dict) -> List[List[Tuple[str, int]]]:
    """Layer kinds of the 12 output blocks.  model/unet.py:621-672."""
    mc, mult, nres = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    ds = 2 ** (len(mult) - 1)
    blocks = []
    for level in reversed(range(len(mult))):
        for i in range(nres + 1):
            layers = [("res", mc * mult[level])]
            if ds in cfg["attention_resolutions"]:
                layers.append(("attn", mc * mult[level]))
            if level and i == nres:
                layers.append(("up", mc * mult[level]))
                ds //= 2
            blocks.append(layers)
    return blocks


def unet_forward(sd: SD, cfg: dict, x, t, ctx, control: Optional[List[torch.Tensor]], p: str = "") -> torch.Tensor:
    """ControlledUnetModel.forward: control added to the middle output and to every skip tensor
    before the concat.  model/controlnet.py:20-41; output head model/unet.py:675-679."""
    emb = time_embed(sd, p, t, cfg["model_channels"])
    hd = cfg["num_head_channels"]
    control = list(control) if control is not None else None
    hs = []
    h = x.float()
    layout = encoder_layout(cfg)
    for i, layers in enumerate(layout):
        h = _run_block(sd, f"{p}input_blocks.{i}.", layers, h, emb, ctx, hd)
        hs.append(h)
    ch = layout[-1][-1][1]
    h = _run_block(sd, f"{p}middle_block.", [("res", ch), ("attn", ch), ("res", ch)], h, emb, ctx, hd)
    if control is not None:
        h = h + control.pop()
    for i, layers in enumerate(decoder_layout(cfg)):
        skip = hs.pop()
        if control is not None:
            skip = skip + control.pop()
        h = _run_block(sd, f"{p}output_blocks.{i}.", layers, torch.cat([h, skip], dim=1), emb, ctx, hd)
    return conv(sd, p + "out.2.", F.silu(gn(sd, p + "out.0.", h, 1e-5)))


def cldm_forward(sd: SD, cfg: dict, x, t, cond: Dict[str, torch.Tensor], control_scales=None) -> torch.Tensor:
    """ControlLDM.forward.  model/cldm.py:166-194 (woSD=False path).  ``sd`` uses the cldm key names
    (``unet.*``, ``controlnet.*``); ``cfg`` is the ControlLDM kwargs dict."""
    ctrl = controlnet_forward(sd, cfg["controlnet_cfg"], x, cond["c_img"], t, cond["c_txt"], p="controlnet.")
    scales = control_scales or [1.0] * 13
    ctrl = [c * s for c, s in zip(ctrl, scales)]
    return unet_forward(sd, cfg["unet_cfg"], x, t, cond["c_txt"], ctrl, p="unet.")


# ------------------------------------------------------------------------------------------
# a17-a19  VAE
# ------------------------------------------------------------------------------------------

def _swish(x):
    return x * torch.sigmoid(x)  # model/vae.py:17-19


def vae_resblock(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """model/vae.py:103-124 (temb None, dropout 0, nin_shortcut when channels change)."""
    h = conv(sd, p + "conv1.", _swish(gn(sd, p + "norm1.", x, 1e-6)))
    h = conv(sd, p + "conv2.", _swish(gn(sd, p + "norm2.", h, 1e-6)))
    if (p + "nin_shortcut.weight") in sd:
        x = conv(sd, p + "nin_shortcut.", x, padding=0)
    return x + h


def vae_attn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Single-head attention over HW tokens with 1x1 conv projections.  model/vae.py:279-308."""
    b, c, hh, ww = x.shape
    n = gn(sd, p + "norm.", x, 1e-6)
    q, k, v = (conv(sd, p + name + ".", n, padding=0).reshape(b, c, hh * ww).transpose(1, 2) for name in "qkv")
    w = torch.softmax((q @ k.transpose(1, 2)) * (1.0 / math.sqrt(c)), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(b, c, hh, ww)
    return x + conv(sd, p + "proj_out.", o, padding=0)


def vae_encoder(sd: SD, dd: dict, x: torch.Tensor, p: str = "encoder.") -> torch.Tensor:
    """model/vae.py:421-446.  Downsample = pad (0,1,0,1) then conv3x3 stride 2 pad 0 (:54-61)."""
    nlev = len(dd["ch_mult"])
    h = conv(sd, p + "conv_in.", x)
    for lvl in range(nlev):
        for blk in range(dd["num_res_blocks"]):
            h = vae_resblock(sd, f"{p}down.{lvl}.block.{blk}.", h)
        if lvl != nlev - 1:
            h = conv(sd, f"{p}down.{lvl}.downsample.conv.", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = vae_resblock(sd, p + "mid.block_1.", h)
    h = vae_attn(sd, p + "mid.attn_1.", h)
    h = vae_resblock(sd, p + "mid.block_2.", h)
    return conv(sd, p + "conv_out.", _swish(gn(sd, p + "norm_out.", h, 1e-6)))


def vae_decoder(sd: SD, dd: dict, z: torch.Tensor, p: str = "decoder.") -> torch.Tensor:
    """model/vae.py:527-560 (give_pre_end False, tanh_out False)."""
    nlev = len(dd["ch_mult"])
    h = conv(sd, p + "conv_in.", z)
    h = vae_resblock(sd, p + "mid.block_1.", h)
    h = vae_attn(sd, p + "mid.attn_1.", h)
    h = vae_resblock(sd, p + "mid.block_2.", h)
    for lvl in reversed(range(nlev)):
        for blk in range(dd["num_res_blocks"] + 1):
            h = vae_resblock(sd, f"{p}up.{lvl}.block.{blk}.", h)
        if lvl != 0:
            h = conv(sd, f"{p}up.{lvl}.upsample.conv.", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return conv(sd, p + "conv_out.", _swish(gn(sd, p + "norm_out.", h, 1e-6)))


def vae_encode(sd: SD, cfg: dict, image: torch.Tensor, p: str = "vae.") -> torch.Tensor:
    """ControlLDM.vae_encode(sample=False): encoder, quant_conv, mode() = first half of the moments,
    times the latent scale.  model/cldm.py:107-134, model/vae.py:725-729, model/distributions.py:24-36,64-65."""
    dd = cfg["vae_cfg"]["ddconfig"]
    moments = conv(sd, p + "quant_conv.", vae_encoder(sd, dd, image, p + "encoder."), padding=0)
    mean, _ = torch.chunk(moments, 2, dim=1)
    return mean * cfg["latent_scale_factor"]


def vae_encode_sample(sd: SD, cfg: dict, image: torch.Tensor, noise: torch.Tensor, p: str = "vae.") -> torch.Tensor:
    """ControlLDM.vae_encode(sample=True), the signature's default: DiagonalGaussianDistribution.sample() =
    mean + exp(0.5 * clamp(logvar, -30, 20)) * noise, times the latent scale (model/cldm.py:131-132,
    model/distributions.py:24-41).  ``noise`` is the torch.randn(mean.shape) draw the reference makes on the host."""
    dd = cfg["vae_cfg"]["ddconfig"]
    moments = conv(sd, p + "quant_conv.", vae_encoder(sd, dd, image, p + "encoder."), padding=0)
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return (mean + std * noise) * cfg["latent_scale_factor"]


def vae_decode(sd: SD, cfg: dict, z: torch.Tensor, p: str = "vae.") -> torch.Tensor:
    """ControlLDM.vae_decode: z / scale, post_quant_conv, decoder.  model/cldm.py:136-156, model/vae.py:731-734."""
    dd = cfg["vae_cfg"]["ddconfig"]
    z = conv(sd, p + "post_quant_conv.", z / cfg["latent_scale_factor"], padding=0)
    return vae_decoder(sd, dd, z, p + "decoder.")


# ------------------------------------------------------------------------------------------
# a20  latent tiling of the ControlLDM forward
# ------------------------------------------------------------------------------------------

def sliding_windows(h: int, w: int, size: int, stride: int) -> List[Tuple[int, int, int, int]]:
    """utils/common.py:351-364 (last window snapped to the edge)."""
    def starts(n):
        s = list(range(0, n - size + 1, stride))
        if (n - size) % stride != 0:
            s.append(n - size)
        return s
    return [(hi, hi + size, wi, wi + size) for hi in starts(h) for wi in starts(w)]


def gaussian_weights(width: int, height: int) -> np.ndarray:
    """utils/common.py:151-165: var 0.01; x midpoint (w-1)/2 but y midpoint h/2 (sic)."""
    var = 0.01
    xs = np.arange(width, dtype=np.float64)
    ys = np.arange(height, dtype=np.float64)
    xp = np.exp(-(xs - (width - 1) / 2) ** 2 / (width * width) / (2 * var)) / np.sqrt(2 * np.pi * var)
    yp = np.exp(-(ys - height / 2) ** 2 / (height * height) / (2 * var)) / np.sqrt(2 * np.pi * var)
    return np.outer(yp, xp)


def tiled_cldm_forward(sd: SD, cfg: dict, x, t, cond, size: int, stride: int) -> torch.Tensor:
    """Gaussian-weighted overlap-add of per-tile forwards, c_img cropped per tile.
    utils/common.py:367-427 with the wrapper of utils/sampler.py:288-303."""
    b, c, h, w = x.shape
    out = torch.zeros_like(x)
    count = torch.zeros_like(x, dtype=torch.float32)
    wts = torch.tensor(gaussian_weights(size, size)[None, None], dtype=x.dtype)
    for hi, he, wi, we in sliding_windows(h, w, size, stride):
        tile_cond = {"c_txt": cond["c_txt"], "c_img": cond["c_img"][..., hi:he, wi:we]}
        out[..., hi:he, wi:we] += cldm_forward(sd, cfg, x[..., hi:he, wi:we], t, tile_cond) * wts
        count[..., hi:he, wi:we] += wts
    return out / count


# ------------------------------------------------------------------------------------------
# a3  sampler loop
# ------------------------------------------------------------------------------------------

def sample(sd: SD, cfg: dict, betas: np.ndarray, x_T: torch.Tensor, used_timesteps: Sequence[int],
           cond: Dict[str, torch.Tensor], noises: Sequence[torch.Tensor], tiled: bool = False,
           tile_size: int = -1, tile_stride: int = -1, return_trace: bool = False):
    """SpacedSampler.manual_sample_with_timesteps / sample with an explicit per-step noise list.
    utils/sampler.py:267-323 (loop :310-321), :206-265.  ``noises[i]`` replaces the
    torch.randn_like of step i (:199)."""
    tabs = schedule_tables(betas, used_timesteps)
    steps = tabs["timesteps"][::-1]
    total = len(steps)
    x = x_T
    trace = {"eps": [], "pred_x0": []}
    b = x.shape[0]
    for i, step in enumerate(steps):
        ts = torch.full((b,), int(step), dtype=torch.int64)
        index = torch.full((b,), total - i - 1, dtype=torch.int64)
        if tiled:
            eps = tiled_cldm_forward(sd, cfg, x, ts, cond, tile_size, tile_stride)
        else:
            eps = cldm_forward(sd, cfg, x, ts, cond)
        x, pred_x0 = p_sample_update(tabs, x, eps, noises[i], index)
        trace["eps"].append(eps)
        trace["pred_x0"].append(pred_x0)
    return (x, trace) if return_trace else x


def restore(sd: SD, cfg: dict, betas: np.ndarray, pre_res: torch.Tensor, c_txt: torch.Tensor,
            noises: Sequence[torch.Tensor], used_timesteps: Sequence[int] = (50, 100, 150, 200),
            start_t: int = 200, return_trace: bool = False):
    """The whole timed path: vae_encode -> q_sample(t=start) -> sampler -> vae_decode.
    demo.py:102-123 / main/det/test_edtr.py:121-135.  ``noises[0]`` feeds q_sample."""
    z_pre = vae_encode(sd, cfg, pre_res * 2 - 1)
    b = z_pre.shape[0]
    x_T = q_sample(betas, z_pre, torch.full((b,), start_t, dtype=torch.int64), noises[0])
    out = sample(sd, cfg, betas, x_T, used_timesteps, {"c_txt": c_txt, "c_img": z_pre}, noises[1:],
                 return_trace=return_trace)
    z, trace = out if return_trace else (out, None)
    img = vae_decode(sd, cfg, z)
    if return_trace:
        trace.update(z_pre=z_pre, x_T=x_T, z=z)
        return img, trace
    return img


# ------------------------------------------------------------------------------------------
# f1  wavelet colour fix (adjacent "next" row)
# ------------------------------------------------------------------------------------------

def _wavelet_blur(img: torch.Tensor, radius: int) -> torch.Tensor:
    """3x3 binomial kernel, dilation = radius, replicate pad.  utils/common.py:99-118."""
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=img.dtype)
    k = k[None, None].repeat(3, 1, 1, 1)
    return F.conv2d(F.pad(img, (radius,) * 4, mode="replicate"), k, groups=3, dilation=radius)


def wavelet_decomposition(img: torch.Tensor, levels: int = 5):
    """utils/common.py:121-133."""
    high = torch.zeros_like(img)
    low = img
    for i in range(levels):
        low = _wavelet_blur(img, 2 ** i)
        high = high + (img - low)
        img = low
    return high, low


def wavelet_reconstruction(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    """content high-frequency + style low-frequency.  utils/common.py:136-147."""
    return wavelet_decomposition(content)[0] + wavelet_decomposition(style)[1]


# ------------------------------------------------------------------------------------------
# a21  tiled VAE (VAEHook): padded tiles, GroupNorm statistics pooled across tiles
# ------------------------------------------------------------------------------------------

def _best_tile_size(lower: int, upper: int) -> int:
    """utils/tilevae/tilevae.py:325-338."""
    divider = 32
    while divider >= 2:
        rem = lower % divider
        if rem == 0:
            return lower
        cand = lower - rem + divider
        if cand <= upper:
            return cand
        divider //= 2
    return lower


def split_tiles(h: int, w: int, tile_size: int, is_decoder: bool):
    """Input / output bounding boxes [x1, x2, y1, y2] of every tile.  utils/tilevae/tilevae.py:340-395
    (pad = 11 latent px for the decoder, 32 image px for the encoder, :315)."""
    pad = 11 if is_decoder else 32
    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    th = _best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
    tw = _best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
    ins, outs = [], []
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * tw, min(pad + (j + 1) * tw, w), pad + i * th, min(pad + (i + 1) * th, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w,
                  ib[2] if ib[2] > pad else 0, ib[3] if ib[3] < h - pad else h]
            outs.append([v * 8 if is_decoder else v // 8 for v in ob])
            ins.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return ins, outs


def _vae_tile_tasks(sd: SD, dd: dict, p: str, is_decoder: bool, x: torch.Tensor):
    """One tile's pass through the encoder / decoder as a generator that suspends at every GroupNorm: it yields
    (tensor, norm prefix) and is resumed with the normalised tensor (affine applied, no activation).
    Task order of build_task_queue / resblock2task / attn2task, utils/tilevae/tilevae.py:77-165."""
    nlev = len(dd["ch_mult"])

    def res(q, h):
        skip = conv(sd, q + "nin_shortcut.", h, padding=0) if (q + "nin_shortcut.weight") in sd else h
        n = yield (h, q + "norm1.")
        h = conv(sd, q + "conv1.", F.silu(n))
        n = yield (h, q + "norm2.")
        return conv(sd, q + "conv2.", F.silu(n)) + skip

    def attn(q, h):
        n = yield (h, q + "norm.")
        b, c, hh, ww = n.shape
        qq, kk, vv = (conv(sd, q + nm + ".", n, padding=0).reshape(b, c, hh * ww).transpose(1, 2) for nm in "qkv")
        w_ = torch.softmax((qq @ kk.transpose(1, 2)) * (1.0 / math.sqrt(c)), dim=-1)
        o = (w_ @ vv).transpose(1, 2).reshape(b, c, hh, ww)
        return h + conv(sd, q + "proj_out.", o, padding=0)      # tile-LOCAL attention (utils/tilevae/attn.py:85-115)

    h = conv(sd, p + "conv_in.", x)
    if is_decoder:
        h = yield from res(p + "mid.block_1.", h)
        h = yield from attn(p + "mid.attn_1.", h)
        h = yield from res(p + "mid.block_2.", h)
        for lvl in reversed(range(nlev)):
            for blk in range(dd["num_res_blocks"] + 1):
                h = yield from res(f"{p}up.{lvl}.block.{blk}.", h)
            if lvl != 0:
                h = conv(sd, f"{p}up.{lvl}.upsample.conv.", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    else:
        for lvl in range(nlev):
            for blk in range(dd["num_res_blocks"]):
                h = yield from res(f"{p}down.{lvl}.block.{blk}.", h)
            if lvl != nlev - 1:
                h = conv(sd, f"{p}down.{lvl}.downsample.conv.", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
        h = yield from res(p + "mid.block_1.", h)
        h = yield from attn(p + "mid.attn_1.", h)
        h = yield from res(p + "mid.block_2.", h)
    n = yield (h, p + "norm_out.")
    return conv(sd, p + "conv_out.", F.silu(n))


def tiled_vae_net(sd: SD, dd: dict, p: str, x: torch.Tensor, tile_size: int, is_decoder: bool) -> torch.Tensor:
    """VAEHook.__call__ / vae_tile_forward (non-fast mode), utils/tilevae/tilevae.py:317-323, 452-579.
    Every tile advances to its next GroupNorm; the per-tile (variance, mean) of each (image, group) are averaged with
    weights proportional to the tile's pixel count (GroupNormParam.summary, :263-278 — between-tile mean spread is
    ignored on purpose) and the shared statistics normalise every tile (custom_group_norm, :188-215, eps 1e-6)."""
    pad = 11 if is_decoder else 32
    b, _, hh, ww = x.shape
    if max(hh, ww) <= 2 * pad + tile_size:
        return vae_decoder(sd, dd, x, p) if is_decoder else vae_encoder(sd, dd, x, p)
    ins, outs = split_tiles(hh, ww, tile_size, is_decoder)
    gens = [_vae_tile_tasks(sd, dd, p, is_decoder, x[:, :, ib[2]:ib[3], ib[0]:ib[1]]) for ib in ins]
    pending = [next(g) for g in gens]
    results = [None] * len(gens)
    while any(r is None for r in results):
        live = [i for i, r in enumerate(results) if r is None]
        stats, pix = [], []
        for i in live:
            t, _ = pending[i]
            c = t.shape[1]
            r = t.reshape(b * 32, (c // 32) * t.shape[2] * t.shape[3])
            stats.append((r.var(dim=1, unbiased=False), r.mean(dim=1)))
            pix.append(float(t.shape[2] * t.shape[3]))
        wts = torch.tensor(pix) / max(pix)
        wts = wts / wts.sum()
        var = sum(wgt * s[0] for wgt, s in zip(wts, stats))
        mean = sum(wgt * s[1] for wgt, s in zip(wts, stats))
        for i in live:
            t, q = pending[i]
            c = t.shape[1]
            shp = (b, 32, 1, 1, 1)
            n = ((t.reshape(b, 32, c // 32, t.shape[2], t.shape[3]) - mean.reshape(shp))
                 / torch.sqrt(var.reshape(shp) + 1e-6)).reshape(t.shape)
            n = n * sd[q + "weight"].reshape(1, -1, 1, 1) + sd[q + "bias"].reshape(1, -1, 1, 1)
            try:
                pending[i] = gens[i].send(n)
            except StopIteration as done:
                results[i] = done.value
    out = None
    for tile, ib, ob in zip(results, ins, outs):
        if out is None:
            out = torch.zeros((b, tile.shape[1], hh * 8 if is_decoder else hh // 8, ww * 8 if is_decoder else ww // 8))
        pb = [v * 8 if is_decoder else v // 8 for v in ib]
        mg = [ob[k] - pb[k] for k in range(4)]      # crop_valid_region, :218-229
        out[:, :, ob[2]:ob[3], ob[0]:ob[1]] = tile[:, :, mg[2]:tile.shape[2] + mg[3], mg[0]:tile.shape[3] + mg[1]]
    return out


def vae_encode_tiled(sd: SD, cfg: dict, image: torch.Tensor, tile_size: int, p: str = "vae.") -> torch.Tensor:
    """ControlLDM.vae_encode(sample=False, tiled=True), model/cldm.py:114-134."""
    dd = cfg["vae_cfg"]["ddconfig"]
    h = tiled_vae_net(sd, dd, p + "encoder.", image, tile_size, False)
    mean, _ = torch.chunk(conv(sd, p + "quant_conv.", h, padding=0), 2, dim=1)
    return mean * cfg["latent_scale_factor"]


def vae_decode_tiled(sd: SD, cfg: dict, z: torch.Tensor, tile_size: int, p: str = "vae.") -> torch.Tensor:
    """ControlLDM.vae_decode(tiled=True), model/cldm.py:142-156."""
    dd = cfg["vae_cfg"]["ddconfig"]
    z = conv(sd, p + "post_quant_conv.", z / cfg["latent_scale_factor"], padding=0)
    return tiled_vae_net(sd, dd, p + "decoder.", z, tile_size, True)


# ==============================================================================================
# CLIP text tower: reference model/clip.py:37-58, model/open_clip/transformer.py:199-254,
# model/open_clip/model.py build_attention_mask (causal).  `sd` holds the reference keys under prefix `p`.
# ==============================================================================================
def clip_text_forward(sd: SD, text_cfg: dict, tokens: torch.Tensor, layer_idx: int = 1, p: str = "clip.") -> torch.Tensor:
    """tokens int64 [B, L] -> fp32 [B, L, W]: token + positional embedding, the first (layers - layer_idx) pre-LN
    causal transformer blocks (layer "penultimate" = layer_idx 1 skips the last block), ln_final."""
    W, heads, layers = text_cfg["width"], text_cfg["heads"], text_cfg["layers"]
    x = sd[p + "model.token_embedding.weight"][tokens] + sd[p + "model.positional_embedding"]
    B, Lc, _ = x.shape
    mask = torch.full((Lc, Lc), float("-inf")).triu_(1)
    for i in range(layers - layer_idx):
        q = f"{p}model.transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), sd[q + "ln_1.weight"], sd[q + "ln_1.bias"], 1e-5)
        qkv = F.linear(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"])
        qh, kh, vh = (t.reshape(B, Lc, heads, W // heads).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(W // heads) + mask, dim=-1) @ vh
        att = att.transpose(1, 2).reshape(B, Lc, W)
        x = x + F.linear(att, sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"])
        h = F.layer_norm(x, (W,), sd[q + "ln_2.weight"], sd[q + "ln_2.bias"], 1e-5)
        h = F.gelu(F.linear(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"]))
        x = x + F.linear(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"])
    return F.layer_norm(x, (W,), sd[p + "model.ln_final.weight"], sd[p + "model.ln_final.bias"], 1e-5)


# ------------------------------------------------------------------------------------------
# SwinIR pre-restoration: the network that produces `pre_res`, the input of the path
# ------------------------------------------------------------------------------------------

def swin_relative_index(ws: int) -> np.ndarray:
    """[ws*ws, ws*ws] row of the (2ws-1)^2 bias table used for the pair (query i, key j): (dy+ws-1)*(2ws-1) + dx+ws-1
    with (dy, dx) = position(i) - position(j).  model/swinir.py:96-108."""
    ys, xs = np.divmod(np.arange(ws * ws), ws)
    return (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)


def swin_shift_mask(H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """[nW, ws*ws, ws*ws] additive mask of the shifted-window attention: after the cyclic shift a window may hold pixels
    of up to 3 x 3 disjoint image regions (bands [0, n-ws), [n-ws, n-shift), [n-shift, n) per axis); pairs from different
    regions get -100.  model/swinir.py:222-243."""
    def band(n):
        i = np.arange(n)
        return np.where(i < n - ws, 0, np.where(i < n - shift, 1, 2))
    label = band(H)[:, None] * 3 + band(W)[None, :]
    win = label.reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    return np.where(win[:, None, :] != win[:, :, None], -100.0, 0.0).astype(np.float32)


def _windows(x: torch.Tensor, ws: int) -> torch.Tensor:
    """[B, H, W, C] -> [B * nW, ws*ws, C], windows in row-major order.  model/swinir.py:37-49."""
    B, H, W, C = x.shape
    return x.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _unwindows(w: torch.Tensor, ws: int, B: int, H: int, W: int) -> torch.Tensor:
    """inverse of _windows.  model/swinir.py:52-66."""
    return w.reshape(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def swin_block(sd: SD, p: str, x: torch.Tensor, H: int, W: int, heads: int, ws: int, shift: int) -> torch.Tensor:
    """One Swin transformer layer on tokens [B, H*W, C]: x += proj(window-attention(LN(x))) ; x += MLP(LN(x)).
    model/swinir.py:245-285 (block), :120-151 (attention), :28-34 (MLP)."""
    B, _, C = x.shape
    d = C // heads
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5).reshape(B, H, W, C)
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    win = _windows(h, ws)                                                    # [B*nW, N, C]
    qkv = F.linear(win, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    q, k, v = qkv.reshape(-1, ws * ws, 3, heads, d).permute(2, 0, 3, 1, 4)    # each [B*nW, heads, N, d]
    att = (q * d ** -0.5) @ k.transpose(-1, -2)
    table = sd[p + "attn.relative_position_bias_table"]                     # [(2ws-1)^2, heads]
    att = att + table[torch.from_numpy(swin_relative_index(ws))].permute(2, 0, 1)
    if shift:
        m = torch.from_numpy(swin_shift_mask(H, W, ws, shift))               # [nW, N, N]
        att = (att.reshape(B, -1, heads, ws * ws, ws * ws) + m[None, :, None]).reshape(-1, heads, ws * ws, ws * ws)
    o = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).reshape(-1, ws * ws, C)
    o = _unwindows(F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]), ws, B, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    x = x + o.reshape(B, H * W, C)
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


SWINIR_RGB_MEAN = (0.4488, 0.4371, 0.4040)


def swinir_forward(sd: SD, cfg: dict, x: torch.Tensor, p: str = "swinir.") -> torch.Tensor:
    """image [B, 3, H, W] in [0, 1] -> pre-restored image, same size: the configuration of configs/det/demo.yaml:2-18
    (pixel-unshuffle by `sf` in front, 1conv residual connection, "nearest+conv" upsampler back to full size).
    model/swinir.py:856-894 (forward), :841-854 (features), :487-488 (RSTB)."""
    if cfg.get("upsampler") != "nearest+conv" or not cfg.get("unshuffle") or cfg.get("resi_connection", "1conv") != "1conv":
        raise NotImplementedError("oracle covers the shipped SwinIR configuration only")
    ws, sf, C = cfg["window_size"], cfg["sf"], cfg["embed_dim"]
    rng = float(cfg.get("img_range", 1.0))
    H0, W0 = x.shape[2:]
    x = F.pad(x, (0, (-W0) % ws, 0, (-H0) % ws), mode="reflect")            # check_image_size (:834-839), image space
    mean = torch.tensor(SWINIR_RGB_MEAN if cfg.get("in_chans", 3) == 3 else (0.0,)).view(1, -1, 1, 1)
    x = (x - mean) * rng
    f0 = conv(sd, p + "conv_first.1.", F.pixel_unshuffle(x, cfg["unshuffle_scale"]))
    B, _, H, W = f0.shape
    t = f0.flatten(2).transpose(1, 2)                                           # tokens [B, H*W, C]
    if cfg.get("patch_norm", True):
        t = F.layer_norm(t, (C,), sd[p + "patch_embed.norm.weight"], sd[p + "patch_embed.norm.bias"], 1e-5)
    for i, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        r = t
        for j in range(depth):
            r = swin_block(sd, f"{p}layers.{i}.residual_group.blocks.{j}.", r, H, W, heads, ws, 0 if j % 2 == 0 else ws // 2)
        r = conv(sd, f"{p}layers.{i}.conv.", r.transpose(1, 2).reshape(B, C, H, W))
        t = t + r.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    f = conv(sd, p + "conv_after_body.", t.transpose(1, 2).reshape(B, C, H, W)) + f0
    f = F.leaky_relu(conv(sd, p + "conv_before_upsample.0.", f), 0.01)       # nn.LeakyReLU() default slope (:776)
    ups = {2: ["conv_up1."], 4: ["conv_up1.", "conv_up2."], 8: ["conv_up1.", "conv_up2.", "conv_up3."]}[sf]
    for name in ups:
        f = F.leaky_relu(conv(sd, p + name, F.interpolate(f, scale_factor=2, mode="nearest")), 0.2)
    f = conv(sd, p + "conv_last.", F.leaky_relu(conv(sd, p + "conv_hr.", f), 0.2))
    return (f / rng + mean)[:, :, :H0 * sf, :W0 * sf]
