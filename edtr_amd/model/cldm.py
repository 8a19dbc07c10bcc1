"""Host-side mirror of the reference's `model.cldm.ControlLDM` (reference model/cldm.py:17-194) and of the
modules it owns, with the SAME constructor arguments, attribute names, method signatures and state-dict keys —
but every forward is a pre-built program of libedtr_hip launches (edtr_amd/engine.py, nets.py).

There is no CPU / PyTorch-operator fallback: calling a forward on a CPU-resident module raises.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Set, Tuple

import torch
from torch import nn

from .. import arch, nets
from .. import ops as ops_mod
from ..engine import Act, Arena, Emitter, EngineCache, Program, WeightStore
from .clip import FrozenOpenCLIPEmbedder
from .params import ParamTree, params_fingerprint

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}


def default_compute_dtype() -> torch.dtype:
    return _DTYPES[os.environ.get("EDTR_AMD_DTYPE", "bf16").lower()]


PRECISIONS = ("fast", "mixed", "high", "hybrid", "robust")

# The hybrid parity mode (round 6): what each SECTION of the path runs.  Plain fp16 storage is already
# inside the north-star 1e-3 on the final LATENT (8.7e-4 at full size); the excess on the image (1.5e-3) is added by the decoder, whose
# few stream-carrying convolutions pass their operand roundings straight into the pixels (decoder alone in fp16 behind an exact
# latent: 1.4e-3).  So: encoder and denoiser in the fast fp16 mode, the DECODER in the mixed mode (fp32 stream, three-part products on
# its stream carriers: edtr_amd/precision.py).  Measured at full size on one device (profiles/r06/hybrid_sweep*.log): latent 8.7e-4,
# image 8.9e-4 at 102.5 images/s against 91.6 for the all-mixed mode (5.4e-4 / 6.1e-4) and 119.1 for bf16 (6.1e-3 / 1.2e-2); the
# encoder in the mixed mode too buys 8.2e-4 / 8.5e-4 for 5 images/s; every cheaper decoder policy (two parts or one on any carrier)
# leaves the image above 9.6e-4.  The encoded latent z_pre itself is at fp16's 1.4e-3 — an intermediate, damped by the sampler.
# EDTR_AMD_HYBRID (JSON: {"cldm": "fast16" | "fastbf16" | "mixed" | "high", "vae.encode": ..., "vae.decode": ...}) overrides the
# table for experiments (tools/exp/r06_hybrid_sweep.py).
HYBRID_SECTIONS = {"cldm": "fast16", "vae.encode": "fast16", "vae.decode": "mixed"}
_SECTION_MODES = {"fast16": ("fast", torch.float16), "fastbf16": ("fast", torch.bfloat16), "mixed": ("mixed", None), "high": ("high", None)}


def hybrid_sections() -> Dict[str, str]:
    table = dict(HYBRID_SECTIONS)
    env = os.environ.get("EDTR_AMD_HYBRID")
    if env:
        import json
        for k, v in json.loads(env).items():
            if k not in table or v not in _SECTION_MODES:
                raise ValueError(f"EDTR_AMD_HYBRID: {k!r} -> {v!r} (sections {sorted(table)}, modes {sorted(_SECTION_MODES)})")
            table[k] = v
    return table


def section_mode(precision: str, compute_dtype, section: str):
    """(precision mode, compute dtype) that ``section`` ("cldm" | "vae.encode" | "vae.decode") runs under ``precision``."""
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
    if precision == "robust":          # the mixed mode's machinery under the robust allocation (edtr_amd/precision.py: robust_policy)
        return "mixed", compute_dtype
    if precision != "hybrid":
        return precision, compute_dtype
    mode, dt = _SECTION_MODES[hybrid_sections()[section]]
    return mode, (dt or compute_dtype)


def default_precision() -> str:
    """"hybrid" / "robust": the two round-6 parity modes (see HYBRID_SECTIONS below and precision.robust_policy).
    "fast": 16-bit activation storage in `compute_dtype` (the throughput modes).  "high": the robust parity mode — fp32
    activation stream, every convolution / linear as a bf16 split-3 product with fp32 accumulation (~16 mantissa bits per
    operand), fp16 attention operands.  "mixed": the fast parity mode — the same fp32 stream, fp16 operands and a per-layer
    number of products (edtr_amd/precision.py).  `compute_dtype` is ignored by the two parity modes.  EDTR_AMD_PRECISION
    selects the default."""
    p = os.environ.get("EDTR_AMD_PRECISION", "fast").lower()
    if p not in PRECISIONS:
        raise ValueError(f"EDTR_AMD_PRECISION must be one of {PRECISIONS}, got {p!r}")
    return p


def _store_dtype(precision: str, compute_dtype, section: str = "cldm"):
    mode, dt = section_mode(precision, compute_dtype, section)
    return {"high": ops_mod.F32S, "mixed": ops_mod.MIXED}.get(mode, dt)


def _policy_for(owner, section: str = "cldm"):
    """The mixed-mode precision policy of ``section`` (None where the section does not run the mixed mode)."""
    from ..precision import mixed_policy, robust_policy
    mode, _ = section_mode(owner.precision, owner.compute_dtype, section)
    if mode != "mixed":
        return None
    return owner.precision_policy or (robust_policy() if owner.precision == "robust" else mixed_policy())


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: the EDTR MI355X path runs only on a ROCm GPU (tensor is on {t.device}); "
                           "there is no CPU fallback. Move the module and inputs to 'cuda'.")


class NansException(Exception):
    """Raised by the tiled VAE when a tile came out NaN (reference utils/tilevae/tilevae.py:62-69,435,548)."""


def disabled_train(self: nn.Module, mode: bool = True) -> nn.Module:
    return self


# ----------------------------------------------------------------------------------------------
# parameter-holding modules (reference class names; forward() of the parts is not on the hot path)
# ----------------------------------------------------------------------------------------------
class _NetPart(ParamTree):
    """A ControlNet / ControlledUnetModel parameter tree that can also run on its own (the reference calls the two nets
    as separate modules, model/cldm.py:169-193; ControlLDM.forward here fuses them into one program and does not come
    through these forwards).  The standalone programs re-project the context every call: they exist for API parity and
    for checking the 13 control tensors, not for speed."""

    _prefix = ""

    def _init_part(self) -> None:
        self.compute_dtype = default_compute_dtype()
        self.precision = default_precision()
        self._part_store = None
        self._part_fp = None
        self._part_engines = EngineCache(lambda e: e.prog.release_graph())
        self.precision_policy = None      # mixed mode: a precision.PrecisionPolicy (None = the shipped allocation)

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _policy(self):
        return _policy_for(self, "cldm")

    def _mode(self):
        """(precision mode, compute dtype) of the programs of this net (the hybrid mode's denoiser section)."""
        return section_mode(self.precision, self.compute_dtype, "cldm")

    def _store(self) -> WeightStore:
        pol = self._policy()
        fp = (params_fingerprint(self), self._mode(), pol.key() if pol else None)
        if fp != self._part_fp:
            self._part_engines.drop_all()
            self._part_store, self._part_fp = None, fp
        if self._part_store is None:
            self._part_store = WeightStore(self.flat_params(self._prefix), _store_dtype(self.precision, self.compute_dtype),
                                           self._device())
        return self._part_store


class ControlledUnetModel(_NetPart):
    """Parameters of reference model/controlnet.py:18 (ControlledUnetModel = UNetModel, model/unet.py:361)."""

    _prefix = "unet."

    def __init__(self, **cfg):
        self.cfg = dict(cfg)
        self.arch = arch.unet_arch(self.cfg, controlnet=False)
        super().__init__(arch.unet_param_spec(self.arch), unet_like=True)
        self.model_channels = self.arch.model_channels
        self.dtype = torch.float32
        self._init_part()

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
        """reference model/controlnet.py:20-41: eps from the latent, the timesteps, the context and the (already
        scaled) 13 control tensors.  ``control`` is consumed (popped) like the reference does."""
        _require_gpu(x, "ControlledUnetModel.forward")
        B, _, h, w = x.shape
        store = self._store()
        mode = "none" if control is None else ("mid" if only_mid_control else "all")
        eng = self._part_engines.fetch((B, h, w, context.shape[1], mode),
                                       lambda: UnetPartEngine(self, store, B, h, w, context.shape[1], mode))
        ctrl = None
        if control is not None:
            ctrl = list(control)
            del control[:]
        return eng.run(x, timesteps, context, ctrl).clone()


class ControlNet(_NetPart):
    """Parameters of reference model/controlnet.py:44 (ControlNet)."""

    _prefix = "controlnet."

    def __init__(self, **cfg):
        self.cfg = dict(cfg)
        self.arch = arch.unet_arch(self.cfg, controlnet=True)
        super().__init__(arch.unet_param_spec(self.arch), unet_like=True)
        self.model_channels = self.arch.model_channels
        self.dtype = torch.float32
        self._init_part()

    @torch.no_grad()
    def forward(self, x, hint, timesteps, context, **kwargs) -> List[torch.Tensor]:
        """reference model/controlnet.py:263-277: the 13 control tensors (fp32 NCHW, unscaled)."""
        _require_gpu(x, "ControlNet.forward")
        B, _, h, w = x.shape
        store = self._store()
        eng = self._part_engines.fetch((B, h, w, context.shape[1]),
                                       lambda: ControlNetPartEngine(self, store, B, h, w, context.shape[1]))
        return [o.clone() for o in eng.run(x, hint, timesteps, context)]


class AutoencoderKL(ParamTree):
    """Parameters of reference model/vae.py:681 (AutoencoderKL: encoder, decoder, quant_conv, post_quant_conv)."""

    def __init__(self, ddconfig, embed_dim, train_encoder=False, train_decoder=False):
        self.cfg = dict(ddconfig=dict(ddconfig), embed_dim=embed_dim)
        assert ddconfig["double_z"]
        super().__init__(arch.vae_param_spec(self.cfg), unet_like=False)
        self.embed_dim = embed_dim
        self.train_encoder, self.train_decoder = train_encoder, train_decoder

    def forward(self, *args, **kwargs):
        raise NotImplementedError("use ControlLDM.vae_encode / vae_decode")


# ----------------------------------------------------------------------------------------------
# engines: one pre-built program set per input shape
# ----------------------------------------------------------------------------------------------
class CldmEngine:
    """ControlNet + ControlledUNet for a fixed (B, h, w): static input/output buffers, a context program
    (cross-attention K / V^T, run only when c_txt changes) and the per-step program."""

    def __init__(self, owner: "ControlLDM", B: int, h: int, w: int, nctx: int):
        dev = owner._device()
        mode, dt = section_mode(owner.precision, owner.compute_dtype, "cldm")
        pol = _policy_for(owner, "cldm")
        self.B, self.h, self.w, self.nctx = B, h, w, nctx
        self.arena = Arena(dev)
        store = owner._store("cldm")
        ua, ca = owner.unet.arch, owner.controlnet.arch
        f32 = torch.float32
        self.x_in = torch.zeros((B, ua.in_channels, h, w), dtype=f32, device=dev)
        self.hint_in = torch.zeros((B, ca.hint_channels, h, w), dtype=f32, device=dev)
        self.t_in = torch.zeros((B,), dtype=torch.int64, device=dev)
        self.ctx_in = torch.zeros((B, nctx, ua.context_dim), dtype=f32, device=dev)
        self.eps_out = torch.zeros((B, ua.out_channels, h, w), dtype=f32, device=dev)
        self.ctx_key = None

        # ---- context program
        self.ctx_prog = Program("cldm.context")
        em = Emitter(self.ctx_prog, self.arena, store, dt, mode, pol)
        ctx16 = em.cast_flat(self.ctx_in, B * nctx * ua.context_dim).view(B * nctx, ua.context_dim)
        self.kv_c = nets.emit_context_kv(em, "controlnet.", ca, ctx16, B, nctx)
        self.kv_u = nets.emit_context_kv(em, "unet.", ua, ctx16, B, nctx)

        # ---- step program
        self.step_prog = Program("cldm.step")
        em = Emitter(self.step_prog, self.arena, store, dt, mode, pol)
        hw = h * w
        cin_c = ca.in_channels + ca.hint_channels
        x8c = em.new(B * hw, arch_round8(cin_c))
        em.to_nhwc(self.x_in, B, ca.in_channels, hw, x8c, coff=0)
        em.to_nhwc(self.hint_in, B, ca.hint_channels, hw, x8c, coff=ca.in_channels,
                   pad_to=arch_round8(cin_c) - ca.in_channels)
        x8u = em.new(B * hw, arch_round8(ua.in_channels))
        em.to_nhwc(self.x_in, B, ua.in_channels, hw, x8u, coff=0, pad_to=arch_round8(ua.in_channels))
        tab_c, offs_c = nets.emit_time_rows(em, "controlnet.", ca, self.t_in, B)
        tab_u, offs_u = nets.emit_time_rows(em, "unet.", ua, self.t_in, B)
        # ControlNet (lane 1, own arena) is independent of the UNet encoder + middle block (lane 0): two graph branches
        self.arena_cn = Arena(dev)
        em_cn = Emitter(self.step_prog, self.arena_cn, store, dt, mode, pol)
        self.step_prog.fork()
        self.step_prog.set_lane(1)
        ctrl = nets.emit_controlnet(em_cn, "controlnet.", ca, Act(x8c, B, h, w, x8c.shape[1]), tab_c, offs_c, self.kv_c,
                                    list(owner.control_scales))
        self.step_prog.set_lane(0)
        eps = nets.emit_unet(em, "unet.", ua, Act(x8u, B, h, w, x8u.shape[1]), tab_u, offs_u, self.kv_u, ctrl,
                             before_control=self.step_prog.join)
        em.to_nchw(eps, B, ua.out_channels, hw, self.eps_out)

    def set_context(self, c_txt: torch.Tensor) -> None:
        """Re-run the context program unless ``c_txt`` is the very tensor OBJECT (same version) the cached K / V^T came
        from.  The cache holds a reference to that tensor, so its storage cannot be recycled for another prompt's
        embedding while the entry is alive (an address/version key alone would alias a new same-shaped tensor)."""
        held = self.ctx_key
        if held is not None and held[0] is c_txt and held[1] == c_txt._version:
            return
        src = c_txt.expand(self.B, -1, -1) if (c_txt.shape[0] == 1 and self.B > 1) else c_txt
        self.ctx_in.copy_(src)
        self.ctx_prog.run()
        self.ctx_key = (c_txt, c_txt._version)

    def step(self, x: torch.Tensor, t: torch.Tensor, c_txt: torch.Tensor, c_img: torch.Tensor) -> torch.Tensor:
        self.set_context(c_txt)
        self.x_in.copy_(x)
        self.hint_in.copy_(c_img)
        self.t_in.copy_(t)
        self.step_prog.run()
        return self.eps_out


class _PartEngineBase:
    def _common(self, part: _NetPart, store: WeightStore, B: int, h: int, w: int, nctx: int, name: str):
        dev = part._device()
        a = part.arch
        self.arena = Arena(dev)
        self.prog = Program(name)
        mode, dt = part._mode()
        em = Emitter(self.prog, self.arena, store, dt, mode, part._policy())
        f32 = torch.float32
        self.x_in = torch.zeros((B, a.in_channels, h, w), dtype=f32, device=dev)
        self.t_in = torch.zeros((B,), dtype=torch.int64, device=dev)
        self.ctx_in = torch.zeros((B, nctx, a.context_dim), dtype=f32, device=dev)
        ctx16 = em.cast_flat(self.ctx_in, B * nctx * a.context_dim).view(B * nctx, a.context_dim)
        kv = nets.emit_context_kv(em, part._prefix, a, ctx16, B, nctx)
        table, offs = nets.emit_time_rows(em, part._prefix, a, self.t_in, B)
        return em, kv, table, offs

    def _load(self, x, t, ctx):
        self.x_in.copy_(x)
        self.t_in.copy_(t)
        self.ctx_in.copy_(ctx.expand(self.x_in.shape[0], -1, -1) if ctx.shape[0] == 1 else ctx)


class ControlNetPartEngine(_PartEngineBase):
    """ControlNet alone for a fixed (B, h, w): 13 fp32 NCHW outputs."""

    def __init__(self, part: "ControlNet", store: WeightStore, B: int, h: int, w: int, nctx: int):
        em, kv, table, offs = self._common(part, store, B, h, w, nctx, "controlnet.alone")
        a = part.arch
        dev = part._device()
        self.hint_in = torch.zeros((B, a.hint_channels, h, w), dtype=torch.float32, device=dev)
        cin = a.in_channels + a.hint_channels
        x8 = em.new(B * h * w, arch_round8(cin))
        em.to_nhwc(self.x_in, B, a.in_channels, h * w, x8, coff=0)
        em.to_nhwc(self.hint_in, B, a.hint_channels, h * w, x8, coff=a.in_channels, pad_to=arch_round8(cin) - a.in_channels)
        ctrl = nets.emit_controlnet(em, part._prefix, a, Act(x8, B, h, w, x8.shape[1]), table, offs, kv, [1.0] * 13)
        self.outs = []
        for c in ctrl:
            o = torch.zeros((B, c.C, c.H, c.W), dtype=torch.float32, device=dev)
            em.to_nchw(c.t, B, c.C, c.H * c.W, o)
            self.outs.append(o)

    def run(self, x, hint, t, ctx):
        self._load(x, t, ctx)
        self.hint_in.copy_(hint)
        self.prog.run()
        return self.outs


class UnetPartEngine(_PartEngineBase):
    """ControlledUnetModel alone for a fixed (B, h, w): control tensors come in as fp32 NCHW."""

    def __init__(self, part: "ControlledUnetModel", store: WeightStore, B: int, h: int, w: int, nctx: int, mode: str):
        em, kv, table, offs = self._common(part, store, B, h, w, nctx, "unet.alone")
        a = part.arch
        dev = part._device()
        x8 = em.new(B * h * w, arch_round8(a.in_channels))
        em.to_nhwc(self.x_in, B, a.in_channels, h * w, x8, coff=0, pad_to=arch_round8(a.in_channels))
        self.ctrl_in: List[Optional[torch.Tensor]] = []
        acts: Optional[List[Optional[Act]]] = None
        if mode != "none":
            acts = []
            shapes = arch.control_shapes(a, h, w)
            for i, (C, hh, ww) in enumerate(shapes):
                if mode == "mid" and i != len(shapes) - 1:
                    self.ctrl_in.append(None)
                    acts.append(None)
                    continue
                src = torch.zeros((B, C, hh, ww), dtype=torch.float32, device=dev)
                dst = em.new(B * hh * ww, C)
                em.to_nhwc(src, B, C, hh * ww, dst)
                self.ctrl_in.append(src)
                acts.append(Act(dst, B, hh, ww, C))
        self.eps_out = torch.zeros((B, a.out_channels, h, w), dtype=torch.float32, device=dev)
        eps = nets.emit_unet(em, part._prefix, a, Act(x8, B, h, w, x8.shape[1]), table, offs, kv, acts)
        em.to_nchw(eps, B, a.out_channels, h * w, self.eps_out)

    def run(self, x, t, ctx, control):
        self._load(x, t, ctx)
        if control is not None:
            if len(control) != len(self.ctrl_in):
                raise ValueError(f"expected {len(self.ctrl_in)} control tensors, got {len(control)}")
            for dst, src in zip(self.ctrl_in, control):
                if dst is not None:
                    dst.copy_(src)
        self.prog.run()
        return self.eps_out


def arch_round8(c: int) -> int:
    return (c + 7) // 8 * 8


class VaeEngine:
    """Encoder (+quant_conv, mode, scale) or decoder (scale^-1, post_quant_conv, decoder) for a fixed shape.
    ``tile_size`` > 0 builds the tiled form (reference utils/tilevae VAEHook, non-fast mode): padded tiles cut from the
    full NHWC tensor, GroupNorm statistics pooled across tiles, valid regions written back."""

    def __init__(self, owner: "ControlLDM", kind: str, B: int, H: int, W: int, tile_size: int = 0, sample: bool = False):
        dev = owner._device()
        section = "vae." + kind
        mode, dt = section_mode(owner.precision, owner.compute_dtype, section)
        self.arena = Arena(dev)
        store = owner._store(section)
        dd = owner.vae.cfg["ddconfig"]
        f32 = torch.float32
        self.prog = Program(f"vae.{kind}" + (".tiled" if tile_size else ""))
        self.nan_probe = None      # tiled form: (rows, cols) index tensors of the first output pixel of every tile
        self.noise_in = None       # encoder with sample=True: the N(0, 1) draw of DiagonalGaussianDistribution.sample()
        em = Emitter(self.prog, self.arena, store, dt, mode, _policy_for(owner, section))
        if kind == "encode" and os.environ.get("EDTR_AMD_BRANCH16_ENC", "1") == "0":
            # mixed mode, A/B switch: the ENCODER with its branch-internal tensors in fp32.  z_pre (the conditioning of every denoise step
            # and the start of the trajectory) is 6.9e-4 from the reference instead of 8.0e-4 of its 1e-3 budget, the final latent /
            # image hardly move (5.40e-4 / 6.07e-4 -> 5.33e-4 / 5.98e-4) and the path is 1.2 % slower (90.5 -> 89.4 images/s, one
            # device): the fp16 form stays the default
            em.branch16 = False
        sf = owner.scale_factor
        nlev = len(dd["ch_mult"])
        is_dec = kind == "decode"
        pad = 11 if is_dec else 32
        tiled = tile_size > 0 and max(H, W) > 2 * pad + tile_size     # tiny inputs run untiled (tilevae.py:317-323)
        if kind == "encode":
            self.inp = torch.zeros((B, dd["in_channels"], H, W), dtype=f32, device=dev)
            h, w = H >> (nlev - 1), W >> (nlev - 1)
            self.out = torch.zeros((B, owner.vae.embed_dim, h, w), dtype=f32, device=dev)
            cp = arch_round8(dd["in_channels"])
            x = em.new(B * H * W, cp)
            em.to_nhwc(self.inp, B, dd["in_channels"], H * W, x, pad_to=cp)
            layers = arch.vae_encoder_arch(dd)
            if not tiled:
                y = nets.emit_vae_net(em, "vae.encoder.", layers, Act(x, B, H, W, cp), final_f32=False)
            else:
                y = self._tiled(em, "vae.encoder.", layers, x, B, H, W, cp, tile_size, False, h, w)
                self.nan_probe = self._probe_index(nets.split_tiles(H, W, tile_size, False)[1], dev)
            # quant_conv 1x1 (model/vae.py:727) then DiagonalGaussianDistribution.mode() = first half (distributions.py:30,64)
            # or .sample() = mean + exp(0.5 * clamp(logvar)) * N(0, 1) (distributions.py:29-41), times scale_factor (cldm.py:131-134)
            m = em.conv(y, "vae.quant_conv.", taps=1, out_f32=True, name="vae.quant_conv")
            em.free(y)
            if sample:
                self.noise_in = torch.zeros_like(self.out)
                em.prog.add(ops_mod.make_gaussian_sample(moments=m.t, ld=m.t.stride(0), noise=self.noise_in, out=self.out, B=B,
                                                         C=owner.vae.embed_dim, HW=h * w, scale=sf))
            else:
                em.to_nchw(m.t, B, owner.vae.embed_dim, h * w, self.out, scale=sf)
        else:
            zc = dd["z_channels"]
            self.inp = torch.zeros((B, zc, H, W), dtype=f32, device=dev)
            up = 1 << (nlev - 1)
            self.out = torch.zeros((B, dd["out_ch"], H * up, W * up), dtype=f32, device=dev)
            cp = arch_round8(zc)
            z = em.new(B * H * W, cp)
            em.to_nhwc(self.inp, B, zc, H * W, z, pad_to=cp, scale=1.0 / sf)           # z / scale_factor (cldm.py:156)
            z2 = em.conv(Act(z, B, H, W, cp), "vae.post_quant_conv.", taps=1, name="vae.post_quant_conv")
            layers = arch.vae_decoder_arch(dd)
            if not tiled:
                # (final_nchw: norm_out + SiLU + conv_out + the layout change as ONE launch where edtr_conv128_out takes them)
                y = nets.emit_vae_net(em, "vae.decoder.", layers, z2, final_f32=True, final_nchw=self.out if dd["out_ch"] <= 4 else None)
            else:
                y = self._tiled(em, "vae.decoder.", layers, z2.t, B, H, W, z2.C, tile_size, True, H * up, W * up)
                self.nan_probe = self._probe_index(nets.split_tiles(H, W, tile_size, True)[1], dev)
            if y is not None:
                em.to_nchw(y.t, B, dd["out_ch"], y.H * y.W, self.out)

    @staticmethod
    def _tiled(em: Emitter, P: str, layers, full: torch.Tensor, B: int, H: int, W: int, C: int, tile_size: int,
               is_dec: bool, OH: int, OW: int) -> Act:
        """Cut padded tiles out of the full NHWC tensor, run them in GroupNorm lock-step, assemble the valid regions."""
        ins, outs = nets.split_tiles(H, W, tile_size, is_dec)
        tiles = []
        for x1, x2, y1, y2 in ins:
            th, tw = y2 - y1, x2 - x1
            t = em.new(B * th * tw, C)
            for b in range(B):      # 2-D strided copy: th rows of tw*C contiguous elements
                em.add(full[(b * H + y1) * W + x1:].reshape(-1)[: (th - 1) * W * C + tw * C].as_strided((th, tw * C), (W * C, 1)),
                       None, th, tw * C, out=t[b * th * tw:(b + 1) * th * tw].reshape(th, tw * C))
            tiles.append(Act(t, B, th, tw, C))
        res = nets.emit_vae_net_tiled(em, P, layers, tiles, final_f32=is_dec)
        Co = res[0].C
        out = em.new(B * OH * OW, Co, torch.float32 if is_dec else None)
        for r, ib, ob in zip(res, ins, outs):
            pb = [v * 8 if is_dec else v // 8 for v in ib]
            mx0, my0 = ob[0] - pb[0], ob[2] - pb[2]                 # crop_valid_region (tilevae.py:218-229)
            cw, ch = ob[1] - ob[0], ob[3] - ob[2]
            for b in range(B):
                src = r.t[(b * r.H + my0) * r.W + mx0:]
                dst = out[(b * OH + ob[2]) * OW + ob[0]:]
                if is_dec:
                    em.prog.add(ops_mod.make_copy3d(src=src, src_plane=0, src_row=r.W * Co, dst=dst, dst_plane=0,
                                                    dst_row=OW * Co, planes=1, rows=ch, cols=cw * Co))
                else:
                    em.add(src.reshape(-1)[: (ch - 1) * r.W * Co + cw * Co].as_strided((ch, cw * Co), (r.W * Co, 1)), None, ch,
                           cw * Co, out=dst.reshape(-1)[: (ch - 1) * OW * Co + cw * Co].as_strided((ch, cw * Co), (OW * Co, 1)))
            em.free(r)
        return Act(out, B, OH, OW, Co)

    @staticmethod
    def _probe_index(out_boxes, dev):
        """Index tensors (built once) of the first output pixel of every tile, for the NaN probe of run()."""
        return (torch.tensor([ob[2] for ob in out_boxes], device=dev), torch.tensor([ob[0] for ob in out_boxes], device=dev))

    def run(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.inp.copy_(x)
        if self.noise_in is not None:
            self.noise_in.copy_(noise)
        self.prog.run()
        if (self.nan_probe is not None and VaeEngine.NAN_PROBE and not torch.cuda.is_current_stream_capturing()):
            # test_for_nans(tile, "vae") of the reference (utils/tilevae/tilevae.py:66-69,435,548): one element per tile is
            # looked at (batch 0, channel 0); like there, this costs a device round trip per call (tiled VAE only).  Skipped
            # under hipGraph capture (a sync is illegal there) and when EDTR_VAE_NAN_PROBE=0 (benchmarks).
            rows, cols = self.nan_probe
            if bool(torch.isnan(self.out[0, 0, rows, cols]).any()):
                raise NansException("vae")
        return self.out

    NAN_PROBE = os.environ.get("EDTR_VAE_NAN_PROBE", "1") != "0"


# ----------------------------------------------------------------------------------------------
# ControlLDM
# ----------------------------------------------------------------------------------------------
class ControlLDM(nn.Module):
    """Drop-in for reference model/cldm.py:17 — same constructor, attributes and methods."""

    def __init__(self, unet_cfg, vae_cfg, clip_cfg, controlnet_cfg, latent_scale_factor, tail_block=False):
        super().__init__()
        if tail_block:
            raise NotImplementedError("tail_block / woSD is dead code in the reference (no caller) and is not built")
        self.unet = ControlledUnetModel(**unet_cfg)
        self.vae = AutoencoderKL(**vae_cfg)
        self.clip = FrozenOpenCLIPEmbedder(**clip_cfg)
        self.controlnet = ControlNet(**controlnet_cfg)
        self.scale_factor = latent_scale_factor
        self.control_scales = [1.0] * 13
        self.compute_dtype = default_compute_dtype()
        self.precision = default_precision()      # "fast" | "mixed" | "high" (parity modes), see default_precision()
        self.precision_policy = None              # mixed mode: a precision.PrecisionPolicy (None = the shipped allocation)
        # engines (static buffers + programs) are cached per shape AND per slot: a caller that keeps two batches in
        # flight on two HIP streams flips the slot so the batches never share a buffer (bench.py --inflight 2)
        self.engine_slot = 0
        self._weights = None
        self._fingerprint = None
        self._cldm_engines = EngineCache(lambda e: (e.step_prog.release_graph(), e.ctx_prog.release_graph()))
        self._vae_engines = EngineCache(lambda e: e.prog.release_graph())

    # -- engine plumbing ---------------------------------------------------------------------
    def _device(self) -> torch.device:
        return next(self.unet.parameters()).device

    def _policy(self):
        """The mixed-mode policy in force (the hybrid mode: of whichever sections run the mixed mode), else None."""
        for section in ("cldm", "vae.encode", "vae.decode"):
            pol = _policy_for(self, section)
            if pol is not None:
                return pol
        return None

    def _check_fresh(self) -> None:
        pol = self._policy()
        wfp = (params_fingerprint(self.unet), params_fingerprint(self.controlnet), params_fingerprint(self.vae),
               self.compute_dtype, self.precision, tuple(sorted(hybrid_sections().items())) if self.precision == "hybrid" else None)
        fp = (wfp, tuple(self.control_scales), pol.key() if pol else None)
        if fp != self._fingerprint:
            # programs depend on everything; the packed store only on the parameters and the storage format, so a change of
            # control_scales / precision policy rebuilds the programs over the SAME store (its entries are keyed by part count
            # and bias scale)
            keep = self._weights if (self._fingerprint is not None and self._fingerprint[0] == wfp) else None
            if keep is None:
                self.release_engines()      # (raises on a rank whose parameters are placeholders: see WeightStore.frozen)
            else:
                self._cldm_engines.drop_all()
                self._vae_engines.drop_all()
            self._fingerprint = fp

    def release_engines(self) -> None:
        """Drop every program / hipGraph AND the packed weights (call after writes through ``.data``)."""
        if self._weights is not None and self._weights.frozen:
            raise RuntimeError(f"cannot rebuild the engines: {self._weights.frozen} (the fp32 parameters of this rank are placeholders; "
                               "changing control_scales / precision / compute_dtype after the packed broadcast needs a "
                               "parallel.broadcast_parameters first)")
        self._cldm_engines.drop_all()
        self._vae_engines.drop_all()
        self._weights = None

    def weights_updated_in_place(self) -> None:
        """The packed weight tensors were overwritten in place (edtr_amd.parallel.broadcast_packed): programs and hipGraphs
        stay valid, but everything DERIVED from the old values — the cached cross-attention K / V^T of each engine — is stale."""
        for eng in self._cldm_engines.values():
            eng.ctx_key = None

    def _store(self, section: str = "cldm") -> WeightStore:
        """The packed weights in the storage format ``section`` runs in.  One root store (the denoiser's format); the hybrid mode's
        other formats are its siblings (WeightStore.for_dtype), so the packed broadcast and its checksum cover them."""
        if self._weights is None:
            params: Dict[str, torch.Tensor] = {}
            params.update(self.unet.flat_params("unet."))
            params.update(self.controlnet.flat_params("controlnet."))
            params.update(self.vae.flat_params("vae."))
            self._weights = WeightStore(params, _store_dtype(self.precision, self.compute_dtype), self._device())
        return self._weights.for_dtype(_store_dtype(self.precision, self.compute_dtype, section))

    def cldm_engine(self, B: int, h: int, w: int, nctx: int = 77) -> CldmEngine:
        self._check_fresh()
        key = (B, h, w, nctx, self.engine_slot)
        return self._cldm_engines.fetch(key, lambda: CldmEngine(self, B, h, w, nctx))

    def vae_engine(self, kind: str, B: int, H: int, W: int, tile_size: int = 0, sample: bool = False) -> VaeEngine:
        self._check_fresh()
        key = (kind + (".sample" if sample else ""), B, H, W, tile_size, self.engine_slot)
        return self._vae_engines.fetch(key, lambda: VaeEngine(self, kind, B, H, W, tile_size, sample))

    # -- checkpoint ingestion (reference model/cldm.py:46-105) -----------------------------------
    @torch.no_grad()
    def load_pretrained_sd(self, sd: Dict[str, torch.Tensor], is_turbo: bool = False) -> Set[str]:
        """Strict key-for-key copy of the SD checkpoint's `model.diffusion_model.*` / `first_stage_model.*` /
        `cond_stage_model.*` (turbo: `conditioner.embedders.0.*`) entries, like the reference."""
        module_map = {"unet": "model.diffusion_model", "vae": "first_stage_model",
                      "clip": "conditioner.embedders.0" if is_turbo else "cond_stage_model"}
        used: Set[str] = set()
        for name, module in (("unet", self.unet), ("vae", self.vae), ("clip", self.clip)):
            init_sd = {}
            for key in module.state_dict():
                target = f"{module_map[name]}.{key}"
                init_sd[key] = sd[target].clone()
                used.add(target)
            module.load_state_dict(init_sd, strict=True)
        for module in (self.clip, self.unet):
            module.eval()
            module.train = disabled_train.__get__(module)
            for p in module.parameters():
                p.requires_grad = False
        return set(sd.keys()) - used

    @torch.no_grad()
    def load_controlnet_from_ckpt(self, sd: Dict[str, torch.Tensor]) -> None:
        self.controlnet.load_state_dict(sd, strict=True)

    @torch.no_grad()
    def load_controlnet_from_unet(self) -> Tuple[Set[str], Set[str]]:
        unet_sd = self.unet.state_dict()
        scratch = self.controlnet.state_dict()
        init_sd, with_zero, with_scratch = {}, set(), set()
        for key, this in scratch.items():
            if key in unet_sd:
                target = unet_sd[key]
                if this.size() == target.size():
                    init_sd[key] = target.clone()
                else:   # the 8-channel input conv: UNet weights for the latent half, zeros for the hint half
                    extra = this.size(1) - target.size(1)
                    oc, _, kh, kw = this.size()
                    init_sd[key] = torch.cat((target, torch.zeros((oc, extra, kh, kw), dtype=target.dtype,
                                                                  device=target.device)), dim=1)
                    with_zero.add(key)
            else:
                init_sd[key] = this.clone()
                with_scratch.add(key)
        self.controlnet.load_state_dict(init_sd, strict=True)
        return with_zero, with_scratch

    # -- VAE (reference model/cldm.py:107-156) ------------------------------------------------------
    @torch.no_grad()
    def vae_encode(self, image: torch.Tensor, sample: bool = True, tiled: bool = False, tile_size: int = -1) -> torch.Tensor:
        _require_gpu(image, "vae_encode")
        B, _, H, W = image.shape
        if tiled and tile_size <= 0:
            raise ValueError("vae_encode(tiled=True) needs a positive tile_size (image pixels)")
        eng = self.vae_engine("encode", B, H, W, tile_size if tiled else 0, sample=bool(sample))
        if not sample:
            return eng.run(image).clone()
        # DiagonalGaussianDistribution.sample(): the reference draws torch.randn(mean.shape) on the HOST generator and moves it
        # to the device (model/distributions.py:38-41); the same call here, so a seeded run sees the same draw
        noise = torch.randn(tuple(eng.out.shape)).to(device=image.device)
        return eng.run(image, noise).clone()

    @torch.no_grad()
    def vae_decode(self, z: torch.Tensor, tiled: bool = False, tile_size: int = -1) -> torch.Tensor:
        _require_gpu(z, "vae_decode")
        B, _, h, w = z.shape
        if tiled and tile_size <= 0:
            raise ValueError("vae_decode(tiled=True) needs a positive tile_size (latent pixels)")
        return self.vae_engine("decode", B, h, w, tile_size if tiled else 0).run(z).clone()

    def prepare_condition(self, clean: torch.Tensor, prompt: List[str]) -> Dict[str, torch.Tensor]:
        if prompt is None:
            prompt = [""] * clean.size(0)
        self.clip.compute_dtype = self.compute_dtype
        return dict(c_txt=self.clip.encode(prompt), c_img=self.vae_encode(clean * 2 - 1, sample=False))

    # -- the denoiser (reference model/cldm.py:166-194) -------------------------------------------
    @torch.no_grad()
    def forward(self, x_noisy: torch.Tensor, t: torch.Tensor, cond: Dict[str, torch.Tensor], woSD: bool = False) -> torch.Tensor:
        if woSD:
            raise NotImplementedError("woSD/tail_block is dead code in the reference and is not built")
        _require_gpu(x_noisy, "ControlLDM.forward")
        c_txt, c_img = cond["c_txt"], cond["c_img"]
        B, _, h, w = x_noisy.shape
        eng = self.cldm_engine(B, h, w, c_txt.shape[1])
        return eng.step(x_noisy, t, c_txt, c_img).clone()
