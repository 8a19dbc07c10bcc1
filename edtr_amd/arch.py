"""Architecture description of the ControlLDM networks, derived from the constructor kwargs
(reference configs/det/demo.yaml:24-86).  One flat list of layer descriptors per network drives
(a) the parameter tree that gives strict state-dict compatibility with the reference
(edtr_amd/model/params.py) and (b) the kernel-program emitters (edtr_amd/nets.py).

Reference constructors being described: model/unet.py:391-685 (UNetModel), model/controlnet.py:46-258
(ControlNet), model/vae.py:326-419 (Encoder), :449-525 (Decoder), :681-704 (AutoencoderKL).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

Shape = Tuple[int, ...]


@dataclass
class Layer:
    kind: str                 # conv | res | attn | down | up
    prefix: str               # state-dict prefix, ending with "."
    cin: int
    cout: int
    heads: int = 0


@dataclass
class UNetArch:
    model_channels: int
    context_dim: int
    in_channels: int
    out_channels: int         # 0 for the ControlNet (no output head)
    hint_channels: int        # 0 for the UNet
    input_blocks: List[List[Layer]] = field(default_factory=list)
    middle: List[Layer] = field(default_factory=list)
    output_blocks: List[List[Layer]] = field(default_factory=list)   # UNet only
    zero_convs: List[Tuple[str, int]] = field(default_factory=list)  # ControlNet only: (prefix, channels)
    skip_channels: List[int] = field(default_factory=list)           # channels of each input-block output

    def res_layers(self) -> List[Layer]:
        out = []
        for blk in self.input_blocks + [self.middle] + self.output_blocks:
            out += [l for l in blk if l.kind == "res"]
        return out

    def attn_layers(self) -> List[Layer]:
        out = []
        for blk in self.input_blocks + [self.middle] + self.output_blocks:
            out += [l for l in blk if l.kind == "attn"]
        return out


def control_shapes(a: UNetArch, h: int, w: int) -> List[Tuple[int, int, int]]:
    """(channels, height, width) of the 13 control tensors / skip activations for a latent of h x w: one per input block,
    then the middle block (reference model/controlnet.py:263-277)."""
    out = []
    for blk, ch in zip(a.input_blocks, a.skip_channels):
        if any(l.kind == "down" for l in blk):
            h, w = (h + 1) // 2, (w + 1) // 2
        out.append((ch, h, w))
    out.append((a.skip_channels[-1], h, w))
    return out


def _check_supported(cfg: dict) -> None:
    if not cfg.get("use_spatial_transformer", False) or not cfg.get("use_linear_in_transformer", False):
        raise NotImplementedError("only the SD-2.x layout (spatial transformer with linear projections) is built")
    if cfg.get("transformer_depth", 1) != 1 or cfg.get("num_head_channels", -1) != 64:
        raise NotImplementedError("the attention kernel is specialised for transformer_depth=1, head width 64")
    for flag in ("use_scale_shift_norm", "resblock_updown", "use_fp16"):
        if cfg.get(flag, False):
            raise NotImplementedError(f"{flag}=True is not used by any EDTR config and is not built")
    if cfg.get("num_classes") is not None or cfg.get("n_embed") is not None:
        raise NotImplementedError("class-conditional / codebook heads are not on the EDTR path")
    if cfg.get("dims", 2) != 2 or not cfg.get("conv_resample", True):
        raise NotImplementedError("only 2-D, conv-resampled networks are built")


def unet_arch(cfg: dict, controlnet: bool = False) -> UNetArch:
    _check_supported(cfg)
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nres = cfg["num_res_blocks"]
    nres = [nres] * len(mult) if isinstance(nres, int) else list(nres)
    attn_res = list(cfg["attention_resolutions"])
    hd = cfg["num_head_channels"]
    hint = cfg.get("hint_channels", 0) if controlnet else 0
    a = UNetArch(model_channels=mc, context_dim=cfg["context_dim"], in_channels=cfg["in_channels"],
                 out_channels=0 if controlnet else cfg["out_channels"], hint_channels=hint)
    a.input_blocks.append([Layer("conv", "input_blocks.0.0.", cfg["in_channels"] + hint, mc)])
    a.skip_channels.append(mc)
    if controlnet:
        a.zero_convs.append(("zero_convs.0.0.", mc))
    ch, ds, idx = mc, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nres[level]):
            layers = [Layer("res", f"input_blocks.{idx}.0.", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(Layer("attn", f"input_blocks.{idx}.1.", ch, ch, ch // hd))
            a.input_blocks.append(layers)
            a.skip_channels.append(ch)
            if controlnet:
                a.zero_convs.append((f"zero_convs.{idx}.0.", ch))
            idx += 1
        if level != len(mult) - 1:
            a.input_blocks.append([Layer("down", f"input_blocks.{idx}.0.", ch, ch)])
            a.skip_channels.append(ch)
            if controlnet:
                a.zero_convs.append((f"zero_convs.{idx}.0.", ch))
            idx += 1
            ds *= 2
    a.middle = [Layer("res", "middle_block.0.", ch, ch), Layer("attn", "middle_block.1.", ch, ch, ch // hd),
                Layer("res", "middle_block.2.", ch, ch)]
    if controlnet:
        a.zero_convs.append(("middle_block_out.0.", ch))
        return a
    skips = list(a.skip_channels)
    idx = 0
    for level in reversed(range(len(mult))):
        for i in range(nres[level] + 1):
            ich = skips.pop()
            layers = [Layer("res", f"output_blocks.{idx}.0.", ch + ich, mc * mult[level])]
            ch = mc * mult[level]
            j = 1
            if ds in attn_res:
                layers.append(Layer("attn", f"output_blocks.{idx}.{j}.", ch, ch, ch // hd))
                j += 1
            if level and i == nres[level]:
                layers.append(Layer("up", f"output_blocks.{idx}.{j}.", ch, ch))
                ds //= 2
            a.output_blocks.append(layers)
            idx += 1
    return a


def unet_param_spec(a: UNetArch) -> List[Tuple[str, Shape]]:
    """(key, shape) in the reference's state-dict order."""
    mc, ted = a.model_channels, a.model_channels * 4
    spec: List[Tuple[str, Shape]] = [
        ("time_embed.0.weight", (ted, mc)), ("time_embed.0.bias", (ted,)),
        ("time_embed.2.weight", (ted, ted)), ("time_embed.2.bias", (ted,)),
    ]

    def add_layer(l: Layer):
        p = l.prefix
        if l.kind == "conv":
            spec.extend([(p + "weight", (l.cout, l.cin, 3, 3)), (p + "bias", (l.cout,))])
        elif l.kind == "down":
            spec.extend([(p + "op.weight", (l.cout, l.cin, 3, 3)), (p + "op.bias", (l.cout,))])
        elif l.kind == "up":
            spec.extend([(p + "conv.weight", (l.cout, l.cin, 3, 3)), (p + "conv.bias", (l.cout,))])
        elif l.kind == "res":
            spec.extend([
                (p + "in_layers.0.weight", (l.cin,)), (p + "in_layers.0.bias", (l.cin,)),
                (p + "in_layers.2.weight", (l.cout, l.cin, 3, 3)), (p + "in_layers.2.bias", (l.cout,)),
                (p + "emb_layers.1.weight", (l.cout, ted)), (p + "emb_layers.1.bias", (l.cout,)),
                (p + "out_layers.0.weight", (l.cout,)), (p + "out_layers.0.bias", (l.cout,)),
                (p + "out_layers.3.weight", (l.cout, l.cout, 3, 3)), (p + "out_layers.3.bias", (l.cout,)),
            ])
            if l.cin != l.cout:
                spec.extend([(p + "skip_connection.weight", (l.cout, l.cin, 1, 1)), (p + "skip_connection.bias", (l.cout,))])
        elif l.kind == "attn":
            c, ctx = l.cout, a.context_dim
            t = p + "transformer_blocks.0."
            spec.extend([
                (p + "norm.weight", (c,)), (p + "norm.bias", (c,)),
                (p + "proj_in.weight", (c, c)), (p + "proj_in.bias", (c,)),
                (t + "attn1.to_q.weight", (c, c)), (t + "attn1.to_k.weight", (c, c)), (t + "attn1.to_v.weight", (c, c)),
                (t + "attn1.to_out.0.weight", (c, c)), (t + "attn1.to_out.0.bias", (c,)),
                (t + "ff.net.0.proj.weight", (8 * c, c)), (t + "ff.net.0.proj.bias", (8 * c,)),
                (t + "ff.net.2.weight", (c, 4 * c)), (t + "ff.net.2.bias", (c,)),
                (t + "attn2.to_q.weight", (c, c)), (t + "attn2.to_k.weight", (c, ctx)), (t + "attn2.to_v.weight", (c, ctx)),
                (t + "attn2.to_out.0.weight", (c, c)), (t + "attn2.to_out.0.bias", (c,)),
                (t + "norm1.weight", (c,)), (t + "norm1.bias", (c,)),
                (t + "norm2.weight", (c,)), (t + "norm2.bias", (c,)),
                (t + "norm3.weight", (c,)), (t + "norm3.bias", (c,)),
                (p + "proj_out.weight", (c, c)), (p + "proj_out.bias", (c,)),
            ])

    for blk in a.input_blocks:
        for l in blk:
            add_layer(l)
    if a.zero_convs:   # ControlNet: input_blocks, zero_convs, middle_block, middle_block_out
        for p, c in a.zero_convs[:-1]:
            spec.extend([(p + "weight", (c, c, 1, 1)), (p + "bias", (c,))])
    for l in a.middle:
        add_layer(l)
    if a.zero_convs:
        p, c = a.zero_convs[-1]
        spec.extend([(p + "weight", (c, c, 1, 1)), (p + "bias", (c,))])
    for blk in a.output_blocks:
        for l in blk:
            add_layer(l)
    if a.out_channels:
        spec.extend([("out.0.weight", (mc,)), ("out.0.bias", (mc,)),
                     ("out.2.weight", (a.out_channels, mc, 3, 3)), ("out.2.bias", (a.out_channels,))])
    return spec


# ----------------------------------------------------------------------------------------------
# VAE
# ----------------------------------------------------------------------------------------------
@dataclass
class VaeLayer:
    kind: str       # conv | res | attn | down | up | norm_out
    prefix: str
    cin: int
    cout: int


def vae_encoder_arch(dd: dict) -> List[VaeLayer]:
    ch, mult, nres = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    if dd.get("attn_resolutions"):
        raise NotImplementedError("per-level VAE attention is not used by the SD VAE and is not built")
    in_mult = [1] + mult
    L = [VaeLayer("conv", "conv_in.", dd["in_channels"], ch)]
    cin = ch
    for lvl in range(len(mult)):
        cin, cout = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(nres):
            L.append(VaeLayer("res", f"down.{lvl}.block.{b}.", cin, cout))
            cin = cout
        if lvl != len(mult) - 1:
            L.append(VaeLayer("down", f"down.{lvl}.downsample.conv.", cin, cin))
    L += [VaeLayer("res", "mid.block_1.", cin, cin), VaeLayer("attn", "mid.attn_1.", cin, cin),
          VaeLayer("res", "mid.block_2.", cin, cin), VaeLayer("norm_out", "norm_out.", cin, cin),
          VaeLayer("conv", "conv_out.", cin, 2 * dd["z_channels"] if dd.get("double_z", True) else dd["z_channels"])]
    return L


def vae_decoder_arch(dd: dict) -> List[VaeLayer]:
    ch, mult, nres = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    cin = ch * mult[-1]
    L = [VaeLayer("conv", "conv_in.", dd["z_channels"], cin),
         VaeLayer("res", "mid.block_1.", cin, cin), VaeLayer("attn", "mid.attn_1.", cin, cin),
         VaeLayer("res", "mid.block_2.", cin, cin)]
    for lvl in reversed(range(len(mult))):
        cout = ch * mult[lvl]
        for b in range(nres + 1):
            L.append(VaeLayer("res", f"up.{lvl}.block.{b}.", cin, cout))
            cin = cout
        if lvl != 0:
            L.append(VaeLayer("up", f"up.{lvl}.upsample.conv.", cin, cin))
    L += [VaeLayer("norm_out", "norm_out.", cin, cin), VaeLayer("conv", "conv_out.", cin, dd["out_ch"])]
    return L


def _vae_layer_spec(l: VaeLayer) -> List[Tuple[str, Shape]]:
    p = l.prefix
    if l.kind in ("conv", "down", "up"):
        return [(p + "weight", (l.cout, l.cin, 3, 3)), (p + "bias", (l.cout,))]
    if l.kind == "norm_out":
        return [(p + "weight", (l.cin,)), (p + "bias", (l.cin,))]
    if l.kind == "res":
        s = [(p + "norm1.weight", (l.cin,)), (p + "norm1.bias", (l.cin,)),
             (p + "conv1.weight", (l.cout, l.cin, 3, 3)), (p + "conv1.bias", (l.cout,)),
             (p + "norm2.weight", (l.cout,)), (p + "norm2.bias", (l.cout,)),
             (p + "conv2.weight", (l.cout, l.cout, 3, 3)), (p + "conv2.bias", (l.cout,))]
        if l.cin != l.cout:
            s += [(p + "nin_shortcut.weight", (l.cout, l.cin, 1, 1)), (p + "nin_shortcut.bias", (l.cout,))]
        return s
    if l.kind == "attn":
        c = l.cin
        s = [(p + "norm.weight", (c,)), (p + "norm.bias", (c,))]
        for n in ("q", "k", "v", "proj_out"):
            s += [(p + f"{n}.weight", (c, c, 1, 1)), (p + f"{n}.bias", (c,))]
        return s
    raise ValueError(l.kind)


def vae_param_spec(vae_cfg: dict) -> List[Tuple[str, Shape]]:
    """AutoencoderKL state-dict keys in reference order (decoder's `up` ModuleList is stored lowest level first,
    model/vae.py:517 `self.up.insert(0, up)`)."""
    dd = vae_cfg["ddconfig"]
    spec: List[Tuple[str, Shape]] = []
    enc = vae_encoder_arch(dd)
    # reference registration order: conv_in, down.*, mid.*, norm_out, conv_out
    for l in enc:
        spec += [("encoder." + k, s) for k, s in _vae_layer_spec(l)]
    dec = vae_decoder_arch(dd)
    head = [l for l in dec if not l.prefix.startswith("up.") and l.kind != "norm_out" and l.prefix != "conv_out."]
    ups = [l for l in dec if l.prefix.startswith("up.")]
    tail = [l for l in dec if l.kind == "norm_out" or l.prefix == "conv_out."]
    ups_sorted = sorted(ups, key=lambda l: int(l.prefix.split(".")[1]))  # stable: keeps block / upsample order per level
    for l in head + ups_sorted + tail:
        spec += [("decoder." + k, s) for k, s in _vae_layer_spec(l)]
    e = vae_cfg["embed_dim"]
    z = dd["z_channels"]
    spec += [("quant_conv.weight", (2 * e, 2 * z, 1, 1)), ("quant_conv.bias", (2 * e,)),
             ("post_quant_conv.weight", (z, e, 1, 1)), ("post_quant_conv.bias", (z,))]
    return spec
