"""Kernel-program emitters for the networks on the hot path.  Each function appends libedtr_hip launches to
the emitter's Program; no arithmetic happens here.  Structure follows edtr_amd/arch.py (derived from the
reference constructors); per-function citations give the reference forward being restated.

Fusions relative to the reference's op-by-op execution:
  * NHWC 16-bit activations end to end: no `b c h w <-> b (h w) c` rearranges (model/attention.py:292,299)
  * bias, time-embedding add, residual add, GEGLU gate, SiLU are GEMM/conv epilogues
  * nearest-2x upsample, the VAE's asymmetric pad and stride-2 are folded into the conv's gather
  * torch.cat([h, skip + control]) is never materialised: producers write column slices of one buffer
  * all ResBlock emb_layers of a net are ONE GEMM; all cross-attention K / V^T projections of a net are
    two GEMMs, computed once per prompt instead of once per denoise step
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .arch import Layer, UNetArch, VaeLayer
from .engine import Act, Emitter, LNRef, LNReg
from .ops import round_up


# ----------------------------------------------------------------------------------------------
# conditioning that does not depend on the latent: time embedding rows, cross-attention K / V^T
# ----------------------------------------------------------------------------------------------
def emit_time_rows(em: Emitter, P: str, a: UNetArch, t_dev: torch.Tensor, B: int) -> Tuple[torch.Tensor, Dict[str, int]]:
    """timestep_embedding -> time_embed MLP -> SiLU -> every ResBlock's emb_layers Linear, as 4 launches.
    Returns the fp32 [B, sum(Cout)] table and the column offset of each ResBlock prefix.
    Reference: model/util.py:98-118, model/unet.py:475-480 / model/controlnet.py:128-133, model/unet.py:212."""
    mc, ted = a.model_channels, a.model_channels * 4
    temb = em.new(B, mc)
    em.prog.add(ops.make_timestep_embedding(dtype=em.io, t=t_dev, B=B, dim=mc, out=temb, ld=mc))
    w0, b0 = em.store.linear([P + "time_embed.0.weight"], [P + "time_embed.0.bias"])
    h = em.gemm(temb, w0, B, ted, mc, bias=b0, act=L.ACT_SILU, name="time_embed.0")
    w2, b2 = em.store.linear([P + "time_embed.2.weight"], [P + "time_embed.2.bias"])
    semb = em.gemm(h, w2, B, ted, ted, bias=b2, act=L.ACT_SILU, name="time_embed.2")   # SiLU(emb): all consumers want it
    res = a.res_layers()
    wall, ball = em.store.linear([P + l.prefix + "emb_layers.1.weight" for l in res],
                                 [P + l.prefix + "emb_layers.1.bias" for l in res])
    total = wall.shape[0]
    table = em.gemm(semb, wall, B, total, ted, bias=ball, out_f32=True, name="emb_layers(all)")
    em.free(temb, h, semb)
    offs, o = {}, 0
    for l in res:
        offs[l.prefix] = o
        o += l.cout
    return table, offs


ATTN_PRESCALE = 1.4426950408889634 / math.sqrt(64.0)      # log2(e) / sqrt(head width)


class ContextKV:
    """Per-net cross-attention keys / transposed values for every transformer layer, from c_txt."""

    def __init__(self):
        self.k_all = None      # [B*Nctx, sumC]   (split attention, high mode: [B*Nctx, 2*sumC] = [hi | lo])
        self.vt_all = None     # [B*sumC, ldv]    (split: [B*sumC, 2*ldv] = [hi | lo])
        self.sumC = 0
        self.ldv = 0
        self.Nctx = 0
        self.split = 0         # Emitter.attn_split the context was prepared for
        self.offs: Dict[str, int] = {}


def emit_context_kv(em: Emitter, P: str, a: UNetArch, ctx16: torch.Tensor, B: int, Nctx: int) -> ContextKV:
    """k = to_k(context), v = to_v(context) of every attn2 (model/attention.py:171-174), batched over layers."""
    layers = a.attn_layers()
    kv = ContextKV()
    t = "transformer_blocks.0.attn2."
    wk, _ = em.store.linear([P + l.prefix + t + "to_k.weight" for l in layers])
    wv, _ = em.store.linear([P + l.prefix + t + "to_v.weight" for l in layers])
    kv.sumC, kv.Nctx = wk.shape[0], Nctx
    kv.split = em.attn_split
    kv.k_all = em.gemm(ctx16, wk, B * Nctx, kv.sumC, a.context_dim, name="ctx_k(all)", out16=kv.split < 1)
    kv.vt_all, kv.ldv = em.vt_gemm(wv, ctx16, B=B, Ntok=Nctx, Cin=a.context_dim, name="ctx_vT(all)", out16=kv.split < 2)
    # fp32 projections (high mode: bf16 MFMA type; split attention in any fp32-stream mode): the attention operands are fp16, cut
    # once per prompt — hi + lo pairs ([hi | lo] columns) where the attention is split, one part otherwise
    if kv.k_all.dtype == torch.float32:
        k32 = kv.k_all
        kv.k_all = em.split16(k32, B * Nctx, kv.sumC) if kv.split >= 1 else em.to16(k32, B * Nctx, kv.sumC)
        em.free(k32)
    if kv.vt_all.dtype == torch.float32:
        v32 = kv.vt_all
        kv.vt_all = em.split16(v32, B * kv.sumC, kv.ldv) if kv.split >= 2 else em.to16(v32, B * kv.sumC, kv.ldv)
        em.free(v32)
    o = 0
    for l in layers:
        kv.offs[l.prefix] = o
        o += l.cout
    return kv


# ----------------------------------------------------------------------------------------------
# UNet / ControlNet blocks
# ----------------------------------------------------------------------------------------------
def emit_resblock(em: Emitter, P: str, l: Layer, x: Act, table: torch.Tensor, offs: Dict[str, int], out=None, mirror=False, gnp_into=None) -> Act:
    """model/unet.py:203-223: GN-SiLU-conv (+bias +emb row) ; GN-SiLU-conv (+bias) + skip(x)."""
    p = P + l.prefix
    # (conv_n: where the halo tile takes the convolution, the GroupNorm is applied inside its operand staging — no apply launch)
    n1 = em.group_norm(x, p + "in_layers.0.", 1e-5, True, feeds=("res.conv1",), conv_n=l.cout)
    h = em.conv(n1, p + "in_layers.2.", rowvec=table[:, offs[l.prefix]:], name="res.conv1", stats=True, feeds=("res.conv2",))
    em.free(n1)
    n2 = em.group_norm(h, p + "out_layers.0.", 1e-5, True, feeds=("res.conv2",), conv_n=l.cout, take=True)
    em.free(h)
    if l.cin != l.cout:
        skip = em.conv(x, p + "skip_connection.", taps=1, name="res.skip1x1").t
    else:
        skip = x.t
    y = em.conv(n2, p + "out_layers.3.", residual=skip, out=out, name="res.conv2", stats=out is None, mirror=mirror, gnp_into=gnp_into)
    into_done = em.gnp_into_done
    em.free(n2)
    if l.cin != l.cout:
        em.free(skip)
    em.gnp_into_done = into_done
    return y


def emit_attention_core(em: Emitter, p: str, x, B: int, N: int, C: int, heads: int,
                        ctx: Optional[Tuple[torch.Tensor, torch.Tensor, int, int, int, int]], residual: torch.Tensor,
                        row_stats: bool = False) -> torch.Tensor:
    """One attention layer on tokens x [B*N, C] (already layer-normed, or an LNRef = the LayerNorm folded into the
    projection): projections, fused attention, output projection with bias and residual.  ``row_stats``: the output
    projection also writes the per-row statistics of its result (em.last_row_stats) for the next folded LayerNorm.
    model/attention.py:176-203."""
    fold = x.prefix if isinstance(x, LNRef) else None
    # softmax scale (1/sqrt(64)) and the exp -> exp2 factor log2(e) are folded into the projections' fp32 epilogues (q.k
    # products arrive "prescaled", include/edtr_hip.h q_prescaled): sqrt(c) on both halves of the fused [Q;K] projection,
    # c on the cross-attention's Q (its K comes from the per-prompt context program) — no extra rounding, one multiply
    # less per score in the attention kernel
    c = ATTN_PRESCALE
    if ctx is None and em.fused_qkv_ok(N, C):
        # self attention: ONE launch for [Wq; Wk; Wv] — q / k row-major (scaled), v^T written transposed by the epilogue
        names = [p + "to_q.weight", p + "to_k.weight", p + "to_v.weight"]
        if isinstance(x, LNReg):    # norm1 + [Wq; Wk; Wv] as ONE launch on the raw rows (edtr_lin320, transposed V part)
            qk, vt, ldv = em.qkv_lin320(x.x, names, B=B, N=N, C=C, ln_prefix=x.prefix, alpha=math.sqrt(c))
        elif fold:
            wqkv, _, c1, c2 = em.store.ln_fold("linear", names, None, fold)
            qk, vt, ldv = em.qkv_gemm(x, wqkv, B=B, N=N, C=C, alpha=math.sqrt(c), ln_vec=(c1, c2))
        else:
            wqkv, _ = em.store.linear(names)
            qk, vt, ldv = em.qkv_gemm(x, wqkv, B=B, N=N, C=C, alpha=math.sqrt(c))
        o = em.flash(qk[:, :C], qk[:, C:], vt, B=B, H=heads, Nq=N, Nk=N, k_bs=N * qk.stride(0), vt_bs=C * ldv, vt_ld=ldv,
                     prescaled=True)
        em.free(qk, vt)
    elif ctx is None:   # (high mode / odd shapes) fused [Wq;Wk] projection, V^T via the operand-swapped GEMM
        wqk, _ = em.store.linear([p + "to_q.weight", p + "to_k.weight"])
        qk = em.gemm(x, wqk, B * N, 2 * C, C, alpha=math.sqrt(c), name="attn1.qk", out16=em.attn_split < 1)
        wv, _ = em.store.linear([p + "to_v.weight"])
        vt, ldv = em.vt_gemm(wv, x, B=B, Ntok=N, Cin=C, name="attn1.vT", out16=em.attn_split < 2)
        o = em.flash(qk[:, :C], qk[:, C:], vt, B=B, H=heads, Nq=N, Nk=N, k_bs=N * qk.stride(0), vt_bs=C * ldv, vt_ld=ldv,
                     prescaled=True)
        em.free(qk, vt)
    else:
        k, vt, k_bs, vt_bs, ldv, nctx, k_lo, vt_lo = ctx
        if isinstance(x, LNReg):    # norm2 + to_q as ONE launch on the raw rows (edtr_lin320: the rows are normalised in registers)
            q = em.lin320(x.x, B * N, C, [p + "to_q.weight"], None, ln_prefix=x.prefix, alpha=c, name="attn2.q")
        elif fold:
            wq, _, c1, c2 = em.store.ln_fold("linear", [p + "to_q.weight"], None, fold)
            q = em.gemm(x, wq, B * N, C, C, alpha=c, name="attn2.q", out16=True, ln_vec=(c1, c2))
        else:
            wq, _ = em.store.linear([p + "to_q.weight"])
            q = em.gemm(x, wq, B * N, C, C, alpha=c, name="attn2.q", out16=em.attn_split < 1)
        o = em.flash(q, k, vt, B=B, H=heads, Nq=N, Nk=nctx, k_bs=k_bs, vt_bs=vt_bs, vt_ld=ldv, prescaled=True, k_lo=k_lo, vt_lo=vt_lo)
        em.free(q)
    if not row_stats and em.lin320_ok(B * N, C, C) and o.dtype == residual.dtype:
        y = em.lin320(o, B * N, C, [p + "to_out.0.weight"], [p + "to_out.0.bias"], residual=residual, name="attn.out")
    else:
        wo, bo = em.store.linear([p + "to_out.0.weight"], [p + "to_out.0.bias"])
        y = em.gemm(o, wo, B * N, C, C, bias=bo, residual=residual, name="attn.out", row_stats=row_stats)
    em.free(o)
    return y


def emit_spatial_transformer(em: Emitter, P: str, l: Layer, x: Act, kv: ContextKV, out=None, mirror=False, gnp_into=None) -> Act:
    """model/attention.py:283-302 + :230-234 + :20-47 (use_linear, depth 1, gated FF)."""
    p = P + l.prefix
    B, N, C = x.B, x.H * x.W, x.C
    rows = B * N
    fold = em.ln_fold_ok(C, B)
    fold1 = fold and em.fused_qkv_ok(N, C)        # (the operand-swapped V^T product would need per-COLUMN scalars)
    lin = em.lin320_ok(rows, C, C)           # the K = 320 projections of the block as row-resident launches (edtr_lin320)
    # (proj_in on edtr_lin320 applies the GroupNorm to the rows it holds: a (scale, shift) table launch instead of the apply launch)
    n = em.group_norm(x, p + "norm.", 1e-6, False, feeds=("st.proj_in",), lin_ok=lin and not fold1)
    wi, bi = em.store.linear([p + "proj_in.weight"], [p + "proj_in.bias"])
    # The three LayerNorms of the block are not launched in the fast modes: the GEMM that writes their input also writes
    # per-row statistics, the GEMMs that read them run on the raw rows with gamma folded into the weights (Emitter.layer_norm).
    if lin and not fold1:
        t = em.lin320(n.t, rows, C, [p + "proj_in.weight"], [p + "proj_in.bias"], gn_table=n.gn_in, rows_per_image=N, name="st.proj_in")
    else:
        t = em.gemm(n.t, wi, rows, C, C, bias=bi, name="st.proj_in", row_stats=fold1)
    st = em.last_row_stats
    em.free(n)
    tb = p + "transformer_blocks.0."
    if (st is None and em.fused_qkv_ok(N, C) and em.lin320_ok(rows, 3 * C, C, ln=True) and N % 32 == 0 and em.attn_dtype == em.dtype
            and os.environ.get("EDTR_LIN320_QKV", "1") != "0"):
        l1 = LNReg(t, C, tb + "norm1.")
    else:
        l1 = em.layer_norm(t, rows, C, tb + "norm1.", feeds=("attn1.qkv",) if em.fused_qkv_ok(N, C) else ("attn1.qk", "attn1.vT"), stats=st)
    t1 = emit_attention_core(em, tb + "attn1.", l1, B, N, C, l.heads, None, t, row_stats=fold)
    st = em.last_row_stats
    em.free(l1, t)
    l2 = (LNReg(t1, C, tb + "norm2.") if (st is None and em.lin320_ok(rows, C, C, ln=True))
          else em.layer_norm(t1, rows, C, tb + "norm2.", feeds=("attn2.q",), stats=st))
    off = kv.offs[l.prefix]
    k_view = kv.k_all[:, off:off + C]
    k_lo = kv.k_all[:, kv.sumC + off:kv.sumC + off + C] if kv.split >= 1 else None
    ldv_all = kv.vt_all.stride(0)          # (2 * ldv when the values are hi + lo pairs)
    vt3 = kv.vt_all.view(B, kv.sumC, ldv_all)
    vt_view = vt3[:, off:off + C, :kv.ldv]
    vt_lo = vt3[:, off:off + C, kv.ldv:] if kv.split >= 2 else None
    ctx = (k_view, vt_view, kv.Nctx * kv.k_all.stride(0), kv.sumC * ldv_all, ldv_all, kv.Nctx, k_lo, vt_lo)
    fused_ff = em.ffn_ok(rows, C)           # (edtr_ffn takes the row statistics from the rows it holds: no row_stats from attn.out)
    t2 = emit_attention_core(em, tb + "attn2.", l2, B, N, C, l.heads, ctx, t1, row_stats=fold and not fused_ff)
    st = em.last_row_stats
    em.free(l2, t1)
    if fused_ff:
        # norm3 + ff.geglu + ff.out + residual as ONE launch on the raw rows (edtr_ffn: the (rows, 4 C) hidden tensor stays on chip)
        t3 = em.ffn(t2, rows, C, tb)
        em.free(t2)
    else:
        l3 = em.layer_norm(t2, rows, C, tb + "norm3.", feeds=("ff.geglu",), stats=st)
        if isinstance(l3, LNRef):
            wg, bg, c1, c2 = em.store.ln_fold("geglu", [tb + "ff.net.0.proj.weight"], [tb + "ff.net.0.proj.bias"], l3.prefix)
            g = em.gemm(l3, wg, rows, 8 * C, C, bias=bg, act=L.ACT_GEGLU, name="ff.geglu", feeds="ff.out", ln_vec=(c1, c2))
        else:
            wg, bg = em.store.geglu(tb + "ff.net.0.proj.weight", tb + "ff.net.0.proj.bias")
            g = em.gemm(l3, wg, rows, 8 * C, C, bias=bg, act=L.ACT_GEGLU, name="ff.geglu", feeds="ff.out")
        em.free(l3)
        wf, bf = em.store.linear([tb + "ff.net.2.weight"], [tb + "ff.net.2.bias"])
        t3 = em.gemm(g, wf, rows, C, 4 * C, bias=bf, residual=t2, name="ff.out", feeds="st.proj_out" if em.branch16 else None)
        em.free(g, t2)
    wo, bo = em.store.linear([p + "proj_out.weight"], [p + "proj_out.bias"])
    y = em.gemm(t3, wo, rows, C, C, bias=bo, residual=x.t, out=out, name="st.proj_out", stats_hw=N if (out is None or gnp_into is not None) else 0,
                mirror=mirror, gnp_into=gnp_into if out is not None else None)
    into_done = em.gnp_into_done
    em.free(t3)
    em.gnp_into_done = into_done
    return Act(y, x.B, x.H, x.W, C, em.last_gnp, gn_slot=em.last_gn_slot)


def mirror_for(em: Emitter, nxt: Optional[Layer]) -> bool:
    """Does the layer that consumes a stream tensor next read it as a one-part GEMM operand (mixed mode: the producer then writes
    the fp16 mirror, Emitter.want_mirror)?  Down / upsample convolutions read their input raw; a ResBlock reads it raw through its
    1x1 skip convolution when the channel count changes; everything else reads it through a normalisation."""
    if nxt is None:
        return False
    if nxt.kind == "down":
        return em.want_mirror("downsample")
    if nxt.kind == "up":
        return em.want_mirror("upsample.conv")
    if nxt.kind == "res" and nxt.cin != nxt.cout:
        return em.want_mirror("res.skip1x1")
    return False


def emit_block(em: Emitter, P: str, layers: List[Layer], x: Act, table, offs, kv, out=None, keep_input=True, mirror_out=False,
               gnp_into=None) -> Act:
    """TimestepEmbedSequential dispatch (model/unet.py:40-48).  ``out`` (a 2-D view) receives the LAST layer's
    result; the block input is freed unless ``keep_input``.  ``mirror_out``: a consumer of the block's result reads it as a
    one-part operand (mirror_for / the ControlNet's zero convolutions)."""
    h = x
    for i, l in enumerate(layers):
        last = i == len(layers) - 1
        tgt = out if last else None
        into = gnp_into if last else None          # (the block's LAST layer fills the caller's slice: it can write that slice's GroupNorm partials)
        mir = mirror_out if last else mirror_for(em, layers[i + 1])
        if l.kind == "conv":
            y = em.conv(h, P + l.prefix, out=tgt, name="conv_in", stats=tgt is None, mirror=mir)
        elif l.kind == "res":
            y = emit_resblock(em, P, l, h, table, offs, out=tgt, mirror=mir, gnp_into=into)
        elif l.kind == "attn":
            y = emit_spatial_transformer(em, P, l, h, kv, out=tgt, mirror=mir, gnp_into=into)
        elif l.kind == "down":
            y = em.conv(h, P + l.prefix + "op.", stride=2, out=tgt, name="downsample", stats=tgt is None, mirror=mir)  # unet.py:99-108
        elif l.kind == "up":
            y = em.conv(h, P + l.prefix + "conv.", ups=True, out=tgt, name="upsample.conv", stats=tgt is None, mirror=mir, gnp_into=into)  # unet.py:70-79
        else:
            raise ValueError(l.kind)
        into_done = em.gnp_into_done if last else False
        if h is not x or not keep_input:
            em.free(h)
        h = y
    em.gnp_into_done = bool(gnp_into is not None and into_done)
    return h


def emit_controlnet(em: Emitter, P: str, a: UNetArch, x8: Act, table, offs, kv, scales: List[float]) -> List[Act]:
    """model/controlnet.py:263-277: 12 encoder blocks + middle block, each tapped by a 1x1 conv.  ``x8`` is
    cat(x, hint) already in NHWC.  The control scale (model/cldm.py:189) is the tap's alpha."""
    outs: List[Act] = []
    h = x8
    zmir = em.want_mirror("zero_conv")
    for i, layers in enumerate(a.input_blocks):
        nxt = a.input_blocks[i + 1][0] if i + 1 < len(a.input_blocks) else a.middle[0]
        y = emit_block(em, P, layers, h, table, offs, kv, keep_input=(i == 0), mirror_out=zmir or mirror_for(em, nxt))
        outs.append(em.conv(y, P + a.zero_convs[i][0], taps=1, alpha=scales[i], name="zero_conv"))
        h = y
    y = emit_block(em, P, a.middle, h, table, offs, kv, keep_input=False, mirror_out=zmir)
    outs.append(em.conv(y, P + a.zero_convs[-1][0], taps=1, alpha=scales[12], name="zero_conv"))
    em.free(y)
    return outs


def emit_unet(em: Emitter, P: str, a: UNetArch, x8: Act, table, offs, kv, control: Optional[List[Act]],
              before_control=None) -> torch.Tensor:
    """model/controlnet.py:20-41 (ControlledUnetModel.forward) + output head model/unet.py:675-679.
    Returns the fp32 NHWC eps [B*h*w, 8] (first out_channels columns valid)."""
    control = list(control) if control is not None else None
    hs: List[Act] = []
    h = x8
    for i, layers in enumerate(a.input_blocks):
        nxt = a.input_blocks[i + 1][0] if i + 1 < len(a.input_blocks) else a.middle[0]
        h = emit_block(em, P, layers, h, table, offs, kv, keep_input=True, mirror_out=mirror_for(em, nxt))
        hs.append(h)
    mid = emit_block(em, P, a.middle, h, table, offs, kv, keep_input=True)
    if before_control is not None:
        before_control()      # e.g. Program.join(): everything above is independent of the ControlNet

    # first decoder input: cat([mid + control_mid, hs[-1] + control[-2]])
    skip = hs.pop()
    # the decoder's concat buffers: every output block starts with a ResBlock whose 1x1 skip convolution reads the concat raw

    def concat_slots(rows: int, hw: int, ctot: int):
        """The GroupNorm of a concatenation needs no pass over it (edtr_gn_stats) when BOTH halves' producers write their columns'
        partials into one buffer of slots (round 6: edtr_igemm gn_ld, edtr_add_stats): (buffer, rows per slot) or (None, 0).  Fast
        modes; EDTR_GN_CONCAT_STATS=0 keeps the statistics launch (A/B runs)."""
        slot = ops.gn_slot_rows(hw)
        if em.hp or em.invariant or not slot or rows % slot or os.environ.get("EDTR_GN_CONCAT_STATS", "1") == "0":
            return None, 0
        return em.arena.alloc((rows // slot, ctot, 2), torch.float32), slot

    cat = em.new_stream(skip.rows, mid.C + skip.C, mirror=em.want_mirror("res.skip1x1"))
    cg, cslot = concat_slots(skip.rows, skip.H * skip.W, mid.C + skip.C)
    c = control.pop() if control is not None else None      # a None entry = no control at that tap (only_mid_control)
    em.add(mid.t, c.t if c is not None else None, mid.rows, mid.C, out=cat[:, :mid.C], stats_into=(cg, 0, cslot) if cg is not None else None)
    if cg is not None and not em.add_stats_done:
        em.arena.free(cg)
        cg = None
    if c is not None:
        em.free(c)
    em.free(mid)
    cur_C = mid.C
    B, H, W = skip.B, skip.H, skip.W
    nblk = len(a.output_blocks)
    for j, layers in enumerate(a.output_blocks):
        # right half of the concat: skip (+ control)
        c = control.pop() if control is not None else None
        em.add(skip.t, c.t if c is not None else None, skip.rows, skip.C, out=cat[:, cur_C:],
               stats_into=(cg, cur_C, cslot) if cg is not None else None)
        if cg is not None and not em.add_stats_done:
            em.arena.free(cg)
            cg = None
        if c is not None:
            em.free(c)
        em.free(skip)
        xin = Act(cat, B, H, W, cur_C + skip.C, cg, gn_slot=cslot or 128)      # (cg: both halves' partials, None = a statistics launch)
        if j + 1 < nblk:
            nskip = hs.pop()
            out_C = layers[-1].cout
            ncat = em.new_stream(nskip.rows, out_C + nskip.C, mirror=em.want_mirror("res.skip1x1"))
            ncg, nslot = concat_slots(nskip.rows, nskip.H * nskip.W, out_C + nskip.C)
            y = emit_block(em, P, layers, xin, table, offs, kv, out=ncat[:, :out_C], keep_input=True,
                           gnp_into=(ncg, 0, nslot) if ncg is not None else None)
            if ncg is not None and not em.gnp_into_done:      # the block's last layer could not write its half: keep the statistics launch
                em.arena.free(ncg)
                ncg = None
            em.free(cat)
            if cg is not None:
                em.arena.free(cg)
            cat, skip, cur_C, cg, cslot = ncat, nskip, out_C, ncg, nslot
            B, H, W = nskip.B, nskip.H, nskip.W
        else:
            y = emit_block(em, P, layers, xin, table, offs, kv, keep_input=True)
            em.free(cat)
            if cg is not None:
                em.arena.free(cg)
    n = em.group_norm(y, P + "out.0.", 1e-5, True, feeds=("unet.out_conv",))
    em.free(y)
    eps = em.conv(n, P + "out.2.", out_f32=True, name="unet.out_conv")
    em.free(n)
    return eps.t


# ----------------------------------------------------------------------------------------------
# VAE
# ----------------------------------------------------------------------------------------------
def _g_norm(em: Emitter, x: Act, prefix: str, silu: bool, feeds=None, conv_n: int = 0, take: bool = False):
    """GroupNorm as a suspension point of a VAE emission generator: yields the activation (the driver answers with the
    fp64 sums slice to fill), emits the statistics launch, yields again (the driver may pool the sums across tiles),
    then emits the apply launch.  Plain and tiled VAE share every other line of emission code."""
    sums = yield x
    apply = em.gn_stats_into(x, prefix, 1e-6, silu, sums, sums_zeroed=True, feeds=feeds, conv_n=conv_n, take=take)     # slots of the program's pre-zeroed pool
    yield None
    return apply()


def _g_vae_resblock(em: Emitter, p: str, l: VaeLayer, x: Act):
    """model/vae.py:103-124 (tiled form: resblock2task, utils/tilevae/tilevae.py:86-106)."""
    n1 = yield from _g_norm(em, x, p + "norm1.", True, ("vae.conv1",), conv_n=l.cout)
    h = em.conv(n1, p + "conv1.", name="vae.conv1", stats=True, feeds=("vae.conv2",))
    em.free(n1)
    n2 = yield from _g_norm(em, h, p + "norm2.", True, ("vae.conv2",), conv_n=l.cout, take=True)
    em.free(h)
    if l.cin != l.cout:
        skip = em.conv(x, p + "nin_shortcut.", taps=1, name="vae.nin_shortcut").t
    else:
        skip = x.t
    y = em.conv(n2, p + "conv2.", residual=skip, name="vae.conv2", stats=True)
    em.free(n2)
    if l.cin != l.cout:
        em.free(skip)
    return y


def _g_vae_attn(em: Emitter, p: str, x: Act):
    """model/vae.py:279-308: single-head attention with d = C (512): ONE launch of edtr_flash_attn512 (round 5: eight waves share
    128 queries, the scores never leave the CU) wherever N is whole 32-key tiles; otherwise — and behind EDTR_ATTN512=0 — the
    QK^T GEMM (fp32 scores) -> row softmax -> PV GEMM form of rounds 1 - 4.
    In the tiled VAE the same code runs per tile (tile-local attention, utils/tilevae/attn.py:85-115)."""
    B, N, C = x.B, x.H * x.W, x.C
    rows = B * N
    if N % 4:
        raise ValueError("VAE attention needs h*w to be a multiple of 4")
    n = yield from _g_norm(em, x, p + "norm.", False, ("vae.attn.qk", "vae.attn.vT"))
    wqk, bqk = em.store.linear([p + "q.weight", p + "k.weight"], [p + "q.bias", p + "k.bias"])
    qk = em.gemm(n.t, wqk, rows, 2 * C, C, bias=bqk, name="vae.attn.qk", out16=True)
    wv, _ = em.store.linear([p + "v.weight"])
    vt, ldv = em.vt_gemm(wv, n.t, B=B, Ntok=N, Cin=C, bias_m=em.store.vec(p + "v.bias"), name="vae.attn.vT")
    em.free(n)
    adt = em.attn_dtype
    if qk.dtype == torch.float32:      # high mode: the two batched products of this one layer run on fp16 operands cut from the fp32 projections
        qk32, vt32 = qk, vt
        qk = em.to16(qk32, rows, 2 * C)
        vt = em.to16(vt32, B * C, ldv)
        em.free(qk32, vt32)
    o = em.new(rows, C)      # fp32 in the high-precision mode
    if ops.flash_attn512_ok(N, C):
        # one launch, the scores never leave the CU (round 5; before: QK^T GEMM -> fp32 score matrix -> edtr_softmax_rows -> PV GEMM)
        em.prog.add(ops.make_flash_attn512(dtype=adt, q=qk[:, :C], k=qk[:, C:], vt=vt, out=o, B=B, N=N, q_bs=N * qk.stride(0), q_ld=qk.stride(0),
                                           k_bs=N * qk.stride(0), k_ld=qk.stride(0), vt_bs=C * ldv, vt_ld=ldv, o_bs=N * o.stride(0),
                                           o_ld=o.stride(0), scale=1.0 / math.sqrt(C), out_f32=em.hp, name="vae.attn.flash"))
        em.free(qk, vt)
    else:
        lds = round_up(N, 8)
        s = em.new(rows, lds, torch.float32)
        em.prog.add(ops.make_igemm(dtype=adt, a1=qk[:, :C], w=qk[:, C:], out=s, M=N, N=lds, n_valid=N, C1=C,
                                   ld1=qk.stride(0), ldw=qk.stride(0), ldc=lds, Z=B, a_zs=(N * qk.stride(0), 0),
                                   w_zs=(N * qk.stride(0), 0), o_zs=(N * lds, 0), alpha=1.0 / math.sqrt(C), out_f32=True,
                                   name="vae.attn.scores"))
        em.free(qk)
        pr = em.arena.alloc((rows, lds), adt)
        em.prog.add(ops.make_softmax_rows(dtype=adt, s=s, rows=rows, cols=N, ld_s=lds, p=pr, ld_p=lds, cols_pad=lds))
        em.free(s)
        # K is padded to a multiple of 8: the pad columns of P are exact zeros, those of V^T hold the (finite) bias
        em.prog.add(ops.make_igemm(dtype=adt, a1=pr, w=vt, out=o, M=N, N=C, C1=lds, ld1=lds, ldw=ldv, ldc=C, Z=B,
                                   a_zs=(N * lds, 0), w_zs=(C * ldv, 0), o_zs=(N * C, 0), out_f32=em.hp, name="vae.attn.pv"))
        em.free(pr, vt)
    wo, bo = em.store.linear([p + "proj_out.weight"], [p + "proj_out.bias"])
    y = em.gemm(o, wo, rows, C, C, bias=bo, residual=x.t, name="vae.attn.proj_out", stats_hw=N)
    em.free(o)
    return Act(y, x.B, x.H, x.W, C, em.last_gnp, gn_slot=em.last_gn_slot)


def gen_vae_net(em: Emitter, P: str, layers: List[VaeLayer], x: Act, final_f32: bool, final_nchw: Optional[torch.Tensor] = None):
    """Encoder.forward (model/vae.py:421-446) / Decoder.forward (:527-560) over the flat layer list, as a generator
    suspended at every GroupNorm (see _g_norm).  ``final_nchw`` (decoder): the fp32 NCHW tensor of the result — where
    edtr_conv128_out takes norm_out + SiLU + conv_out it writes that tensor itself and the generator returns None."""
    h = x
    first = True
    out_ch = final_nchw.shape[1] if final_nchw is not None else 0
    for l in layers:
        p = P + l.prefix
        last = l is layers[-1]
        if l.kind == "conv" and last and h.gn_in is not None and h.C == 128:
            em.conv128_out(h, p, final_nchw, out_ch)
            y = None
        elif l.kind == "conv":
            y = em.conv(h, p, out_f32=(final_f32 and last), name="vae.conv_in" if first else "vae.conv_out", stats=first)
        elif l.kind == "res":
            y = yield from _g_vae_resblock(em, p, l, h)
        elif l.kind == "attn":
            y = yield from _g_vae_attn(em, p, h)
        elif l.kind == "down":
            y = em.conv(h, p, stride=2, pad_tl=0, name="vae.downsample", stats=True)     # pad (0,1,0,1) + stride 2: vae.py:54-61
        elif l.kind == "up":
            y = em.conv(h, p, ups=True, name="vae.upsample.conv", stats=True)           # nearest x2 + conv: vae.py:35-39
        elif l.kind == "norm_out":
            y = yield from _g_norm(em, h, p, True, ("vae.conv_out",), conv_n=-out_ch if out_ch else 0, take=True)
        else:
            raise ValueError(l.kind)
        if not first:
            em.free(h)
        first = False
        h = y
    return h


def emit_vae_net(em: Emitter, P: str, layers: List[VaeLayer], x: Act, final_f32: bool, final_nchw: Optional[torch.Tensor] = None) -> Optional[Act]:
    """Untiled network: every GroupNorm uses its own statistics.  Returns None when the last convolution wrote ``final_nchw`` itself."""
    gen = gen_vae_net(em, P, layers, x, final_f32, final_nchw)
    sums = None
    try:
        req = next(gen)
        while True:
            sums = em.prog.sums_slot(em.arena, req.B)
            gen.send(sums)            # statistics launch emitted
            req = gen.send(None)      # apply launch emitted, emission continues up to the next GroupNorm
    except StopIteration as done:
        return done.value


def emit_vae_net_tiled(em: Emitter, P: str, layers: List[VaeLayer], tiles: List[Act], final_f32: bool) -> List[Act]:
    """VAEHook.vae_tile_forward (utils/tilevae/tilevae.py:452-579, non-fast mode): all tiles advance to their next
    GroupNorm, their per-(image, group) statistics are pooled with pixel-count weights (GroupNormParam.summary,
    :263-278) by edtr_gn_pool, the shared statistics normalise every tile, repeat."""
    gens = [gen_vae_net(em, P, layers, t, final_f32) for t in tiles]
    reqs = [next(g) for g in gens]
    results: List[Optional[Act]] = [None] * len(gens)
    dev = tiles[0].t.device
    while any(r is None for r in results):
        T = len(gens)
        BG = reqs[0].B * 32
        pix = [float(r.H * r.W) for r in reqs]
        weights = torch.tensor([v / sum(pix) for v in pix], dtype=torch.float32, device=dev)
        counts = torch.tensor([v * (reqs[0].C // 32) for v in pix], dtype=torch.float32, device=dev)
        sums = em.prog.sums_slot(em.arena, reqs[0].B, count=T).view(T, BG, 2)
        for k, g in enumerate(gens):
            g.send(sums[k])
        em.prog.add(ops.make_gn_pool(sums=sums, weights=weights, counts=counts, T=T, BG=BG))
        for k, g in enumerate(gens):
            try:
                reqs[k] = g.send(None)
            except StopIteration as done:
                results[k] = done.value
    return results


def split_tiles(h: int, w: int, tile_size: int, is_decoder: bool):
    """Tile input / output boxes [x1, x2, y1, y2] (VAEHook.split_tiles + get_best_tile_size,
    utils/tilevae/tilevae.py:325-395; pad 11 latent px for the decoder, 32 image px for the encoder)."""
    pad = 11 if is_decoder else 32

    def best(lower: int, upper: int) -> int:
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

    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    th = best(math.ceil((h - 2 * pad) / nh), tile_size)
    tw = best(math.ceil((w - 2 * pad) / nw), tile_size)
    ins, outs = [], []
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * tw, min(pad + (j + 1) * tw, w), pad + i * th, min(pad + (i + 1) * th, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w,
                  ib[2] if ib[2] > pad else 0, ib[3] if ib[3] < h - pad else h]
            outs.append([v * 8 if is_decoder else v // 8 for v in ob])
            ins.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return ins, outs
