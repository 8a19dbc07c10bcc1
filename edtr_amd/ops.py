"""Thin torch-tensor front end over the C ABI (edtr_amd/lib.py).  torch is plumbing here: it owns the
device buffers and the stream; every computation is a libedtr_hip launch on torch's current stream.

Launch records: each ``make_*`` returns a ``(c_function, params_struct, keepalive)`` tuple that can
be replayed any number of times with ``launch(rec)`` — the engine (edtr_amd/engine.py) pre-builds
them once per shape so the steady-state host cost per kernel is one ctypes call.
"""
from __future__ import annotations

import ctypes as ct
import os
from typing import Optional, Sequence, Tuple

import torch

from . import lib as L

F32S = "f32_split"     # high-precision mode: fp32 activation stream, bf16 split-3 GEMM operands (edtr_hip.h EDTR_F32_SPLIT)
# mixed-precision mode: fp32 activation stream, fp16 GEMM operands of 1 / 2 / 3 parts (edtr_hip.h EDTR_F32_H1 / H2 / H3)
F32H = {1: "f32_h1", 2: "f32_h2", 3: "f32_h3"}
MIXED = "f32_mixed"    # WeightStore dtype of the mixed mode: fp16 matrices packed per requested part count
PARTS_2W = 4           # precision-policy code of the WEIGHTS-EXACT two-part product: x16 . [Wh | Wl] with the A columns read twice
_DT = {torch.bfloat16: L.BF16, torch.float16: L.F16, F32S: L.F32_SPLIT, F32H[1]: L.F32_H1, F32H[2]: L.F32_H2, F32H[3]: L.F32_H3}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"libedtr_hip stores activations as bfloat16 or float16, got {dtype}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Rec:
    """One pre-built kernel launch."""
    __slots__ = ("fn", "args", "keep", "name", "flops", "bytes", "tag")

    def __init__(self, fn, args, keep, name, flops=0.0, nbytes=0.0):
        self.fn, self.args, self.keep, self.name, self.flops, self.bytes = fn, args, keep, name, flops, nbytes
        self.tag = ""           # shape tag for the per-shape bench breakdown

    def launch(self, stream: int) -> None:
        code = self.fn(*self.args, stream)
        if code != 0:
            L.check(code, self.name)


def launch(rec: Rec) -> None:
    rec.launch(stream_ptr())


# --------------------------------------------------------------------------------------------
# igemm
# --------------------------------------------------------------------------------------------
def make_igemm(*, dtype: torch.dtype, a1: torch.Tensor, w: torch.Tensor, out: torch.Tensor, M: int, N: int,
               C1: int, ld1: int, ldw: int, ldc: int, taps: int = 1, a2: Optional[torch.Tensor] = None, C2: int = 0,
               ld2: int = 0, spatial: Optional[Tuple[int, int, int, int, int, int, int, int]] = None, Z: int = 1,
               zdiv: int = 1, a_zs=(0, 0), w_zs=(0, 0), o_zs=(0, 0), alpha: float = 1.0,
               bias_n: Optional[torch.Tensor] = None, bias_m: Optional[torch.Tensor] = None,
               rowvec: Optional[torch.Tensor] = None, rowvec_ld: int = 0, rows_per_image: int = 0, act: int = 0,
               residual: Optional[torch.Tensor] = None, ldr: int = 0, out_f32: bool = False, n_valid: int = 0,
               tile: int = 0, splitk: int = 1, workspace: Optional[torch.Tensor] = None,
               gn_partial: Optional[torch.Tensor] = None, act_slope: float = 0.0, residual_f32: bool = False,
               vt_out: Optional[torch.Tensor] = None, vt_col0: int = 0, vt_ld: int = 0, vt_alpha: float = 1.0,
               row_stats: Optional[torch.Tensor] = None, ln_stats: Optional[torch.Tensor] = None, ln_C: int = 0, ln_valid: int = 0,
               ln_eps: float = 1e-5, ln_c1: Optional[torch.Tensor] = None, ln_c2: Optional[torch.Tensor] = None,
               w_phase_stride: int = 0, out16: Optional[torch.Tensor] = None, a_wrap: int = 0, a_gn: Optional[torch.Tensor] = None,
               a_gn_silu: bool = True, gn_slot_rows: int = 0, gn_ld: int = 0, name: str = "igemm") -> Rec:
    p = L.IgemmParams()
    p.dtype, p.taps, p.M, p.N, p.K = dt_code(dtype), taps, M, N, taps * (C1 + C2)
    p.n_valid, p.Z, p.zdiv = n_valid, Z, zdiv
    p.a1, p.a2, p.C1, p.C2, p.ld1, p.ld2 = ptr(a1), ptr(a2), C1, C2, ld1, ld2
    p.a_zs_outer, p.a_zs_inner = a_zs
    if spatial is not None:
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l, p.upsample2x = spatial
    p.w, p.ldw = ptr(w), ldw
    p.w_phase_stride = w_phase_stride     # sub-pixel upsample convolution (spatial[7] == 2): four pre-summed phase matrices
    if out16 is not None:                 # fp16 mirror of an fp32 stream output (mixed mode)
        p.out16, p.ld16 = ptr(out16), out16.stride(0)
    p.a_wrap = a_wrap                     # weights-exact two-part product: A columns read twice against [Wh | Wl]
    p.a_gn, p.a_gn_silu = ptr(a_gn), int(a_gn_silu)     # GroupNorm apply (+ SiLU) of the input fused into the halo tile's staging
    p.gn_slot_rows, p.gn_ld = int(gn_slot_rows), int(gn_ld)
    p.w_zs_outer, p.w_zs_inner = w_zs
    p.alpha = alpha
    p.bias_n, p.bias_m, p.rowvec = ptr(bias_n), ptr(bias_m), ptr(rowvec)
    p.rowvec_ld, p.rows_per_image, p.act = rowvec_ld, rows_per_image, act
    p.residual, p.ldr = ptr(residual), ldr
    p.out, p.ldc, p.out_f32 = ptr(out), ldc, int(out_f32)
    p.o_zs_outer, p.o_zs_inner = o_zs
    p.tile = tile
    p.splitk = splitk
    if splitk > 1:
        p.workspace, p.workspace_bytes = ptr(workspace), workspace.numel() * workspace.element_size()
    p.gn_partial = ptr(gn_partial)
    p.act_slope = act_slope
    p.residual_f32 = int(residual_f32)
    if vt_out is not None:
        p.vt_out, p.vt_col0, p.vt_ld, p.vt_alpha = ptr(vt_out), vt_col0, vt_ld, vt_alpha
    p.row_stats = ptr(row_stats)
    if ln_stats is not None:
        p.ln_stats, p.ln_slots, p.ln_C, p.ln_eps, p.ln_c1, p.ln_c2 = ptr(ln_stats), ln_C // 32, ln_valid or ln_C, ln_eps, ptr(ln_c1), ptr(ln_c2)
    flops = 2.0 * M * N * p.K * Z
    # algorithmic HBM bytes: every operand once (a conv reads its input image once, not once per tap)
    a_rows = (M // (p.OH * p.OW)) * p.IH * p.IW if spatial else M
    n_out = N // 2 if act == L.ACT_GEGLU else N
    nbytes = Z * (2.0 * a_rows * (C1 + C2) + 2.0 * N * p.K + (4.0 if out_f32 else 2.0) * M * n_out
                  + ((4.0 if residual_f32 else 2.0) * M * n_out if residual is not None else 0.0))
    rec = Rec(L.load().edtr_igemm, (ct.byref(p),), (p, a1, a2, w, out, bias_n, bias_m, rowvec, residual, workspace,
                                                    gn_partial, vt_out, row_stats, ln_stats, ln_c1, ln_c2, out16, a_gn), name, flops, nbytes)
    rec.tag = (f"taps{taps} M{M} N{N} K{p.K} Z{Z}" + (f" C2={C2}" if C2 else "") + (f" s{p.stride}" if spatial and p.stride != 1 else "")
               + ((" up2" if p.upsample2x == 1 else " up2sp") if spatial and p.upsample2x else "") + (f" sk{splitk}" if splitk > 1 else "") + (f" act{act}" if act else "")
               + (" f32" if out_f32 else "") + (" gnp" if gn_partial is not None else "") + (" vT" if vt_out is not None else "")
               + (" ln" if ln_stats is not None else "") + (" rs" if row_stats is not None else "")
               + (" m16" if out16 is not None else "") + (" 2w" if a_wrap else "") + (" gnin" if a_gn is not None else ""))
    return rec


LN_FOLD_MAX_BATCH = 4     # UNet / ControlNet LayerNorm fold: on by default up to this batch size (engine.Emitter.ln_fold_ok has the measurements)


def batch_invariant() -> bool:
    """EDTR_AMD_BATCH_INVARIANT=1: launch choices must not depend on the batch size (engine.Emitter)."""
    return os.environ.get("EDTR_AMD_BATCH_INVARIANT", "0") == "1"


def invariant_tile(C1: int, C2: int) -> int:
    """The one tile geometry of the batch-invariant mode: the 128x128 LDS-DMA loop where the operand layout allows it
    (Cin % 64 == 0, no fused concat), the 128x128 register-staged loop otherwise — never chosen from M."""
    return 3 if (C1 % 64 == 0 and not C2) else 1


def choose_splitk(M: int, N: int, K: int, Z: int = 1, act: int = 0, img8: bool = False) -> Tuple[int, int]:
    """(tile, splitk) heuristic for the 256-CU MI355X: when the 128x128 tile grid cannot fill the chip, cut K so
    that ~480 workgroups exist — but only while every split keeps enough K-tiles (of 64) to amortise the fp32 slab
    round trip of the reducer: >= 20 per split (>= 12 when fewer than 64 tiles exist at all).  Measured with
    tools/bench_kernels.py gsplitk / splitk: M=2048 N=1280 K=1280 runs 18.5 us unsplit vs 24.8 us at 3 splits, while
    K=5120 gains (53.5 -> 42.3 us) and the 8x8-level convolutions (M=512, K=11520) gain 3.4x at 6 splits."""
    if Z != 1 or act == L.ACT_GEGLU:
        return 0, 1
    if img8 and M % 256 == 0 and N % 128 == 0 and K % (9 * 64) == 0 and K // (9 * 64) >= 10 and (M // 256) * (N // 128) <= 64:
        # 3x3 convolutions of the 8x8 latent level on the halo kernel's 4-images-per-workgroup geometry: split over the
        # 64-channel chunks so that every workgroup multiplies 2 (K = 11520) .. 4 (K = 23040) chunks.  Measured
        # (profiles/r03/ab_tiles_3_vs_16_smallm.log, one device): 31.1 us at 10 splits against 36.7 us for the 128x128 loop at its
        # 6 splits (K = 11520), 42.2 against 59.3 us (K = 23040); batch 4: 27.1 against 33.1 us
        return 0, 10
    nkt = (K + 63) // 64
    b128 = ((M + 127) // 128) * ((N + 127) // 128)
    if b128 >= 200 or nkt < 24:
        return 0, 1
    want = max(1, round(480 / b128))
    per_split = 12 if b128 <= 64 else 20
    s = max(1, min(want, nkt // per_split, 6))
    return 0, s      # tile 0 = library default (LDS-DMA 128x128 main loop whenever Cin % 64 == 0)


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def make_flash_attn(*, dtype, q, k, vt, out, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, vt_bs, vt_ld, o_bs, o_ld,
                    scale: float, causal: bool = False, prescaled: bool = False, q_lo=None, k_lo=None, vt_lo=None,
                    out_f32: bool = False, name: str = "flash_attn64") -> Rec:
    """``q_lo`` / ``k_lo`` (/ ``vt_lo``): low halves of hi + lo operand pairs, same strides as q / k / vt (edtr_hip.h: split operands)."""
    p = L.AttnParams()
    p.dtype, p.B, p.H, p.Nq, p.Nk = dt_code(dtype), B, H, Nq, Nk
    p.q, p.q_bs, p.q_ld = ptr(q), q_bs, q_ld
    p.k, p.k_bs, p.k_ld = ptr(k), k_bs, k_ld
    p.vt, p.vt_bs, p.vt_ld = ptr(vt), vt_bs, vt_ld
    p.out, p.o_bs, p.o_ld = ptr(out), o_bs, o_ld
    p.scale = scale
    p.causal = int(causal)
    p.q_prescaled = int(prescaled)
    p.q_lo, p.k_lo, p.vt_lo, p.out_f32 = ptr(q_lo), ptr(k_lo), ptr(vt_lo), int(out_f32)
    flops = 4.0 * B * H * Nq * Nk * 64
    # algorithmic HBM bytes: Q and O once, K and V^T once per (image, head)
    nbytes = 2.0 * B * H * 64 * (2 * Nq + 2 * Nk)
    rec = Rec(L.load().edtr_flash_attn64, (ct.byref(p),), (p, q, k, vt, out, q_lo, k_lo, vt_lo), name, flops, nbytes)
    rec.tag = f"attn B{B} H{H} Nq{Nq} Nk{Nk}" + (" causal" if causal else "") + (" split" if q_lo is not None else "") + (" pv" if vt_lo is not None else "")
    return rec


def flash_attn512_ok(N: int, C: int) -> bool:
    """Does edtr_flash_attn512 take the VAE's single-head attention over N positions of C channels?  (head width 512, whole 32-key
    tiles; EDTR_ATTN512=0 keeps the GEMM -> softmax -> GEMM form for A/B runs)"""
    return os.environ.get("EDTR_ATTN512", "1") != "0" and C == 512 and N % 32 == 0


def make_flash_attn512(*, dtype, q, k, vt, out, B, N, q_bs, q_ld, k_bs, k_ld, vt_bs, vt_ld, o_bs, o_ld, scale: float, out_f32: bool = False,
                       name: str = "flash_attn512") -> Rec:
    """The VAE AttnBlock's softmax(q k^T scale) v in one launch (edtr_hip.h: edtr_flash_attn512): q / k rows of 512 channels, vt = V^T."""
    p = L.AttnParams()
    p.dtype, p.B, p.H, p.Nq, p.Nk = dt_code(dtype), B, 1, N, N
    p.q, p.q_bs, p.q_ld = ptr(q), q_bs, q_ld
    p.k, p.k_bs, p.k_ld = ptr(k), k_bs, k_ld
    p.vt, p.vt_bs, p.vt_ld = ptr(vt), vt_bs, vt_ld
    p.out, p.o_bs, p.o_ld = ptr(out), o_bs, o_ld
    p.scale, p.out_f32 = scale, int(out_f32)
    flops = 4.0 * B * N * N * 512
    nbytes = 2.0 * B * 512 * (3 * N) + (4.0 if out_f32 else 2.0) * B * N * 512
    rec = Rec(L.load().edtr_flash_attn512, (ct.byref(p),), (p, q, k, vt, out), name, flops, nbytes)
    rec.tag = f"attn512 B{B} N{N}" + (" f32" if out_f32 else "")
    return rec


def make_window_attn(*, dtype, qkv, ld_qkv, out, ld_out, B, H, W, heads, head_dim, c_pad, shift, bias, labels, scale,
                     name: str = "window_attn") -> Rec:
    """SwinIR shifted-window attention (edtr_hip.h: edtr_window_attn): qkv [B*H*W, 3*heads*32] -> out [B*H*W, c_pad]."""
    p = L.WindowAttnParams()
    p.dtype, p.B, p.H, p.W, p.heads, p.head_dim, p.shift = dt_code(dtype), B, H, W, heads, head_dim, shift
    p.qkv, p.ld_qkv, p.out, p.ld_out, p.c_pad = ptr(qkv), ld_qkv, ptr(out), ld_out, c_pad
    p.bias, p.labels, p.scale = ptr(bias), ptr(labels), scale
    tokens = B * H * W
    flops = 4.0 * tokens * 64 * heads * head_dim
    nbytes = 2.0 * tokens * (3 * heads * 32 + heads * head_dim)
    return Rec(L.load().edtr_window_attn, (ct.byref(p),), (p, qkv, out, bias, labels), name, flops, nbytes)


SWIN_MLP_C, SWIN_MLP_HIDDEN = 192, 384          # the one shape edtr_swin_mlp is built for (include/edtr_hip.h)


def _swin_image_rows(w: torch.Tensor) -> torch.Tensor:
    """16-bit [32 t, 192] -> t LDS images of a 32-row tile: chunk c of row r in slot c ^ ((r >> 1) & 7) (edtr_hip.h: edtr_swin_mlp w1)."""
    T, C = w.shape[0] // 32, w.shape[1]
    a = w.reshape(T, 32, C // 8, 8)
    r = torch.arange(32, device=w.device)[:, None]
    c = torch.arange(C // 8, device=w.device)[None, :]
    img = torch.empty_like(a)
    img[:, r, c ^ ((r >> 1) & 7)] = a[:, r, c]
    return img.reshape(-1).contiguous()


def _swin_image_slices(w: torch.Tensor) -> torch.Tensor:
    """16-bit [192, 32 t] -> t LDS images of a 32-column slice: chunk c of row r in slot c ^ ((r >> 2) & 3) (edtr_swin_mlp w2)."""
    C, T = w.shape[0], w.shape[1] // 32
    b = w.reshape(C, T, 4, 8).permute(1, 0, 2, 3)
    r = torch.arange(C, device=w.device)[:, None]
    c = torch.arange(4, device=w.device)[None, :]
    img = torch.empty((T, C, 4, 8), dtype=w.dtype, device=w.device)
    img[:, r, c ^ ((r >> 2) & 3)] = b[:, r, c]
    return img.reshape(-1).contiguous()


def pack_swin_mlp_weights(w1g: torch.Tensor, w2: torch.Tensor, dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 [384, 192] (gamma-scaled fc1) and [192, 384] (fc2), both already zero-padded -> the two LDS-image tensors
    edtr_swin_mlp copies by LDS-DMA (layout: include/edtr_hip.h, edtr_swin_mlp_params.w1 / .w2)."""
    assert tuple(w1g.shape) == (SWIN_MLP_HIDDEN, SWIN_MLP_C) and tuple(w2.shape) == (SWIN_MLP_C, SWIN_MLP_HIDDEN)
    return _swin_image_rows(w1g.to(dtype)), _swin_image_slices(w2.to(dtype))


SWIN_ATTN_HEADS = 6          # the one structure edtr_swin_attn is built for: 6 heads of <= 32 columns, 192 token columns, window 8


def pack_swin_attn_weights(wqkv: torch.Tensor, wproj: torch.Tensor, dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 [3 * heads * 32, 192] (the head-padded, gamma- and q-scaled qkv matrix, rows (s, h, e) as model.swinir.pack_qkv orders
    them) and [192, heads * 32] (proj with head-padded input columns) -> the LDS images of edtr_swin_attn (edtr_hip.h)."""
    H = SWIN_ATTN_HEADS
    assert tuple(wqkv.shape) == (3 * H * 32, SWIN_MLP_C) and tuple(wproj.shape) == (SWIN_MLP_C, H * 32)
    by_head = wqkv.reshape(3, H, 32, SWIN_MLP_C).permute(1, 0, 2, 3).reshape(3 * H * 32, SWIN_MLP_C)        # image 3 h + s
    return _swin_image_rows(by_head.to(dtype)), _swin_image_slices(wproj.to(dtype))


def swin_attn_bias(bias: torch.Tensor) -> torch.Tensor:
    """[heads, 64 queries, 64 keys] (model.swinir.expand_bias) -> [heads, 16 key groups, 64 queries, 4] for edtr_swin_attn."""
    h = bias.shape[0]
    return bias.reshape(h, 64, 16, 4).permute(0, 2, 1, 3).contiguous()


def make_swin_attn(*, dtype, x, ldx, out, ldo, B, H, W, head_dim, shift, c_valid, eps, wqkv, wproj, c1, c2b, bproj, bias, labels,
                   name="swin.attn") -> Rec:
    """One Swin layer's x + proj(WindowAttention(LayerNorm(x))) (edtr_hip.h: edtr_swin_attn)."""
    p = L.SwinAttnParams()
    p.dtype, p.B, p.H, p.W, p.heads, p.head_dim, p.shift = dt_code(dtype), B, H, W, SWIN_ATTN_HEADS, head_dim, shift
    p.C, p.c_valid, p.eps = SWIN_MLP_C, c_valid, eps
    p.x, p.ldx, p.out, p.ldo = ptr(x), ldx, ptr(out), ldo
    p.wqkv, p.wproj, p.c1, p.c2b, p.bproj, p.bias, p.labels = ptr(wqkv), ptr(wproj), ptr(c1), ptr(c2b), ptr(bproj), ptr(bias), ptr(labels)
    tokens = B * H * W
    flops = 2.0 * tokens * SWIN_MLP_C * 4 * SWIN_ATTN_HEADS * 32 + 4.0 * tokens * 64 * SWIN_ATTN_HEADS * head_dim
    nbytes = 2.0 * tokens * 2 * SWIN_MLP_C + 2.0 * 4 * SWIN_ATTN_HEADS * 32 * SWIN_MLP_C
    return Rec(L.load().edtr_swin_attn, (ct.byref(p),), (p, x, out, wqkv, wproj, c1, c2b, bproj, bias, labels), name, flops, nbytes)


def make_swin_mlp(*, dtype, x, ldx, rows, c_valid, eps, w1, w2, c1, c2b, b2, out, ldo, row_stats=None, name="swin.mlp") -> Rec:
    """One Swin layer's x + fc2(GELU(fc1(LayerNorm(x)))) (edtr_hip.h: edtr_swin_mlp)."""
    p = L.SwinMlpParams()
    p.dtype, p.rows, p.C, p.hidden, p.c_valid, p.eps = dt_code(dtype), rows, SWIN_MLP_C, SWIN_MLP_HIDDEN, c_valid, eps
    p.x, p.ldx, p.w1, p.w2 = ptr(x), ldx, ptr(w1), ptr(w2)
    p.c1, p.c2b, p.b2 = ptr(c1), ptr(c2b), ptr(b2)
    p.out, p.ldo, p.row_stats = ptr(out), ldo, ptr(row_stats)
    flops = 4.0 * rows * SWIN_MLP_C * SWIN_MLP_HIDDEN
    nbytes = 2.0 * rows * 2 * SWIN_MLP_C + 4.0 * SWIN_MLP_C * SWIN_MLP_HIDDEN
    return Rec(L.load().edtr_swin_mlp, (ct.byref(p),), (p, x, w1, w2, c1, c2b, b2, out, row_stats), name, flops, nbytes)


FFN_D, FFN_H, FFN_ROWS = 320, 1280, 128       # what edtr_ffn is built for (include/edtr_hip.h)
FFN_PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def ffn_ok(M: int, D: int, H: int) -> bool:
    """Does edtr_ffn (the feed-forward half of a transformer block in one launch) take this shape — and is it the faster form?  A
    workgroup owns 128 tokens for the WHOLE hidden dimension, so the launch lasts ~95 us however few rows there are: it pays where the
    rows fill at least half the chip (batch 8: 256 workgroups, + 1.4 % on the whole path; batch 4: 128 workgroups beside the other lane's
    launches, + 0.7 %), not on the 4096 rows of a latent tile of the tiled sampler (32 workgroups against two ~20-us GEMMs).
    EDTR_FFN=0 keeps the two-GEMM form everywhere (A/B runs); EDTR_FFN_MIN_ROWS overrides the threshold."""
    if os.environ.get("EDTR_FFN", "1") == "0" or D != FFN_D or H != FFN_H or M <= 0 or M % FFN_ROWS:
        return False
    return M >= int(os.environ.get("EDTR_FFN_MIN_ROWS", "16384"))


def pack_ffn_w2(w2: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[D, H] fp32 -> 16-bit with the columns permuted inside every aligned group of 16 (edtr_hip.h: edtr_ffn_params.w2)."""
    d, hid = w2.shape
    assert hid % 16 == 0
    perm = (torch.arange(hid // 16, device=w2.device)[:, None] * 16 + torch.tensor(FFN_PERM16, device=w2.device)[None, :]).reshape(-1)
    return w2[:, perm].to(dtype).contiguous()


def pack_ffn_constants(c2b: torch.Tensor) -> torch.Tensor:
    """The per-unit constants W1 beta + b1 in edtr_ffn's per-chunk order (edtr_hip.h: edtr_ffn_params.cst).  ``c2b`` is indexed by the
    PACKED (value / gate interleaved) w1 row; gate entries are stored halved."""
    n = c2b.numel()
    assert n % 128 == 0
    dev = c2b.device
    c, hh, q, vg, lh, e = torch.meshgrid(torch.arange(n // 128, device=dev), torch.arange(2, device=dev), torch.arange(4, device=dev),
                                         torch.arange(2, device=dev), torch.arange(2, device=dev), torch.arange(4, device=dev), indexing="ij")
    R = 128 * c + 64 * hh + 32 * vg + e + 8 * q + 4 * lh
    out = c2b.float()[R] * torch.where(vg == 1, 0.5, 1.0)
    return out.reshape(-1).contiguous()                      # [chunk][half][q][vg][lh][e]


def make_ffn(*, dtype, x, ldx, M, w1, w2, cst, b2, out, ldo, eps=1e-5, name="ff.fused") -> Rec:
    """x + ff(LayerNorm(x)) in one launch (edtr_hip.h: edtr_ffn)."""
    p = L.FfnParams()
    p.dtype, p.M, p.D, p.H, p.eps = dt_code(dtype), M, FFN_D, FFN_H, eps
    p.x, p.ldx, p.w1, p.w2, p.cst, p.b2 = ptr(x), ldx, ptr(w1), ptr(w2), ptr(cst), ptr(b2)
    p.out, p.ldo = ptr(out), ldo
    flops = 2.0 * M * FFN_D * 3 * FFN_H
    nbytes = 2.0 * M * 2 * FFN_D + 2.0 * 3 * FFN_D * FFN_H
    rec = Rec(L.load().edtr_ffn, (ct.byref(p),), (p, x, w1, w2, cst, b2, out), name, flops, nbytes)
    rec.tag = f"ffn M{M} D{FFN_D} H{FFN_H}"
    return rec


LIN320_K, LIN320_ROWS = 320, 128              # what edtr_lin320 is built for (include/edtr_hip.h)


def lin320_ok(M: int, N: int, K: int, ln: bool = False) -> bool:
    """Does edtr_lin320 (a K = 320 linear layer as a row-resident product, optionally behind its LayerNorm) take this shape — and is it
    the faster form?  Measured against the launches it replaces (profiles/r06/lin320_time.log): with a LayerNorm in front and N = 320 it
    wins from half a chip of 128-row workgroups (16384 rows: 15 against 20 us), everything else from a full one (32768 rows; at 16384 the
    plain forms and the N = 960 projection are 4 - 17 % slower than edtr_igemm's tiles).  EDTR_LIN320=0 keeps the edtr_igemm form
    everywhere (A/B runs); EDTR_LIN320_MIN_ROWS overrides the thresholds."""
    if os.environ.get("EDTR_LIN320", "1") == "0" or K != LIN320_K or M <= 0 or M % LIN320_ROWS or N % 64 or N > 1024:
        return False
    env = os.environ.get("EDTR_LIN320_MIN_ROWS")
    return M >= (int(env) if env else (16384 if (ln and N <= 320) else 32768))


def pack_lin320_w(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[N, 320] fp32 -> 16-bit in edtr_lin320's fragment order (edtr_hip.h: edtr_lin320_params.w): 16 bytes per (chunk of 32 rows, k-step of
    16, lane) = w[32 c + (lane & 31)][16 s + 8 (lane >> 5) .. + 7]."""
    n, k = w.shape
    assert k == LIN320_K and n % 32 == 0
    w16 = w.to(dtype).reshape(n // 32, 32, k // 16, 2, 8)            # [c][l31][s][lh][8]
    return w16.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)         # [c][s][lh][l31][8]: lane = 32 lh + l31


def make_lin320(*, dtype, x, ldx, M, N, w, cvec=None, alpha=1.0, ln=False, eps=1e-5, residual=None, ldr=0, out, ldo, vt_out=None, vt_col0=0,
                vt_ld=0, vt_alpha=1.0, rows_per_image=0, gn_table=None, name="lin320") -> Rec:
    """out = alpha * (LayerNorm?)(x) w^T + cvec (+ residual) in one launch; the columns from ``vt_col0`` on transposed into ``vt_out`` (the V^T
    operand of a fused [Wq; Wk; Wv] projection) where given (edtr_hip.h: edtr_lin320)."""
    p = L.Lin320Params()
    p.dtype, p.M, p.N, p.K, p.ln, p.eps, p.alpha = dt_code(dtype), M, N, LIN320_K, int(ln), eps, alpha
    p.x, p.ldx, p.w, p.cvec = ptr(x), ldx, ptr(w), ptr(cvec)
    p.residual, p.ldr, p.out, p.ldo = ptr(residual), ldr, ptr(out), ldo
    p.vt_out, p.vt_col0, p.vt_ld, p.vt_alpha, p.rows_per_image = ptr(vt_out), vt_col0, vt_ld, vt_alpha, rows_per_image
    p.gn_table = ptr(gn_table)
    flops = 2.0 * M * N * LIN320_K
    nbytes = 2.0 * M * (LIN320_K + N * (2 if residual is not None else 1)) + 2.0 * N * LIN320_K
    rec = Rec(L.load().edtr_lin320, (ct.byref(p),), (p, x, w, cvec, residual, out, vt_out, gn_table), name, flops, nbytes)
    rec.tag = (f"lin320 M{M} N{N}" + (" ln" if ln else "") + (" gnin" if gn_table is not None else "") + (" res" if residual is not None else "")
               + (" vT" if vt_out is not None else ""))
    return rec


def make_swin_layer(attn: Rec, mlp: Rec, name="swin.layer") -> Rec:
    """A whole Swin layer in one launch from the two half-layer records (edtr_hip.h: edtr_swin_layer): the attention record's x / out
    are the layer's input / output, the MLP record contributes its weights and constants."""
    pa, pm = attn.keep[0], mlp.keep[0]
    return Rec(L.load().edtr_swin_layer, (ct.byref(pa), ct.byref(pm)), (attn.keep, mlp.keep), name, attn.flops + mlp.flops,
               attn.bytes + mlp.bytes - 4.0 * pa.B * pa.H * pa.W * SWIN_MLP_C)


def pack_conv64_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[Cout <= 64, 64 (or fewer, zero padded), 3, 3] fp32 -> the nine 8-KiB LDS images of edtr_conv64 (edtr_hip.h)."""
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and co <= 64 and ci <= 64
    full = torch.zeros((9, 64, 8, 8), dtype=torch.float32, device=w.device)                   # [tap][n][chunk][j]
    full.reshape(9, 64, 64)[:, :co, :ci] = w.permute(2, 3, 0, 1).reshape(9, co, ci)
    n = torch.arange(64, device=w.device)[:, None]
    c = torch.arange(8, device=w.device)[None, :]
    img = torch.empty_like(full)
    img[:, n, c ^ ((n >> 1) & 7)] = full[:, n, c]
    return img.to(dtype).reshape(-1).contiguous()


def pack_conv128_out_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[Cout <= 32, 128, 3, 3] fp32 -> the nine 8-KiB LDS images of edtr_conv128_out (edtr_hip.h)."""
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and co <= 32 and ci == 128
    full = torch.zeros((9, 32, 16, 8), dtype=torch.float32, device=w.device)                  # [tap][n][chunk][j]
    full.reshape(9, 32, 128)[:, :co] = w.permute(2, 3, 0, 1).reshape(9, co, ci)
    n = torch.arange(32, device=w.device)[:, None]
    c = torch.arange(16, device=w.device)[None, :]
    img = torch.empty_like(full)
    img[:, n, c ^ (n & 15)] = full[:, n, c]
    return img.to(dtype).reshape(-1).contiguous()


def conv128_out_ok(H: int, W: int, cin: int, cout: int) -> bool:
    """Does edtr_conv128_out take this convolution (the VAE decoder's conv_out)?  EDTR_CONV128_OUT=0 keeps the three launches."""
    return os.environ.get("EDTR_CONV128_OUT", "1") != "0" and cin == 128 and cout <= 4 and H % 16 == 0 and W % 16 == 0


def make_conv128_out(*, dtype, x, ldx, w, bias, out, B, H, W, n_valid, gn_table=None, alpha=1.0, name="vae.conv_out") -> Rec:
    p = L.Conv128OutParams()
    p.dtype, p.B, p.H, p.W = dt_code(dtype), B, H, W
    p.x, p.ldx, p.gn_table, p.w, p.bias = ptr(x), ldx, ptr(gn_table), ptr(w), ptr(bias)
    p.alpha, p.out, p.n_valid = alpha, ptr(out), n_valid
    flops = 2.0 * B * H * W * 128 * 9 * n_valid
    nbytes = 2.0 * B * H * W * 128 + 4.0 * B * H * W * n_valid
    return Rec(L.load().edtr_conv128_out, (ct.byref(p),), (p, x, w, bias, out, gn_table), name, flops, nbytes)


def conv64_ok(H: int, W: int, cin: int, cout: int) -> bool:
    """Does edtr_conv64 take this 3x3 convolution (output H x W)?  EDTR_CONV64=0 keeps edtr_igemm (A/B runs)."""
    return os.environ.get("EDTR_CONV64", "1") != "0" and cin == 64 and cout <= 64 and H % 16 == 0 and W % 16 == 0


def make_conv64(*, dtype, x, ldx, w, bias, out, B, H, W, upsample2x=False, act=0, act_slope=0.0, alpha=1.0, ldo=0, out_nchw_f32=False,
                n_valid=0, name="conv64") -> Rec:
    p = L.Conv64Params()
    p.dtype, p.B, p.H, p.W, p.upsample2x = dt_code(dtype), B, H, W, int(upsample2x)
    p.x, p.ldx, p.w, p.bias = ptr(x), ldx, ptr(w), ptr(bias)
    p.act, p.act_slope, p.alpha = act, act_slope, alpha
    p.out, p.ldo, p.out_nchw_f32, p.n_valid = ptr(out), ldo, int(out_nchw_f32), n_valid
    flops = 2.0 * B * H * W * 64 * 9 * (n_valid if out_nchw_f32 else 64)
    src = B * H * W * 64 * 2 / (4 if upsample2x else 1)
    nbytes = src + (B * H * W * n_valid * 4 if out_nchw_f32 else B * H * W * 64 * 2)
    return Rec(L.load().edtr_conv64, (ct.byref(p),), (p, x, w, bias, out), name, flops, nbytes)


# --------------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------------
def make_embed_tokens(*, dtype, tokens, table, pos, rows, L_ctx, D, out, ld, name="embed_tokens") -> Rec:
    args = (dt_code(dtype), ptr(tokens), ptr(table), ptr(pos), rows, L_ctx, D, table.shape[0], ptr(out), ld)
    return Rec(L.load().edtr_embed_tokens, args, (tokens, table, pos, out), name, 0.0, 10.0 * rows * D)


def make_zero(t: torch.Tensor, name: str = "zero") -> Rec:
    nbytes = t.numel() * t.element_size()
    return Rec(L.load().edtr_zero_bytes, (ptr(t), nbytes), (t,), name, 0.0, float(nbytes))


# edtr_gn_apply folds the producer's per-tile partials itself up to this many 128-row tiles per image (the kernel accepts 64).
# Every one of the apply launch's ~2000 workgroups repeats the fold for its channels, so it only pays while tiles x channels is
# small next to a workgroup's own slab: the 16x16 and 32x32 latent levels (2 / 8 tiles); at 64x64 (32 tiles) the whole path
# lost 6 % (profiles/r03/experiments_gn_fold.log) and the separate edtr_gn_finalize launch stays.
GN_FOLD_MAX_TILES = int(__import__("os").environ.get("EDTR_GN_FOLD_MAX", "8"))


def gn_foldable(HW: int, C: int, groups: int = 32, tiles: int = 0) -> bool:
    """Can edtr_gn_apply fold the producing igemm's partials itself (no edtr_gn_finalize launch)?  ``tiles``: slots per image of the
    partials (0 = HW / 128: the main-loop epilogues' 128-row slots)."""
    if tiles <= 0:
        if HW % 128:
            return False
        tiles = HW // 128
    return tiles <= GN_FOLD_MAX_TILES and C // groups <= 64


def make_gn(*, dtype, x, ldx, B, HW, C, sums, gamma, beta, eps, silu, y, ldy, groups: int = 32, name="gn",
            sums_zeroed: bool = False, partial=None, tiles_per_image: int = 0):
    """Returns (stats_rec, apply_rec).  ``partial``: the producing igemm's gn_partial tensor — the apply launch folds it
    itself (gn_foldable) and ``sums`` may be None."""
    p = L.GnParams()
    p.dtype, p.B, p.HW, p.C, p.groups = dt_code(dtype), B, HW, C, groups
    p.x, p.ldx, p.sums = ptr(x), ldx, ptr(sums)
    p.gamma, p.beta, p.eps, p.silu = ptr(gamma), ptr(beta), eps, int(silu)
    p.y, p.ldy = ptr(y), ldy
    p.sums_zeroed = int(sums_zeroed)
    if partial is not None:
        p.partial, p.tiles_per_image = ptr(partial), tiles_per_image or HW // 128
    keep = (p, x, sums, gamma, beta, y, partial)
    lib = L.load()
    nb = 2.0 * B * HW * C
    return (Rec(lib.edtr_gn_stats, (ct.byref(p),), keep, name + ".stats", 0.0, nb),
            Rec(lib.edtr_gn_apply, (ct.byref(p),), keep, name + ".apply", 0.0, 2 * nb))


def make_gn_finalize(*, partial, tiles_per_image, B, C, sums, groups: int = 32, name="gn.finalize") -> Rec:
    return Rec(L.load().edtr_gn_finalize, (ptr(partial), tiles_per_image, B, C, groups, ptr(sums)), (partial, sums), name)


def make_gn_table(*, partial, tiles_per_image, sums, B, C, HW, gamma, beta, eps, table, groups: int = 32, name="gn.table") -> Rec:
    """(scale, shift) per image and channel for a consumer that normalises the tensor itself (edtr_hip.h: edtr_gn_table, a_gn)."""
    return Rec(L.load().edtr_gn_table, (ptr(partial), tiles_per_image, ptr(sums), B, C, groups, HW, ptr(gamma), ptr(beta), eps, ptr(table)),
               (partial, sums, gamma, beta, table), name)


def igemm_fast_addressable(rows_in: int, ld: int, IW: int, K: int, N: int, ldw: int) -> bool:
    """Mirror of igemm.hip: igemm_fast_addressable — the buffer-addressed kernels (every LDS-DMA tile, the halo tiles) reach their
    operands through 32-bit byte offsets; edtr_igemm answers EDTR_E_UNSUPPORTED for an explicit halo request beyond that, so the
    predicates below must not promise the halo tile there (the untiled 1024 x 1024 decode at batch 8 has a 4.29-GB
    operand).  tests/test_host_logic.py checks the predicates against edtr_igemm_plan."""
    a_bytes = (rows_in + 3 * IW + 3) * ld * 2 + K * 2
    w_bytes = N * ldw * 2 + K * 2
    return a_bytes < 0xF0000000 and w_bytes < 0xF0000000


def gn_in_conv_ok(B: int, H: int, W: int, C: int, N: int, splitk: int = 1, ld: int = 0) -> bool:
    """Does the halo tile take this 3x3 / stride 1 / pad 1 convolution in its 16 x 16-patch geometry with the GroupNorm of its input
    fused into the patch staging (edtr_hip.h: a_gn)?  The shape rules of edtr_igemm's automatic halo choice: 128-column tiles
    without padding and >= 48 units — and N <= EDTR_GN_IN_CONV_MAXN (default 128): measured on the MI355X (profiles/r04/gn_in_conv_ab.log),
    the fused form wins where ONE 128-column tile covers the output channels (the VAE's 512 x 512 level: headline +1.4 %) and loses
    beyond (N = 256: -0.5 % against that, N = 512 / 640 / 1280: the whole path -1 .. -2 %), because every column tile of a patch
    normalises the patch again.  EDTR_GN_IN_CONV=0 keeps the edtr_gn_apply launch everywhere (A/B runs)."""
    if os.environ.get("EDTR_GN_IN_CONV", "1") == "0" or os.environ.get("EDTR_IGEMM_HALO", "1") == "0":
        return False
    if H % 16 or W % 16 or C % 64 or N % 128 or (H, W) == (8, 8) or N > int(os.environ.get("EDTR_GN_IN_CONV_MAXN", "128")):
        return False         # (every 128-column tile of the convolution normalises the whole patch again: N / 128 times the arithmetic of edtr_gn_apply)
    if max(splitk, 1) > C // 64 or not igemm_fast_addressable(B * H * W, ld or C, W, 9 * C, N, 9 * C):
        return False         # (edtr_igemm's own preconditions for the halo tile: split-K over whole chunks, 32-bit operand offsets)
    return (B * H * W // 256) * (N // 128) * max(splitk, 1) >= 48


def gn_slot_rows(hw: int) -> int:
    """Rows per slot of the fused GroupNorm partials for images of ``hw`` pixels (0 = no fused statistics): 128-row slots wherever an
    image is whole slots; 64-row slots for the 8 x 8 images of the deepest latent level (the split-K reducer writes those)."""
    if hw > 0 and hw % 128 == 0:
        return 128
    return 64 if hw == 64 else 0


def gn_fusable(M: int, N: int, C1: int, hw: int, taps: int = 1, C2: int = 0, splitk: int = 1, invariant: bool = False) -> bool:
    """Can the producing edtr_igemm also emit GroupNorm partials?  Without split-K: whole 128-row tiles inside one image on the
    128x128 kernels.  With split-K (round 6) the REDUCER writes them: any slot size gn_slot_rows() knows, N % 32 == 0.
    EDTR_GN_SPLITK_STATS=0 keeps the edtr_gn_stats launch behind split-K producers (A/B runs)."""
    if C2 or N % 32:
        return False
    if splitk > 1:
        sr = gn_slot_rows(hw)
        return os.environ.get("EDTR_GN_SPLITK_STATS", "1") != "0" and sr > 0 and M % sr == 0
    if hw % 128 or M % 128:
        return False
    dma_ok = C1 % 64 == 0
    big = ((M + 127) // 128) * ((N + 127) // 128)
    return dma_ok or invariant or big >= 200      # (register-staged shapes: the 128x128 tile 1, which the invariant mode always takes)


def gn_tiles(hw: int, splitk: int = 1) -> int:
    """Slots per image of a fused-partials buffer written by a producer with this split-K count."""
    return hw // (gn_slot_rows(hw) if splitk > 1 else 128)


def make_layernorm(*, dtype, x, rows, C, ldx, gamma, beta, eps, y, ldy, c_valid: int = 0, name="layernorm") -> Rec:
    """c_valid (0 = C): statistics over the first c_valid columns only; the remaining columns of y are written as zeros."""
    args = (dt_code(dtype), ptr(x), rows, C, c_valid, ldx, ptr(gamma), ptr(beta), eps, ptr(y), ldy)
    return Rec(L.load().edtr_layernorm, args, (x, gamma, beta, y), name, 0.0, 4.0 * rows * C)


def make_softmax_rows(*, dtype, s, rows, cols, ld_s, p, ld_p, cols_pad=0, name="softmax_rows") -> Rec:
    args = (dt_code(dtype), ptr(s), rows, cols, ld_s, ptr(p), ld_p, cols_pad)
    return Rec(L.load().edtr_softmax_rows, args, (s, p), name, 0.0, 10.0 * rows * cols)


# --------------------------------------------------------------------------------------------
# layout / elementwise
# --------------------------------------------------------------------------------------------
def make_nchw_to_nhwc(*, dtype, src, B, C, HW, dst, ld, coff=0, zero_pad_to=0, scale=1.0, shift=0.0,
                      name="nchw_to_nhwc") -> Rec:
    args = (dt_code(dtype), ptr(src), B, C, HW, ptr(dst), ld, coff, zero_pad_to, scale, shift)
    return Rec(L.load().edtr_nchw_to_nhwc, args, (src, dst), name, 0.0, 6.0 * B * C * HW)


def make_pixel_unshuffle(*, dtype, src, B, C, H, W, r, dst, ld, sub=None, scale=1.0, zero_pad_to=0,
                         name="pixel_unshuffle") -> Rec:
    args = (dt_code(dtype), ptr(src), B, C, H, W, r, ptr(sub), scale, ptr(dst), ld, zero_pad_to)
    return Rec(L.load().edtr_pixel_unshuffle, args, (src, sub, dst), name, 0.0, 6.0 * B * C * H * W)


def make_nhwc_to_nchw(*, dtype, src, src_f32, B, C, HW, ld, dst, scale=1.0, name="nhwc_to_nchw") -> Rec:
    args = (dt_code(dtype), ptr(src), int(src_f32), B, C, HW, ld, ptr(dst), scale)
    return Rec(L.load().edtr_nhwc_to_nchw, args, (src, dst), name, 0.0, 6.0 * B * C * HW)


def make_add(*, dtype, a, lda, b, ldb, out, ldo, rows, C, name="add") -> Rec:
    args = (dt_code(dtype), ptr(a), lda, ptr(b), ldb, ptr(out), ldo, rows, C)
    return Rec(L.load().edtr_add, args, (a, b, out), name, 0.0, (6.0 if b is not None else 4.0) * rows * C)


def make_add_stats(*, dtype, a, lda, b, ldb, out, ldo, rows, C, gn_partial, gn_ld, slot_rows, name="add") -> Rec:
    """a (+ b) -> out and the result's per-slot column statistics in edtr_igemm's gn_partial format (edtr_hip.h: edtr_add_stats);
    ``gn_partial`` is a 1-D view that starts at this tensor's first column inside slot 0."""
    args = (dt_code(dtype), ptr(a), lda, ptr(b), ldb, ptr(out), ldo, rows, C, ptr(gn_partial), gn_ld, slot_rows)
    return Rec(L.load().edtr_add_stats, args, (a, b, out, gn_partial), name, 0.0, (6.0 if b is not None else 4.0) * rows * C)


def make_add_mirror(*, a, lda, b, ldb, out, ldo, out16, rows, C, name="add") -> Rec:
    """fp32 a (+ b) -> fp32 out and its fp16 mirror (edtr_hip.h: edtr_add_mirror)."""
    args = (ptr(a), lda, ptr(b), ldb, ptr(out), ldo, ptr(out16), out16.stride(0), rows, C)
    return Rec(L.load().edtr_add_mirror, args, (a, b, out, out16), name, 0.0, (14.0 if b is not None else 10.0) * rows * C)


def make_timestep_embedding(*, dtype, t, B, dim, out, ld, name="timestep_embedding") -> Rec:
    return Rec(L.load().edtr_timestep_embedding, (dt_code(dtype), ptr(t), B, dim, ptr(out), ld), (t, out), name)


def make_sampler_update(*, x, eps, noise, coefs: Sequence[float], x_prev, pred_x0, n, name="sampler_update") -> Rec:
    a, b, c1, c2, sigma = (float(v) for v in coefs)
    args = (ptr(x), ptr(eps), ptr(noise), a, b, c1, c2, sigma, ptr(x_prev), ptr(pred_x0), n)
    return Rec(L.load().edtr_sampler_update, args, (x, eps, noise, x_prev, pred_x0), name, 0.0, 20.0 * n)


def make_axpby(*, x, y, a, b, out, n, name="axpby") -> Rec:
    return Rec(L.load().edtr_axpby, (ptr(x), ptr(y), float(a), float(b), ptr(out), n), (x, y, out), name, 0.0, 12.0 * n)


def make_q_sample(*, x, noise, t, tab_a, tab_b, out, name="q_sample") -> Rec:
    B = x.shape[0]
    per = x.numel() // B
    args = (ptr(x), ptr(noise), ptr(t), ptr(tab_a), ptr(tab_b), tab_a.numel(), ptr(out), B, per)
    return Rec(L.load().edtr_q_sample, args, (x, noise, t, tab_a, tab_b, out), name, 0.0, 12.0 * x.numel())


def make_split3(*, src: torch.Tensor, rows: int, C: int, dst: torch.Tensor, pattern: int = 0, name="split3") -> Rec:
    code = L.F32_SPLIT if src.dtype == torch.float32 else dt_code(src.dtype)
    args = (code, ptr(src), rows, C, src.stride(0), pattern, ptr(dst), dst.stride(0))
    return Rec(L.load().edtr_split3, args, (src, dst), name, 0.0, (src.element_size() + 6.0) * rows * C)


def make_split_operand(*, src: torch.Tensor, rows: int, C: int, dst: torch.Tensor, fmt, name="split_operand") -> Rec:
    """[rows, C] fp32 / 16-bit -> GEMM operand in format ``fmt`` (F32S: bf16 [hi|lo|hi]; F32H[p]: fp16, p parts)."""
    code = L.F32_SPLIT if src.dtype == torch.float32 else dt_code(src.dtype)
    parts = 3 if fmt == F32S else {v: k for k, v in F32H.items()}[fmt]
    args = (code, ptr(src), rows, C, src.stride(0), dt_code(fmt), ptr(dst), dst.stride(0))
    return Rec(L.load().edtr_split_operand, args, (src, dst), name, 0.0, (src.element_size() + 2.0 * parts) * rows * C)


def make_sampler_update_indexed(*, x, eps, noise, index, coefs, x_prev, pred_x0, name="sampler_update") -> Rec:
    B = x.shape[0]
    per = x.numel() // B
    args = (ptr(x), ptr(eps), ptr(noise), ptr(index), ptr(coefs), coefs.shape[0], ptr(x_prev), ptr(pred_x0), B, per)
    return Rec(L.load().edtr_sampler_update_indexed, args, (x, eps, noise, index, coefs, x_prev, pred_x0), name, 0.0, 20.0 * x.numel())


def make_gaussian_sample(*, moments, ld, noise, out, B, C, HW, scale, name="gaussian_sample") -> Rec:
    args = (ptr(moments), ld, ptr(noise), ptr(out), B, C, HW, float(scale))
    return Rec(L.load().edtr_gaussian_sample, args, (moments, noise, out), name, 0.0, 16.0 * B * C * HW)


def make_cast16(*, dtype, src: torch.Tensor, rows: int, C: int, dst: torch.Tensor, name="cast16") -> Rec:
    args = (dt_code(dtype), ptr(src), rows, C, src.stride(0), ptr(dst), dst.stride(0))
    return Rec(L.load().edtr_cast16, args, (src, dst), name, 0.0, 6.0 * rows * C)


def make_tile_accumulate(*, tile, wts, out, count, B, C, H, W, th, tw, hi, wi, name="tile_accumulate") -> Rec:
    args = (ptr(tile), ptr(wts), ptr(out), ptr(count), B, C, H, W, th, tw, hi, wi)
    return Rec(L.load().edtr_tile_accumulate, args, (tile, wts, out, count), name)


def make_divide(*, num, den, out, n, name="divide") -> Rec:
    return Rec(L.load().edtr_divide, (ptr(num), ptr(den), ptr(out), n), (num, den, out), name)


def make_gn_pool(*, sums, weights, counts, T, BG, name="gn_pool") -> Rec:
    return Rec(L.load().edtr_gn_pool, (ptr(sums), ptr(weights), ptr(counts), T, BG), (sums, weights, counts), name)


def make_copy3d(*, src, src_plane, src_row, dst, dst_plane, dst_row, planes, rows, cols, name="copy3d") -> Rec:
    args = (ptr(src), src_plane, src_row, ptr(dst), dst_plane, dst_row, planes, rows, cols)
    return Rec(L.load().edtr_copy3d_f32, args, (src, dst), name, 0.0, 8.0 * planes * rows * cols)


def make_wavelet_level(*, src, low, high, planes, H, W, radius, name="wavelet_level") -> Rec:
    args = (ptr(src), ptr(low), ptr(high), planes, H, W, radius)
    return Rec(L.load().edtr_wavelet_level, args, (src, low, high), name, 0.0, 12.0 * planes * H * W)


# --------------------------------------------------------------------------------------------
# weight packing (host side, torch CPU or GPU tensors)
# --------------------------------------------------------------------------------------------
def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def split3_weight(w: torch.Tensor, dtype: torch.dtype = torch.bfloat16, parts: int = 3) -> torch.Tensor:
    """fp32 [..., C] -> 16-bit [..., parts*C] along the last axis, the weight side of a multi-part product:
    parts 3 = [hi | hi | lo] against the activation operand [hi | lo | hi] (xh*wh + xl*wh + xh*wl), parts 2 = [hi | hi] against
    [hi | lo] (the activation exact, the weight rounded once), parts 1 = [hi]."""
    hi = w.to(dtype)
    if parts == 1:
        return hi.contiguous()
    if parts == 2:
        return torch.cat([hi, hi], dim=-1).contiguous()
    lo = (w - hi.float()).to(dtype)
    if parts == PARTS_2W:       # [hi | lo] against the activation's one part read twice (edtr_hip.h: a_wrap): x16 . (Wh + Wl)
        return torch.cat([hi, lo], dim=-1).contiguous()
    return torch.cat([hi, hi, lo], dim=-1).contiguous()


def op_parts(parts: int) -> int:
    """Parts the ACTIVATION operand of a `parts`-policy product has (the weights-exact form reads one part twice)."""
    return 1 if parts == PARTS_2W else parts


def k_mult(parts: int) -> int:
    """K of a `parts`-policy product in units of the logical K."""
    return 2 if parts == PARTS_2W else parts


def _weight_format(dtype, parts: int):
    """(16-bit storage dtype, part count) of a packed matrix for a WeightStore dtype."""
    if dtype == F32S:
        return torch.bfloat16, 3
    if dtype == MIXED:
        if parts not in (1, 2, 3, PARTS_2W):
            raise ValueError(f"mixed-precision weights have 1..3 parts (or {PARTS_2W} = weights-exact two-part), got {parts}")
        return torch.float16, parts
    return dtype, 1


def pack_conv_weight(w: torch.Tensor, dtype, cin_pad: Optional[int] = None,
                     cout_pad: Optional[int] = None, parts: int = 1) -> torch.Tensor:
    """[Cout, Cin, kh, kw] fp32 -> [Cout_pad][kh][kw][Cin_pad] 16-bit, flattened to [N][K] (K contiguous).
    dtype == F32S / MIXED: [Cout_pad][kh][kw][parts*Cin_pad] with the channel axis split (split3_weight)."""
    co, ci, kh, kw = w.shape
    cip = cin_pad or round_up(ci, 8)
    cop = cout_pad or round_up(co, 8)
    out = torch.zeros((cop, kh, kw, cip), dtype=torch.float32, device=w.device)
    out[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
    dt16, parts = _weight_format(dtype, parts)
    return split3_weight(out, dt16, parts).reshape(cop, -1)


# sub-pixel form of nearest-2x upsample + 3x3 conv: which 3x3 taps add up on source offset d (0 / 1) of output parity p
SUBPIXEL_TAPS = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}


def pack_conv_weight_subpixel(w: torch.Tensor, dtype, cin_pad: Optional[int] = None, parts: int = 1) -> torch.Tensor:
    """[Cout, Cin, 3, 3] fp32 -> four phase matrices [4 * Cout_pad][2 * 2 * parts * Cin_pad] of the sub-pixel form of
    `interpolate(nearest, x2) -> conv3x3` (include/edtr_hip.h: w_phase_stride; reference model/unet.py:70-79, model/vae.py:35-39):
    output pixel (2s + py, 2r + px) = sum over (dy, dx) in {0, 1}^2 of W[py, px][dy][dx] . src[s + py - 1 + dy, r + px - 1 + dx] with
    W[py, px][dy][dx] = sum of w[ky][kx] over ky in SUBPIXEL_TAPS[py][dy], kx in SUBPIXEL_TAPS[px][dx].  The sums are taken in
    fp32 BEFORE the 16-bit (or multi-part) rounding, so the packed operand is at least as close to the fp32 kernel as nine
    separately rounded taps."""
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("the sub-pixel form is for 3x3 kernels")
    cip = cin_pad or round_up(ci, 8)
    cop = round_up(co, 8)
    wf = w.float()
    out = torch.zeros((4, cop, 2, 2, cip), dtype=torch.float32, device=w.device)
    for py in (0, 1):
        for px in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    acc = torch.zeros((co, ci), dtype=torch.float32, device=w.device)
                    for ky in SUBPIXEL_TAPS[py][dy]:
                        for kx in SUBPIXEL_TAPS[px][dx]:
                            acc = acc + wf[:, :, ky, kx]
                    out[2 * py + px, :co, dy, dx, :ci] = acc
    dt16, parts = _weight_format(dtype, parts)
    return split3_weight(out, dt16, parts).reshape(4 * cop, -1)


def subpixel_ok(H: int, W: int, Ce: int, N: int, B: int, ld: int = 0) -> bool:
    """Does the halo kernel's sub-pixel geometry take this nearest-2x upsample convolution (source H x W, Ce operand channels incl.
    parts, N output channels)?  16 x 16 source blocks, 64-channel chunks, 128-column tiles, and enough units to be worth a
    146-KiB workgroup (the halo tile's own threshold).  EDTR_SUBPIXEL=0 keeps the 9-tap gather (A/B runs)."""
    if os.environ.get("EDTR_SUBPIXEL", "1") == "0":
        return False
    units = B * (H // 16) * (W // 16) * 4 * ((N + 127) // 128)
    if not igemm_fast_addressable(B * H * W, ld or Ce, W, 9 * Ce, 4 * N, 4 * Ce):     # (four phase matrices of [N][4 Ce])
        return False
    return H % 16 == 0 and W % 16 == 0 and Ce % 64 == 0 and N % 128 == 0 and units >= 48


def pack_linear_weight(w: torch.Tensor, dtype, n_pad: Optional[int] = None, parts: int = 1) -> torch.Tensor:
    n, k = w.shape
    npad = n_pad or round_up(n, 8)
    out = torch.zeros((npad, round_up(k, 8)), dtype=torch.float32, device=w.device)
    out[:n, :k] = w
    dt16, parts = _weight_format(dtype, parts)
    return split3_weight(out, dt16, parts)


def geglu_perm(inner: int) -> torch.Tensor:
    """Row permutation that interleaves 32-row value / gate blocks (see edtr_hip.h, GEGLU epilogue)."""
    assert inner % 32 == 0
    j = torch.arange(inner // 32)
    val = (j[:, None] * 32 + torch.arange(32)[None, :])            # [J, 32]
    gate = val + inner
    return torch.cat([val, gate], dim=1).reshape(-1)               # [J * 64]


def pad_bias(b: Optional[torch.Tensor], n_pad: int) -> Optional[torch.Tensor]:
    if b is None:
        return None
    out = torch.zeros(n_pad, dtype=torch.float32, device=b.device)
    out[: b.numel()] = b.float()
    return out
