/*
 * edtr_hip.h — C ABI of libedtr_hip.so, the MI355X (gfx950) kernel library underneath the
 * EDTR ControlLDM restoration path.
 *
 * The reference (JaehaKim97/EDTR) has no native code: every device op on its hot path is an
 * ATen call made from torch.nn modules.  Each entry point below therefore names the reference
 * call site(s) (file:line under the reference tree) whose ATen op it replaces.  The Python
 * host (edtr_amd/) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD parameter structs; no torch / HIP types in signatures
 *     (a stream is passed as void* = hipStream_t; NULL = the default stream).
 *   - the caller owns every buffer (device memory unless stated); the library allocates
 *     nothing on the device and keeps no global mutable state; every launch is asynchronous
 *     on the given stream and is legal inside hipStreamBeginCapture (hipGraph capture).
 *   - activations are NHWC ("pixel-major") 16-bit (bf16 or fp16, chosen by `dtype`) with
 *     fp32 accumulation inside kernels; boundary tensors of the reference API (latents,
 *     images, eps) are NCHW fp32 and are converted by edtr_nchw_to_nhwc / edtr_nhwc_to_nchw.
 *   - 16-bit rows must be 16-byte aligned: channel counts, leading dimensions and column
 *     offsets are multiples of 8 elements.
 *   - return value: 0 = success; negative = EDTR_E* argument error (nothing was launched);
 *     positive = hipError_t from the launch.  Never throws, never exits.
 */
#ifndef EDTR_HIP_H
#define EDTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDTR_ABI_VERSION 10

enum edtr_dtype {
    EDTR_BF16 = 0, EDTR_F16 = 1,
    /* High-precision mode (accepted by the normalisation / layout / elementwise entry points that say so, never by
     * edtr_igemm / edtr_flash_attn64): the activation stream is fp32, and a tensor that feeds a GEMM is written as the bf16
     * "split-3" operand [hi | lo | hi] (3*C columns; hi = bf16(x), lo = bf16(x - hi)).  edtr_igemm (dtype EDTR_BF16, K = 3*C
     * per tap) multiplies it with weights packed [Wh | Wh | Wl]: hi*Wh + lo*Wh + hi*Wl = x*w to ~16 mantissa bits, fp32
     * accumulation, fp32 output.  This is what lets the path meet the 1e-3 parity target that 16-bit operands cannot. */
    EDTR_F32_SPLIT = 2,
    /* Mixed-precision mode: the same fp32 activation stream, but GEMM operands are FP16 and the number of operand parts is
     * chosen per layer (edtr_amd/precision.py): H1 = [x] (one fp16 rounding of the activation: a plain fp16 GEMM over an fp32
     * stream), H2 = [hi | lo] (2*C columns, hi = fp16(x), lo = fp16(x - hi): the activation is exact to ~22 bits, weights
     * packed [Wh | Wh] keep their single fp16 rounding), H3 = [hi | lo | hi] against [Wh | Wh | Wl] (~22 bits on both
     * sides).  A consumer may read a PREFIX of the parts ([hi], [hi | lo]) of a wider operand through ld1.  Every entry
     * point that accepts EDTR_F32_SPLIT accepts these; those that write no GEMM operand treat all four alike. */
    EDTR_F32_H1 = 3, EDTR_F32_H2 = 4, EDTR_F32_H3 = 5
};

enum edtr_error {
    EDTR_OK = 0,
    EDTR_E_NULL = -1,      /* required pointer is NULL */
    EDTR_E_SHAPE = -2,     /* non-positive or inconsistent extent */
    EDTR_E_ALIGN = -3,     /* pointer / leading dimension violates the 16-byte rule */
    EDTR_E_DTYPE = -4,     /* unknown dtype / flag value */
    EDTR_E_UNSUPPORTED = -5
};

enum edtr_act { EDTR_ACT_NONE = 0, EDTR_ACT_GEGLU = 1, EDTR_ACT_SILU = 2, EDTR_ACT_GELU = 3 /* exact erf GELU: the CLIP text MLP, model/open_clip/transformer.py:220-224; SwinIR Mlp, model/swinir.py:28-34 */,
                EDTR_ACT_LRELU = 4 /* x > 0 ? x : act_slope * x — SwinIR reconstruction convs, model/swinir.py:776,787,878-886 */ };

typedef void* edtr_stream_t; /* hipStream_t */

int edtr_abi_version(void);
const char* edtr_error_string(int code);
/* number of compute units / HBM bytes of the current device (0 on failure) */
int edtr_device_info(int* compute_units, int64_t* hbm_bytes, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear / batched GEMM on the matrix cores (MFMA 32x32x16).
 *
 *   out[z][m][n] = epilogue( alpha * sum_k A(z, m, k) * W[z][n][k] )
 *
 * replaces: F.conv2d 3x3 / 1x1 (reference model/unet.py:152,178,99-101,76-78,189;
 *           model/controlnet.py:138,261; model/vae.py:35-39,54-61,74-101), nn.Linear
 *           (model/attention.py:23,43,170-174,266,280; model/unet.py:168,476-480), the
 *           torch.cat feeding a conv (model/controlnet.py:35,37,266), F.interpolate(nearest,x2)
 *           feeding a conv (model/unet.py:76; model/vae.py:36), F.pad(0,1,0,1) (model/vae.py:57),
 *           and the bmm pair of the single-head VAE attention (model/vae.py:298).
 *
 * A operand ("im2col on the fly"): k = tap * (C1 + C2) + c, tap = ky * 3 + kx for taps == 9.
 *   pixel (b, oy, ox) = unflatten(m; OH, OW);  iy = oy*stride + ky - pad_t;  ix likewise;
 *   out-of-range taps read 0;  upsample2x != 0 reads source pixel (iy>>1, ix>>1) of an
 *   (IH, IW) source (logical input is 2IH x 2IW; upsample2x == 2: the same function in its sub-pixel
 *   form, see w_phase_stride);  channels c < C1 come from a1 (pixel stride
 *   ld1), the rest from a2 (pixel stride ld2) — the fused channel concat.
 *   taps == 1 and OH == 0: plain row-major matrix, A(z, m, k) = a1[z][m*ld1 + k] (concat too).
 * W operand: row-major [N][K] (K contiguous, i.e. conv weights packed [Cout][ky][kx][Cin]).
 *   Rows n >= n_valid read as zero.
 * z (grid.z) offsets: operand_offset = (z / zdiv) * zs_outer + (z % zdiv) * zs_inner (elements).
 * Epilogue order: *alpha, +bias_n[n], +bias_m[m], GEGLU (value/gate column blocks of 32
 *   interleaved by the weight packer; output has N/2 columns), +rowvec[(m / rows_per_image)][n],
 *   SiLU / GELU / LeakyReLU, +residual[m][n], then store as 16-bit or fp32 at out[m*ldc + n].
 * ---------------------------------------------------------------------------------------- */
typedef struct edtr_igemm_params {
    int32_t dtype;          /* edtr_dtype of a1/a2/w/residual and of a 16-bit output */
    int32_t taps;           /* 1 or 9 */
    int32_t M, N, K;        /* K = taps * (C1 + C2), multiple of 8; N multiple of 8 */
    int32_t n_valid;        /* W rows >= n_valid are zero (0 means N) */
    int32_t Z, zdiv;        /* batch count (grid.z) and its split (zdiv >= 1) */
    /* A */
    const void* a1; const void* a2;
    int32_t C1, C2, ld1, ld2;
    int64_t a_zs_outer, a_zs_inner;
    int32_t IH, IW, OH, OW; /* spatial mode when OH > 0 */
    int32_t stride, pad_t, pad_l, upsample2x;
    /* W */
    const void* w; int32_t ldw;
    int64_t w_zs_outer, w_zs_inner;
    /* epilogue */
    float alpha;
    const float* bias_n; const float* bias_m;
    const float* rowvec; int32_t rowvec_ld; int32_t rows_per_image;
    int32_t act;            /* edtr_act */
    const void* residual; int32_t ldr;
    /* out */
    void* out; int32_t ldc; int32_t out_f32;
    int64_t o_zs_outer, o_zs_inner;
    int32_t tile;           /* 0 = auto; explicit main-loop choice (tests / A-B runs): 1 = 128x128 register-staged, 2 = 64x64,
                               3 = 128x128 LDS-DMA (2 stages), 6 = 256x256 ping-pong, 8 = 128x160, 14 = 256x32 for N <= 32 (automatic for
                               large-M skinny-N convolutions), 16 = halo tile (3x3 / stride 1 / pad 1 convolutions — plain, nearest-2x
                               upsampled as a gather or in the sub-pixel form, or four whole 8x8 images per workgroup — on outputs whose
                               height and width are multiples of 16: the 18x18 input patch of a 16x16 output patch stays in LDS for
                               all taps; automatic for N % 128 == 0 and >= 48 units incl. split-K; EDTR_E_UNSUPPORTED for any other
                               shape); 17 = the halo tile on 32 x 16-pixel units (ABI 9, halo512.hip: 32-channel chunks, a wave owns 128
                               pixels x 64 channels over ALL of K, epilogue straight from the accumulators with whole-line stores; plain
                               3x3 / stride 1 / pad 1 convolutions with OW % 32 == 0, OH % 16 == 0, N % 128 == 0, C1 % 32 == 0, no split-K /
                               activation / bias_m; a_gn allowed; automatic where tile 16 would run and >= 256 such units exist:
                               1.10 - 1.13 x over tile 16 on the VAE's convolutions); 20 = the halo tile on 16 x 16 pixels x 160 channels
                               (ABI 9, halo512.hip: N % 160 == 0, otherwise tile 17's rules with OW % 16 == 0 and no a_gn; automatic
                               where the 128x160 tile would run and the units fill one round of the chip, 192 .. 256: the 64 x 64-latent
                               ResBlock convolutions at batch 8); 21 = tile 17 as a persistent kernel (ABI 10, halo512.hip: one workgroup per
                               CU walks its units, the finished 16-bit output of a unit leaves under the next unit's multiply loop; bit-identical
                               to tile 17; 16-bit output, no residual / rowvec, N <= 512, C1 % 64 == 0; never automatic — 1.00 - 1.05 x in
                               isolation, neutral on the whole path: profiles/r06/halo512p_*.log; EDTR_IGEMM_HALO512P=N picks it from N units
                               per CU for A/B runs).  4, 5, 7, 9 - 13, 15, 18, 19 were experiments (3-stage BK32, 256x128 tiles, 64x128,
                               16x16x32 at 128x128, deeper LDS rings, bank-swizzled epilogue staging, an 8-wave ping-pong 128x128 tile
                               for small grids, two-workgroup and persistent halo variants), measured without a whole-path gain
                               (profiles/r01 - r03) and removed: EDTR_E_DTYPE */
    /* split-K (small-M problems that cannot fill 256 CUs): K is cut into `splitk` runs of K-tiles, each
     * workgroup row writes an fp32 partial slab into `workspace` ([splitk][M][N] floats, caller-owned), and a
     * second launch sums the slabs and applies the epilogue.  splitk <= 1 disables it.  Needs Z == 1, no GEGLU. */
    int32_t splitk;
    void* workspace; int64_t workspace_bytes;
    /* fused GroupNorm statistics of the OUTPUT (optional): the epilogue also writes, per 128-row tile, the per-column
     * sum and sum of squares of the values it stores: gn_partial[(M/128)][N][2] fp32 (which rows a slot covers is the
     * kernel's business — the halo tile fills slot 2k with a 256-pixel patch and zeroes slot 2k+1 — only the per-image
     * totals over an image's H*W/128 consecutive slots are defined).  edtr_gn_finalize folds them into
     * the fp64 sums edtr_gn_apply consumes, so the separate statistics pass over the tensor (edtr_gn_stats) disappears.
     * Needs M % 128 == 0, no GEGLU / z-batching, tile 0/1/3/6/8/16 (16-bit or fp32 output).
     * With split-K (ABI 10) the REDUCER writes them (the main loops only write slabs): slots of gn_slot_rows rows (0 = 128; 64 for the
     * 8 x 8 images of the deepest latent level, where a 128-row slot would straddle two images), gn_partial[M / gn_slot_rows][N][2];
     * needs M % gn_slot_rows == 0 and N % 32 == 0.  Without split-K gn_slot_rows must be 0 or 128. */
    float* gn_partial;
    float act_slope;        /* negative-side slope of EDTR_ACT_LRELU (0 <= slope <= 1) */
    int32_t residual_f32;   /* nonzero: `residual` is fp32 (ldr in floats, multiple of 4): the fp32 activation stream of the
                               high / mixed precision modes adds its skip inside the epilogue instead of in a separate launch */
    /* Transposed second output (optional): the fused [Wq; Wk; Wv] projection of a self-attention in ONE launch.  Output columns
     * n >= vt_col0 are not stored at out[m][n] but TRANSPOSED, as the V^T operand edtr_flash_attn64 reads:
     *   vt_out[(m / rows_per_image) * (N - vt_col0) + (n - vt_col0)][m % rows_per_image]   (row stride vt_ld, 16-bit)
     * scaled by vt_alpha instead of alpha (the q / k halves carry the softmax scale, v does not); bias_n applies to all columns.
     * Needs rows_per_image % 8 == 0, M % 8 == 0, vt_ld % 8 == 0, vt_col0 a multiple of the column-tile width (128, or 160 for
     * tile 8: every SD width), Z == 1, no split-K / GEGLU / residual / rowvec, tile 0 / 1 / 3 / 8.  replaces: `to_v(x)` +
     * the `b n (h d) -> (b h) n d` rearrange of v, reference model/attention.py:172-178. */
    void* vt_out; int32_t vt_col0; int32_t vt_ld; float vt_alpha;
    /* LayerNorm folded into the GEMMs around it (fast modes; reference model/attention.py:222-224,230-234: x + attn(norm(x))).
     * LayerNorm is affine per row, so  LN(x) W^T = rstd_r (x (gamma . W)^T - mean_r c1) + c2  with c1[n] = sum_k (gamma . W)[n][k]
     * (of the packed 16-bit values) and c2[n] = sum_k beta[k] W[n][k]: the GEMM runs on the RAW rows against weights pre-multiplied
     * by gamma, and its epilogue applies the two row scalars — the normalised tensor is never written or read, the edtr_layernorm
     * launch disappears and one 16-bit rounding (of the normalised activations) with it.
     *   producer side — row_stats (optional): the launch that WRITES x also writes, per row, the sum and the sum of squares of
     *     the values it stores: row_stats[m][N / 32][2] fp32; a column tile fills the slot of its first 32 columns and zeroes
     *     the other slots it covers, so the totals over a row's N / 32 slots are defined whatever tile ran (no atomics:
     *     results stay bit-reproducible).  Needs N % 32 == 0, Z == 1, no split-K / GEGLU, a row-major tile (not tile 16).
     *   consumer side — ln_stats (optional): row statistics of the A operand's rows (ln_slots = K / 32 slots per row; ln_C <= K
     *     = the number of real columns the mean / variance run over — pad columns must be zero —, eps ln_eps); the epilogue computes out = rstd (alpha acc - mean alpha c1[n]) + alpha c2[n] + bias_n[n]
     *     (vt_alpha for the transposed V columns), then GEGLU / activation / residual as usual.  Needs Z == 1, no split-K,
     *     taps == 1, tile 0 / 1 / 3 (an automatic 128x160 choice falls back to tile 3). */
    float* row_stats;
    const float* ln_stats; int32_t ln_slots; int32_t ln_C; float ln_eps;
    const float* ln_c1; const float* ln_c2;
    int32_t stagger;        /* set by edtr_igemm itself (the caller's value is ignored): cycles by which the second workgroup of
                               every CU starts late, so that the two resident workgroups of the 128-row tiles do not run their K
                               loops and their store bursts in lockstep */
    int32_t debug_flags;    /* set by edtr_igemm itself from the environment (A/B measurements on one device): bit 0 =
                               EDTR_IGEMM_GENERAL_EPILOGUE=1, every launch takes the general epilogue row loop; bit 1 = EDTR_IGEMM_N160_TWO_PASS=1,
                               the 128x160 tile stages its accumulators in two passes of 64 rows; bit 2 = EDTR_IGEMM_GEGLU_SERIAL=1, the GEGLU epilogue evaluates
                               its gates one value after the other (the form before the eight-value lockstep evaluation) */
    /* Sub-pixel form of `F.interpolate(x, scale_factor=2, mode="nearest")` + 3x3 conv (ABI 7; upsample2x == 2; reference
     * model/unet.py:70-79, model/vae.py:35-39).  The 2 x 2 blocks of the upsampled image are constant, so output pixel
     * (2s + py, 2r + px) is a 2 x 2 convolution of the SOURCE image whose weights are sums of the 3 x 3 kernel's rows / columns:
     *   rows    py = 0: {w[0]} on source row s - 1, {w[1] + w[2]} on s;     py = 1: {w[0] + w[1]} on s, {w[2]} on s + 1   (columns alike)
     * i.e. 4 multiply-adds per output element and input channel instead of 9.  `w` then holds FOUR phase matrices
     * [2 py + px][N][dy][dx][C1] (K' = 4 C1 per row, row stride ldw >= 4 C1), `w_phase_stride` elements apart, summed in fp32 by
     * the packer BEFORE the 16-bit (or multi-part) rounding.  taps stays 9 and K = 9 C1 (the algorithmic shape).  Halo kernel only
     * (tile 0 / 16): IH, IW multiples of 16, C1 % 64 == 0, stride 1, pad 1; anything else is EDTR_E_UNSUPPORTED. */
    int64_t w_phase_stride;
    /* 16-bit MIRROR of an fp32 output (ABI 7; mixed-precision mode): when out_f32 != 0 and out16 != NULL the epilogue also stores
     * the final values rounded to `dtype` at out16[m * ld16 + n] — the one-part GEMM operand that the 16-bit consumers of the fp32
     * residual stream (1x1 skip / zero convolutions, down / upsample convolutions) read, instead of a separate cast launch per
     * consumer (edtr_split_operand).  Needs Z == 1, no GEGLU / transposed output; ld16 % 8 == 0. */
    void* out16; int32_t ld16;
    /* Weights-exact two-part product (ABI 7): a_wrap > 0 reads the A operand's columns TWICE, A(m, k) = a1[m * ld1 + (k mod a_wrap)],
     * K = 2 * a_wrap, against weights packed [Wh | Wl] (hi / lo halves of the fp32 weight): x16 . (Wh + Wl) — the weight rounding
     * disappears at twice the MFMA work, with no low part of the activation to form.  Plain GEMMs (taps 1, non-spatial, no concat),
     * a_wrap % 64 == 0, tiles 0 / 1 / 2 / 3 / 8. */
    int32_t a_wrap;
    /* GroupNorm apply (+ SiLU) of the INPUT fused into the operand staging (ABI 8; fast 16-bit modes): a_gn = fp32 [B][C1][2], the
     * (scale, shift) edtr_gn_table derives from the tensor's statistics; the convolution then multiplies
     *   act(a1 * scale[image][c] + shift[image][c]),  act = SiLU when a_gn_silu != 0, zero outside the image (the padding),
     * rounded to `dtype` exactly as edtr_gn_apply stores it — the normalised tensor is never written or read.  Halo tile only, in
     * its 16 x 16-patch geometry (taps 9, stride 1, pad 1, OH % 16 == OW % 16 == 0, C1 % 64 == 0, no upsample, not the 8 x 8 image
     * form; split-K allowed); anything else is EDTR_E_UNSUPPORTED and the caller issues edtr_gn_apply.
     * replaces: `F.silu(self.norm1(x))` / `in_layers[:2]` in front of the 3 x 3 convolutions of the ResBlocks, reference
     * model/vae.py:103-114, model/unet.py:203-218 (GroupNorm32 + SiLU, model/util.py:146-163). */
    const float* a_gn; int32_t a_gn_silu;
    int32_t gn_slot_rows;   /* rows per gn_partial slot when the split-K reducer writes the statistics (ABI 10; see gn_partial): 0 / 128 / 64 */
    int32_t gn_ld;          /* channels per gn_partial slot (ABI 10): 0 = N; > N when the output is a column slice of a concatenation whose
                               halves share one buffer of slots — gn_partial then points at this launch's first column inside a slot
                               (slot s, column n of this launch at gn_partial[(s * gn_ld + n) * 2]); edtr_add_stats writes the other half */
} edtr_igemm_params;

int edtr_igemm(const edtr_igemm_params* p, edtr_stream_t stream);
/* Which kernel would edtr_igemm run for these parameters (ABI 9)?  All of edtr_igemm's validation and shape rules, no launch and no
 * HIP call (works without a GPU): > 0 = the tile number (see `tile` above), < 0 = the error edtr_igemm would return.  The host side
 * uses it to keep its own launch-shape predicates (edtr_amd/ops.py: gn_in_conv_ok, subpixel_ok) provably inside the library's. */
int edtr_igemm_plan(const edtr_igemm_params* p);

/* ------------------------------------------------------------------------------------------
 * Fused multi-head attention, head width 64: out = softmax(q k^T * scale) v per (batch, head),
 * online softmax, scores never leave the CU.
 * replaces: F.scaled_dot_product_attention, reference model/attention.py:193 (46 calls/step).
 *   q   : [B][Nq][q_ld]  head h occupies columns h*64 .. h*64+63
 *   k   : [B][Nk][k_ld]  same column convention
 *   vt  : [B][H*64][vt_ld]  V transposed ("key-major"): row h*64+d holds v[:, h*64+d] over the
 *         keys; vt_ld >= roundup8(Nk) and the padding keys must be zero
 *   out : [B][Nq][o_ld]
 * ---------------------------------------------------------------------------------------- */
typedef struct edtr_attn_params {
    int32_t dtype;
    int32_t B, H, Nq, Nk;
    const void* q; int64_t q_bs; int32_t q_ld;
    const void* k; int64_t k_bs; int32_t k_ld;
    const void* vt; int64_t vt_bs; int32_t vt_ld;
    void* out; int64_t o_bs; int32_t o_ld;
    float scale;
    int32_t causal;                 /* nonzero: key j contributes to query i only if j <= i (the CLIP text tower's attn_mask,
                                       reference model/open_clip/model.py build_attention_mask; Nq == Nk) */
    int32_t q_prescaled;            /* nonzero: the q.k products already carry scale*log2(e) (the projection GEMMs that produced q
                                       and/or k applied it in fp32 before their single 16-bit rounding), so the kernel evaluates
                                       exp2(q.k - max) directly and ignores `scale`: one v_exp and no multiply per score */
    /* Split operands (ABI 7; the robust parity mode): q_lo / k_lo (both or neither) and optionally vt_lo hold the LOW halves of
     * hi + lo fp16 (bf16) pairs, laid out like q / k / vt (same strides): x = hi + lo to ~22 bits, as edtr_split_operand writes
     * them ([hi | lo] columns: lo = hi + C elements).  The products then run as three MFMA products each,
     *   S = Qh Kh^T + Ql Kh^T + Qh Kl^T,   O = Ph Vh + Pl Vh + Ph Vl  (P split in registers; PV only with vt_lo),
     * which removes the 16-bit rounding of the attention operands — on sharp attention (tests/golden/heavy.npz) the rounding
     * of q and k alone costs 2.7e-3 of a denoiser evaluation (tests/heavy_attention_budget.py).  out_f32 != 0 (split mode only):
     * fp32 output (o_ld / o_bs in floats).  replaces: the same F.scaled_dot_product_attention call in fp32. */
    const void* q_lo; const void* k_lo; const void* vt_lo;
    int32_t out_f32;
} edtr_attn_params;

int edtr_flash_attn64(const edtr_attn_params* p, edtr_stream_t stream);

/* Fused single-head attention with head width 512 (ABI 9): the VAE's AttnBlock, softmax(q k^T * scale) v over the latent positions.
 * Same parameter block as edtr_flash_attn64 with H == 1 and rows of 512 channels: q [B][Nq][q_ld], k [B][Nk][k_ld],
 * vt = V^T [B][512][vt_ld] (vt_ld >= Nk), out [B][Nq][o_ld] (16-bit, or fp32 with out_f32); `scale` is applied inside.
 * Nk must be a multiple of 32 (EDTR_E_UNSUPPORTED otherwise: the caller keeps GEMM -> edtr_softmax_rows -> GEMM); any Nq;
 * no causal mask / pre-scaled q / split operands.  The score matrix never exists in memory (4096^2 fp32 per image before).
 * replaces: F.scaled_dot_product_attention in AttnBlock.forward, reference model/vae.py:279-308 (and the tile-local attention of
 * utils/tilevae/attn.py:85-115). */
int edtr_flash_attn512(const edtr_attn_params* p, edtr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (32 groups in the reference; `groups` here) over NHWC 16-bit activations, fp32 math.
 * replaces: GroupNorm32 (model/util.py:146-163, eps 1e-5), Normalize (model/attention.py:50-51,
 *           model/vae.py:22-23, eps 1e-6) and the SiLU / x*sigmoid(x) that follows
 *           (model/unet.py:150,175,677; model/vae.py:17-19,104-113,443-444,555-556).
 * Two launches: edtr_gn_stats accumulates per-(image, group) sum / sum of squares in fp64
 * (sums[B][groups][2], zeroed by the call itself); edtr_gn_apply normalises, applies the affine
 * and optionally SiLU.
 * ---------------------------------------------------------------------------------------- */
typedef struct edtr_gn_params {
    int32_t dtype;
    int32_t B, HW, C, groups;
    const void* x; int32_t ldx;     /* pixel stride of x (>= C) */
    double* sums;                   /* [B][groups][2] workspace */
    const float* gamma; const float* beta; /* [C] */
    float eps;
    int32_t silu;                   /* 0 / 1 */
    void* y; int32_t ldy;
    int32_t sums_zeroed;            /* edtr_gn_stats only: nonzero = the caller has already zeroed `sums` (e.g. one edtr_zero_bytes
                                       over a pool of them), so no per-call memset node is enqueued */
    /* edtr_gn_apply only (optional): fold the per-tile column partials of the producing edtr_igemm (gn_partial,
     * tiles_per_image = H*W/128 — or H*W/64 where the split-K reducer wrote 64-row slots — <= 64 tiles per image) inside the apply launch itself — `sums` is then ignored and the
     * edtr_gn_finalize launch disappears (one launch less per GroupNorm of the UNet / ControlNet levels). */
    const float* partial; int32_t tiles_per_image;
} edtr_gn_params;

int edtr_gn_stats(const edtr_gn_params* p, edtr_stream_t stream);
/* Fold the per-tile column partials written by an edtr_igemm epilogue (gn_partial, tiles_per_image = H*W/128 tiles per
 * image, C columns) into sums[B][groups][2] — a drop-in replacement for edtr_gn_stats on that tensor. */
int edtr_gn_finalize(const float* partial, int tiles_per_image, int B, int C, int groups, double* sums,
                     edtr_stream_t stream);
int edtr_gn_apply(const edtr_gn_params* p, edtr_stream_t stream);
/* (scale, shift) = (gamma[c] rstd, beta[c] - mean gamma[c] rstd) per image and channel, table[B][C][2] fp32, from the tile partials
 * of the producing edtr_igemm (partial != NULL) or from the fp64 sums of edtr_gn_stats / edtr_gn_finalize: what edtr_gn_apply forms
 * per channel before it touches the tensor — for edtr_igemm's a_gn, which applies it while staging its operand (ABI 8). */
int edtr_gn_table(const float* partial, int tiles_per_image, const double* sums, int B, int C, int groups, int HW,
                  const float* gamma, const float* beta, float eps, float* table, edtr_stream_t stream);

/* LayerNorm over the last dimension, rows x C (C <= 2048, multiple of 8), fp32 math, 16-bit in/out.
 * c_valid (0 = C): only the first c_valid columns are real — the statistics run over them, and columns c_valid..C-1 of y are
 * written as zeros (SwinIR keeps its 180 channels in rows of 192 so that every GEMM has K % 64 == 0; gamma / beta hold C entries).
 * replaces: nn.LayerNorm, reference model/attention.py:222-224; model/swinir.py:208,215,753 (norm1 / norm2 / norm). */
int edtr_layernorm(int dtype, const void* x, int64_t rows, int C, int c_valid, int ldx, const float* gamma,
                   const float* beta, float eps, void* y, int ldy, edtr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Shifted-window multi-head attention of SwinIR, window 8 x 8 (64 tokens), head width <= 32, one launch per layer:
 * cyclic shift, window partition, q k^T * scale + relative-position bias + region mask, softmax, P v, window merge and
 * reverse shift.  One wavefront per (window, head); the 64 x 64 scores never leave registers.
 * replaces: torch.roll + window_partition + WindowAttention.forward (up to, not including, `proj`) + window_reverse +
 *           torch.roll, reference model/swinir.py:254-279 and :120-148.
 *   qkv    : [B*H*W][ld_qkv] 16-bit, token (b, y, x) at row (b*H + y)*W + x of the UNSHIFTED image; columns
 *            s*heads*32 + h*32 + e  (s = 0/1/2 for q/k/v, e < 32; e >= head_dim must be zero — the packed projection
 *            has zero weight rows there)
 *   out    : [B*H*W][ld_out] 16-bit; head h writes columns h*head_dim .. h*head_dim+head_dim-1, and columns
 *            heads*head_dim .. c_pad-1 are written as zeros
 *   bias   : fp32 [heads][64][64], bias[h][i][j] added to the score of query i and key j (the gathered
 *            relative_position_bias_table)
 *   labels : NULL when shift == 0; else uint8 [H][W], the image region of each pixel of the SHIFTED frame; a pair whose
 *            labels differ gets -100 added (the reference's attn_mask values, model/swinir.py:241)
 * ---------------------------------------------------------------------------------------- */
typedef struct edtr_window_attn_params {
    int32_t dtype;
    int32_t B, H, W;                /* token grid; H % 8 == 0, W % 8 == 0 */
    int32_t heads, head_dim;        /* head_dim even, <= 32 */
    int32_t shift;                  /* 0 <= shift < 8: window (wy, wx) token (ty, tx) is pixel ((8wy+ty+shift) % H, (8wx+tx+shift) % W) */
    const void* qkv; int32_t ld_qkv;
    void* out; int32_t ld_out; int32_t c_pad;
    const float* bias;
    const uint8_t* labels;
    float scale;
} edtr_window_attn_params;

int edtr_window_attn(const edtr_window_attn_params* p, edtr_stream_t stream);

/* The MLP half of a Swin layer in one launch (ABI 8):  out = x + fc2(GELU(fc1(LayerNorm(x)))) on 16-bit token rows.
 * replaces: `x = x + self.drop_path(self.mlp(self.norm2(x)))`, reference model/swinir.py:281-283 with Mlp.forward :31-37
 *           (nn.LayerNorm eps 1e-5, nn.GELU exact, both dropouts are p = 0).
 * Specialised for the shipped pre-restorer (configs/det/demo.yaml:2-18: embed_dim 180, mlp_ratio 2): C = 192 padded token
 * columns, hidden = 384 padded hidden units; anything else is EDTR_E_UNSUPPORTED and the caller issues the two edtr_igemm
 * launches instead.  Pad columns of x must be zero and stay zero (zero weight rows / bias entries).
 *   x      : [rows][ldx] 16-bit; LayerNorm statistics run over the first c_valid columns' stored values
 *   w1     : fc1 weights pre-multiplied by the LayerNorm gamma, rounded to `dtype`, as hidden/32 LDS images of 12288 bytes:
 *            image t, byte r*384 + ((c ^ ((r >> 1) & 7)) << 4) + 2 j  =  (gamma . W1)[32 t + r][8 c + j]    (r < 32, c < 24, j < 8)
 *   w2     : fc2 weights as hidden/32 images of 12288 bytes:
 *            image t, byte r*64 + ((c ^ ((r >> 2) & 3)) << 4) + 2 j   =  W2[r][32 t + 8 c + j]                (r < 192, c < 4, j < 8)
 *            (the XOR keys make every ds_read_b128 of an MFMA operand conflict-free; the images are copied by LDS-DMA as they are)
 *   c1     : fp32 [hidden], row sums of the PACKED 16-bit (gamma . W1)    c2b : fp32 [hidden], W1 beta + fc1 bias
 *            pre[u] = rstd (acc[u] - mean c1[u]) + c2b[u]   — the folded LayerNorm of edtr_igemm's ln_stats, with the row
 *            statistics taken inside the kernel from the x rows it already holds
 *   b2     : fp32 [C] fc2 bias (pad entries zero)
 *   out    : [rows][ldo] 16-bit (may not alias x)
 *   row_stats : optional fp32 [rows][C/32][2], the (sum, sum of squares) slots of the stored output rows in edtr_igemm's
 *            row_stats format (slots 0 and 3 carry the sums of columns 0..95 / 96..191, the others are zero), for the
 *            LayerNorm folded into the next layer's qkv projection.
 * ldx % 8 == 0, ldo % 8 == 0, all pointers 16-byte aligned. */
typedef struct edtr_swin_mlp_params {
    int32_t dtype;
    int32_t rows, C, hidden, c_valid;
    float eps;
    const void* x; int32_t ldx;
    const void* w1; const void* w2;
    const float* c1; const float* c2b; const float* b2;
    void* out; int32_t ldo;
    float* row_stats;
} edtr_swin_mlp_params;

int edtr_swin_mlp(const edtr_swin_mlp_params* p, edtr_stream_t stream);

/* The attention half of a Swin layer in one launch (ABI 8):  out = x + proj(WindowAttention(LayerNorm(x))) on 16-bit token rows.
 * replaces: `shortcut + window_reverse(attn(window_partition(roll(norm1(x)))))`, reference model/swinir.py:254-279 with
 *           WindowAttention.forward :120-148 (qkv linear, q * scale, relative-position bias, shift mask, softmax, proj linear; the
 *           dropouts are p = 0) — i.e. the qkv edtr_igemm, edtr_window_attn and the proj edtr_igemm of the three-launch form.
 * Specialised for the shipped pre-restorer (embed_dim 180 -> C = 192 padded columns, 6 heads of width 30 -> 32, window 8); anything
 * else is EDTR_E_UNSUPPORTED and the caller issues the three launches.  Pad columns of x must be zero and stay zero.
 *   x, out : [B*H*W][ld] 16-bit, token (b, y, x) at row (b*H + y)*W + x; out must not alias x (workgroups gather rows others scatter)
 *   wqkv   : heads * 3 LDS images of 12288 bytes, image 3 h + s (s = 0 / 1 / 2: q / k / v), layout as edtr_swin_mlp's w1 images:
 *            byte r*384 + ((c ^ ((r >> 1) & 7)) << 4) + 2 j = (gamma . Wqkv)[s*C' + h*d + r][8 c + j] for r < d, zero rows above
 *            (C' = heads*d, d = head_dim); the q rows are pre-multiplied by the softmax scale d^-1/2
 *   wproj  : heads images of 12288 bytes, image h, layout as edtr_swin_mlp's w2 images:
 *            byte r*64 + ((c ^ ((r >> 2) & 3)) << 4) + 2 j = Wproj[r][h*d + 8 c + j]   (zero for 8 c + j >= d, r >= C')
 *   c1, c2b: fp32 [3][heads][32]: row sums of the packed 16-bit (gamma . Wqkv) rows / Wqkv beta + bias (q entries scaled), zero pads
 *   bproj  : fp32 [C] (pad entries zero)
 *   bias   : fp32 [heads][16][64][4]: bias[h][kg][i][e] is added to the score of query i and key 4 kg + e (the gathered
 *            relative_position_bias_table, key-group major so that the 32 queries of a wave read contiguous 16-byte entries)
 *   labels : as edtr_window_attn (NULL when shift == 0)
 * LayerNorm statistics come from the gathered rows themselves (c_valid real columns, eps).  ld % 8 == 0, W % 4 == 0, pointers
 * 16-byte aligned. */
typedef struct edtr_swin_attn_params {
    int32_t dtype;
    int32_t B, H, W;                /* token grid; H % 8 == 0, W % 8 == 0 */
    int32_t heads, head_dim, shift;
    int32_t C, c_valid;
    float eps;
    const void* x; int32_t ldx;
    const void* wqkv; const void* wproj;
    const float* c1; const float* c2b; const float* bproj;
    const float* bias;
    const uint8_t* labels;
    void* out; int32_t ldo;
} edtr_swin_attn_params;

int edtr_swin_attn(const edtr_swin_attn_params* p, edtr_stream_t stream);

/* A whole Swin layer in one launch (ABI 8): edtr_swin_attn followed by edtr_swin_mlp on the token tile the attention half leaves in
 * LDS — `attn->out` receives x + attention half + MLP half; `mlp->x`, `->out`, `->rows`, `->row_stats` and the ld fields are ignored
 * (the tile never leaves the workgroup between the halves), its weight / constant operands are those of edtr_swin_mlp.  Same
 * shape and alignment rules as the two entry points.  replaces: SwinTransformerBlock.forward, reference model/swinir.py:254-283. */
int edtr_swin_layer(const edtr_swin_attn_params* attn, const edtr_swin_mlp_params* mlp, edtr_stream_t stream);

/* The feed-forward half of a BasicTransformerBlock in one launch (ABI 10):  out = x + W2 GEGLU(W1 LayerNorm(x) + b1) + b2.
 * replaces: `x = self.ff(self.norm3(x)) + x`, reference model/attention.py:233 with FeedForward :30-47 / GEGLU :20-27
 *           (nn.LayerNorm eps 1e-5, exact GELU, Dropout p = 0) — i.e. the LayerNorm launch (or fold) and the two edtr_igemm
 *           launches `ff.geglu` / `ff.out` of the two-launch form; the (rows, 4 d) hidden tensor never reaches memory.
 * Built for the 64 x 64-latent level of the SD-2.1 UNet / ControlNet: D = 320, H = 4 D = 1280, M % 128 == 0; anything else is
 * EDTR_E_UNSUPPORTED (edtr_ffn_plan answers without a launch) and the caller issues the two edtr_igemm launches.
 *   x   : [M][ldx] 16-bit RAW rows (before the LayerNorm): the kernel normalises them in registers (two-pass statistics over the
 *         stored values, (x - mean) rstd rounded to 16 bits) and reads them again for the residual
 *   w1  : [2 H][D] 16-bit, the GEGLU projection pre-multiplied by the LayerNorm gamma, value / gate rows interleaved in blocks of
 *         32 exactly as edtr_igemm's EDTR_ACT_GEGLU operand (rows 64 J .. 64 J + 31 = values of gated units 32 J .., the next 32
 *         rows their gates)
 *   w2  : [D][H] 16-bit, the output projection with its COLUMNS permuted inside every aligned group of 16:
 *         stored column 16 g + i = original column 16 g + {0,1,2,3, 8,9,10,11, 4,5,6,7, 12,13,14,15}[i]
 *         (the accumulator registers of the first product are then the operand of the second without a shuffle)
 *   cst : fp32 [H / 64][2][64]: the per-unit constant (W1 beta + b1)[R] of chunk c, half hh, at (q*2 + vg)*8 + lh*4 + e  for the
 *         packed w1 row R = 128 c + 64 hh + 32 vg + (e + 8 q + 4 lh)   (q, e < 4; vg = 0 value / 1 gate; lh < 2); the GATE
 *         entries (vg = 1) are stored HALVED (the kernel evaluates gelu from gate / 2)
 *   b2  : fp32 [D]
 *   out : [M][ldo] 16-bit, must not alias x (rows are re-read as the residual)
 * ldx % 8 == 0, ldo % 8 == 0, all pointers 16-byte aligned. */
typedef struct edtr_ffn_params {
    int32_t dtype;
    int32_t M, D, H;
    float eps;
    const void* x; int32_t ldx;
    const void* w1; const void* w2;
    const float* cst; const float* b2;
    void* out; int32_t ldo;
} edtr_ffn_params;

int edtr_ffn(const edtr_ffn_params* p, edtr_stream_t stream);
int edtr_ffn_plan(const edtr_ffn_params* p);      /* no HIP call: EDTR_OK or the error edtr_ffn would return */

/* The K = 320 linear layers of the 64 x 64-latent transformer blocks as a row-resident product, optionally with the LayerNorm in front
 * (ABI 10, lin320.hip):
 *     out[m][n] = alpha * sum_k xhat[m][k] w[n][k] + cvec[n] (+ residual[m][n]),   xhat = x, or (x - mean_m) rstd_m rounded to 16 bits
 * replaces: `attn2.to_q(norm2(x))` (model/attention.py:171, 224-232: ln = 1, alpha = the softmax prescale), `to_out[0](o) + x`
 * (:195), `proj_in` / `proj_out(x) + x_in` (:283-302) — an edtr_layernorm launch + an edtr_igemm launch, or one edtr_igemm launch.
 * A wave keeps 32 token rows in registers (loaded once, normalised in place), LDS holds only the weight stream.
 *   x        [M][ldx] 16-bit rows, K = 320 valid columns
 *   w        the [N][320] weight matrix in FRAGMENT order: 16 bytes per (chunk c of 32 output columns, k-step s of 16, lane l) at
 *            ((c * 20 + s) * 64 + l) * 16 = w[32 c + (l & 31)][16 s + 8 (l >> 5) .. + 7] (ops.pack_lin320_w); with ln != 0 the caller
 *            has folded the LayerNorm's gamma into the columns and put alpha * (W beta) (+ bias) into cvec
 *   cvec     fp32 [N] or NULL; residual: 16-bit [M][ldr] or NULL; out: 16-bit [M][ldo], != x
 *   vt_out   optional (the fused [Wq; Wk; Wv] projection of a self-attention, model/attention.py:170-178): the columns n >= vt_col0 are
 *            not stored at out[m][n] but TRANSPOSED and scaled by vt_alpha instead of alpha, exactly as edtr_igemm's vt_out:
 *            vt_out[(m / rows_per_image) * (N - vt_col0) + (n - vt_col0)][m % rows_per_image], row stride vt_ld, 16-bit (cvec applies to
 *            all columns).  Needs vt_col0 % 64 == 0, rows_per_image % 32 == 0, M % rows_per_image == 0, no residual; ldo >= vt_col0.
 *   gn_table optional (`proj_in(norm(x))`, model/attention.py:283-296): fp32 [M / rows_per_image][320][2] = (scale, shift) per image and
 *            channel, as edtr_gn_table writes it; the rows become x * scale + shift rounded to 16 bits in registers — the edtr_gn_apply launch
 *            in front of the projection is gone.  Not together with ln; rows_per_image % 128 == 0, M % rows_per_image == 0.
 * Needs K == 320, M % 128 == 0, N % 64 == 0, N <= 1024; anything else is EDTR_E_UNSUPPORTED (edtr_lin320_plan answers without a
 * launch) and the caller issues the edtr_igemm form. */
typedef struct edtr_lin320_params {
    int32_t dtype;
    int32_t M, N, K;
    int32_t ln; float eps;
    float alpha;
    const void* x; int32_t ldx;
    const void* w;
    const float* cvec;
    const void* residual; int32_t ldr;
    void* out; int32_t ldo;
    void* vt_out; int32_t vt_col0; int32_t vt_ld; float vt_alpha; int32_t rows_per_image;
    const float* gn_table;
} edtr_lin320_params;

int edtr_lin320(const edtr_lin320_params* p, edtr_stream_t stream);
int edtr_lin320_plan(const edtr_lin320_params* p);      /* no HIP call: EDTR_OK or the error edtr_lin320 would return */

/* 3 x 3 / stride 1 / pad 1 convolution of a 64-channel NHWC image into <= 64 channels, persistent workgroups with the nine tap
 * matrices resident in LDS (ABI 8) — SwinIR's reconstruction tail at the pixel levels, where edtr_igemm's 128-column tiles are
 * half padding.  replaces: conv_up1 / conv_up2 / conv_up3 (behind `F.interpolate(scale_factor=2, mode="nearest")`), conv_hr and
 * conv_last with their LeakyReLUs and the `x / img_range + mean` of the output, reference model/swinir.py:878-886.
 *   x      : [B][SH][SW][ldx] 16-bit, 64 channels read; source size (SH, SW) = (H, W), or (H / 2, W / 2) when upsample2x != 0
 *            (output pixel (Y, X) then reads source pixel (Y >> 1, X >> 1) — the nearest-neighbour upsample is never stored)
 *   w      : 9 LDS images of 8192 bytes, tap t = 3 ky + kx: byte n*128 + ((c ^ ((n >> 1) & 7)) << 4) + 2 j = weight[n][8 c + j][ky][kx]
 *            (n < 64 output channels, zero rows above the real count; 64 input channels)
 *   bias   : fp32 [64];   out = act(alpha * conv + bias), act = EDTR_ACT_NONE or EDTR_ACT_LRELU (act_slope)
 *   out    : 16-bit [B][H][W][ldo] (64 channels written), or — out_nchw_f32 != 0 — fp32 [B][n_valid][H][W] (n_valid <= 4)
 * H % 16 == 0, W % 16 == 0; ldx % 8 == 0, ldo % 8 == 0; pointers 16-byte aligned. */
typedef struct edtr_conv64_params {
    int32_t dtype;
    int32_t B, H, W;
    int32_t upsample2x;
    const void* x; int32_t ldx;
    const void* w;
    const float* bias;
    int32_t act; float act_slope; float alpha;
    void* out; int32_t ldo;
    int32_t out_nchw_f32, n_valid;
} edtr_conv64_params;

int edtr_conv64(const edtr_conv64_params* p, edtr_stream_t stream);

/* The VAE decoder's last step in one launch (ABI 8): GroupNorm + SiLU + 3 x 3 / stride 1 / pad 1 convolution of a 128-channel NHWC
 * image into <= 4 channels, written as fp32 NCHW planes.  replaces: `self.conv_out(nonlinearity(self.norm_out(h)))`, reference
 * model/vae.py:553-560, and the NHWC -> NCHW of the result (model/cldm.py:136-156 returns NCHW).
 *   x        : [B][H][W][ldx] 16-bit, 128 channels read
 *   gn_table : fp32 [B][128][2] from edtr_gn_table (NULL: no normalisation, a plain convolution)
 *   w        : 9 LDS images of 8192 bytes, tap t = 3 ky + kx: byte n*256 + ((c ^ (n & 15)) << 4) + 2 j = weight[n][8 c + j][ky][kx]
 *              (n < 32 rows, zero above the real output channels; 128 input channels)
 *   bias     : fp32 [32];   out = alpha * conv + bias -> fp32 [B][n_valid][H][W]
 * H % 16 == 0, W % 16 == 0, ldx % 8 == 0, n_valid <= 4; pointers 16-byte aligned. */
typedef struct edtr_conv128_out_params {
    int32_t dtype;
    int32_t B, H, W;
    const void* x; int32_t ldx;
    const float* gn_table;
    const void* w;
    const float* bias;
    float alpha;
    void* out; int32_t n_valid;
} edtr_conv128_out_params;

int edtr_conv128_out(const edtr_conv128_out_params* p, edtr_stream_t stream);

/* Row softmax: fp32 scores [rows][cols] (ld_s) -> 16-bit probabilities [rows][cols] (ld_p); columns cols..cols_pad-1
 * of every output row are written as zeros (so the row can feed a GEMM whose K is padded to a multiple of 8).
 * replaces: the softmax inside F.scaled_dot_product_attention of the d=512 single-head VAE
 *           attention, reference model/vae.py:298. */
int edtr_softmax_rows(int dtype, const float* s, int64_t rows, int cols, int64_t ld_s, void* p,
                      int64_t ld_p, int cols_pad, edtr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layout / elementwise helpers.
 * ---------------------------------------------------------------------------------------- */
/* Token + positional embedding of the CLIP text tower: out[row][:] = table[tokens[row]][:] + pos[row % L][:], 16-bit out
 * (rows = B*L, D % 8 == 0).  replaces: `self.model.token_embedding(text) + self.model.positional_embedding`, reference
 * model/clip.py:41-43. */
int edtr_embed_tokens(int dtype, const int64_t* tokens, const float* table, const float* pos, int rows, int L, int D,
                      int vocab, void* out, int ld, edtr_stream_t stream);
/* Zero `bytes` bytes (a multiple of 16, 16-byte aligned) with ONE kernel launch (hipMemsetAsync costs two tiny kernels per
 * node inside a hipGraph).  replaces: the implicit zero-initialisation of the reduction buffers that torch's
 * native_group_norm allocates per call (reference model/util.py:146-163 via nn.GroupNorm). */
int edtr_zero_bytes(void* ptr, int64_t bytes, edtr_stream_t stream);
/* Pixel-unshuffle front end of SwinIR: NCHW fp32 image [B][C][H][W] -> NHWC 16-bit tokens [B*(H/r)*(W/r)][ld] with
 * dst[row(b, y, x)][c*r*r + dy*r + dx] = (src[b][c][y*r+dy][x*r+dx] - sub[c]) * scale   (sub == NULL: 0);
 * columns C*r*r .. zero_pad_to-1 are written as zeros.  r in 1..8, H % r == 0, W % r == 0.
 * replaces: `(x - self.mean) * self.img_range` and nn.PixelUnshuffle (channel order c*r*r + dy*r + dx), reference
 * model/swinir.py:861 and :700-704. */
int edtr_pixel_unshuffle(int dtype, const float* src, int B, int C, int H, int W, int r, const float* sub, float scale,
                         void* dst, int ld, int zero_pad_to, edtr_stream_t stream);
/* NCHW fp32 [B][C][HW] -> NHWC 16-bit: dst[(b*HW+p)*ld + coff + c] = scale*src + shift; when
 * zero_pad_to > C the channels C..zero_pad_to-1 (relative to coff) are written as 0.
 * replaces: `.type(self.dtype)` + rearranges (model/controlnet.py:266-269; model/attention.py:292)
 * and the `*2-1` / `/scale_factor` scalings at the callers (demo.py:102; model/cldm.py:156). */
int edtr_nchw_to_nhwc(int dtype, const float* src, int B, int C, int64_t HW, void* dst, int ld,
                      int coff, int zero_pad_to, float scale, float shift, edtr_stream_t stream);
/* NHWC (16-bit, or fp32 when src_f32) -> NCHW fp32, first C channels, times scale. */
int edtr_nhwc_to_nchw(int dtype, const void* src, int src_f32, int B, int C, int64_t HW, int ld,
                      float* dst, float scale, edtr_stream_t stream);
/* out[r][c] = a[r][c] + b[r][c] over rows x C 16-bit elements with row strides lda/ldb/ldo (b == NULL: strided copy).
 * Writing straight into a column slice of a wider buffer is how the channel concat is made for free.
 * replaces: `hs.pop() + control.pop()`, `h += control.pop()` and the torch.cat of model/controlnet.py:31,35,37. */
int edtr_add(int dtype, const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows, int C,
             edtr_stream_t stream);
/* edtr_add on 16-bit tensors that also writes the per-slot column statistics of its result in edtr_igemm's gn_partial format (ABI 10):
 * gn_partial[(s * gn_ld + c) * 2 + {0, 1}] = sum / sum of squares over the slot's slot_rows rows (128, or 64 for 8 x 8 images) of column c,
 * of the fp32 sums before the 16-bit store — the skip + control half of the UNet decoder's concatenations (reference
 * model/controlnet.py:35-37), whose GroupNorm then needs no pass over the concatenated tensor (edtr_gn_stats).  rows % slot_rows == 0,
 * C % 32 == 0, gn_ld >= C; otherwise as edtr_add (b may be NULL: a copy). */
int edtr_add_stats(int dtype, const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows, int C,
                   float* gn_partial, int gn_ld, int slot_rows, edtr_stream_t stream);
/* The fp32-stream form of edtr_add that also writes the result's fp16 MIRROR out16[r * ld16 + c] (ABI 7; see edtr_igemm_params.out16):
 * the `hs.pop() + control.pop()` half of a decoder concat whose 1x1 skip convolution reads the mirror (model/controlnet.py:35-37,
 * model/unet.py:189).  C % 8 == 0. */
int edtr_add_mirror(const float* a, int lda, const float* b, int ldb, float* out, int ldo, void* out16, int ld16, int64_t rows, int C,
                    edtr_stream_t stream);
/* Sinusoidal timestep embedding [cos | sin] written as 16-bit rows of `dim` (even).
 * replaces: timestep_embedding, reference model/util.py:98-118. */
int edtr_timestep_embedding(int dtype, const int64_t* t, int B, int dim, void* out, int ld,
                            edtr_stream_t stream);
/* Spaced-sampler update on fp32 NCHW latents, n elements:
 *   pred_x0 = c_recip*x - c_recipm1*eps;  x_prev = coef1*pred_x0 + coef2*x + sigma*noise
 * replaces: utils/sampler.py:160-164,150-153,199-203 (sigma = sqrt(var) * [index != 0]). */
int edtr_sampler_update(const float* x, const float* eps, const float* noise, float c_recip,
                        float c_recipm1, float coef1, float coef2, float sigma, float* x_prev,
                        float* pred_x0, int64_t n, edtr_stream_t stream);
/* The same update with the step index read on the DEVICE (no host round trip for a caller that follows the reference
 * signature literally and hands p_sample a GPU `index` tensor, utils/sampler.py:189-204,311-312):
 *   coefs[n_steps][5] fp32 rows = (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1,
 *   posterior_mean_coef2, sqrt(posterior_variance) * [i != 0]);  image b of B uses row index[b] (clamped into the table). */
int edtr_sampler_update_indexed(const float* x, const float* eps, const float* noise, const int64_t* index,
                                const float* coefs, int n_steps, float* x_prev, float* pred_x0, int B,
                                int64_t per_image, edtr_stream_t stream);
/* DiagonalGaussianDistribution.sample() of the VAE posterior on the quant_conv output (model/distributions.py:24-41,
 * model/cldm.py:131-132):  out[b][c][p] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise[b][c][p]) * scale  with
 * mean = moments[(b*HW+p)*ld + c], logvar = moments[(b*HW+p)*ld + C + c] (fp32 NHWC rows of 2*C valid columns);
 * noise == NULL gives mode() * scale.  out / noise: NCHW fp32 [B][C][HW]. */
int edtr_gaussian_sample(const float* moments, int ld, const float* noise, float* out, int B, int C, int64_t HW,
                         float scale, edtr_stream_t stream);
/* out = a*x + b*y (fp32).  replaces: Diffusion.q_sample, model/gaussian_diffusion.py:80-84. */
int edtr_axpby(const float* x, const float* y, float a, float b, float* out, int64_t n,
               edtr_stream_t stream);
/* Diffusion.q_sample with per-image timesteps read on the device (no host round trip for a GPU-resident `t`):
 *   out[b][i] = tab_a[t[b]] * x[b][i] + tab_b[t[b]] * noise[b][i],  tab_a = sqrt_alphas_cumprod, tab_b = sqrt_one_minus_alphas_cumprod
 * (fp32 tables of n_tab entries, t clamped into the table).  replaces: extract_into_tensor + the two multiplies and the add,
 * reference model/gaussian_diffusion.py:34-37,80-84 as called with a device `t` at demo.py:107-108, main/det/test_edtr.py:127-128. */
int edtr_q_sample(const float* x, const float* noise, const int64_t* t, const float* tab_a, const float* tab_b,
                  int n_tab, float* out, int B, int64_t per_image, edtr_stream_t stream);
/* bf16 split-3 GEMM operand of a [rows][C] matrix (high-precision mode, see EDTR_F32_SPLIT): dst[rows][3*C] =
 * [hi | lo | hi] (pattern 0, the activation side) or [hi | hi | lo] (pattern 1, the weight side of an activation x activation
 * product).  src_dtype: EDTR_F32_SPLIT = fp32 source, EDTR_BF16 / EDTR_F16 = 16-bit source.  C, ld_src, ld_dst multiples of 8. */
int edtr_split3(int src_dtype, const void* src, int64_t rows, int C, int64_t ld_src, int pattern, void* dst, int64_t ld_dst,
                edtr_stream_t stream);
/* The general operand writer: dst[rows][parts*C] in the format `op_fmt` (EDTR_F32_SPLIT = bf16 [hi|lo|hi]; EDTR_F32_H1/H2/H3 =
 * fp16 [x] / [hi|lo] / [hi|lo|hi]) from an fp32 (src_dtype = any fp32-stream code) or 16-bit source. */
int edtr_split_operand(int src_dtype, const void* src, int64_t rows, int C, int64_t ld_src, int op_fmt, void* dst,
                       int64_t ld_dst, edtr_stream_t stream);
/* fp32 [rows][C] -> 16-bit [rows][C] with independent row strides (high-precision mode: the fp16 q / k / v^T operands of
 * edtr_flash_attn64 are cut from fp32 projection outputs).  C, ld_dst multiples of 8; ld_src multiple of 4. */
int edtr_cast16(int dst_dtype, const float* src, int64_t rows, int C, int64_t ld_src, void* dst, int64_t ld_dst,
                edtr_stream_t stream);
/* Gaussian-weighted overlap-add of one latent tile (fp32 NCHW):
 *   out[b][c][hi+y][wi+x] += tile[b][c][y][x] * wts[y][x];  count[...] += wts[y][x]
 * replaces: utils/common.py:415-424 (make_tiled_fn accumulation). */
int edtr_tile_accumulate(const float* tile, const float* wts, float* out, float* count, int B,
                         int C, int H, int W, int th, int tw, int hi, int wi, edtr_stream_t stream);
/* out = num / den elementwise (fp32).  replaces: utils/common.py:425. */
int edtr_divide(const float* num, const float* den, float* out, int64_t n, edtr_stream_t stream);

/* Tiled VAE (reference utils/tilevae/tilevae.py:232-304, GroupNormParam): sums is [T][BG][2] fp64 holding, per tile,
 * the edtr_gn_stats result of each (image, group).  Replaces them IN PLACE by the pair that makes edtr_gn_apply use the
 * tile-pooled statistics: mean = sum_t weights[t]*mean_t, var = sum_t weights[t]*var_t (biased per-tile variances; the
 * between-tile spread of the means is ignored exactly as the reference does); counts[t] = elements per group of tile t. */
int edtr_gn_pool(double* sums, const float* weights, const float* counts, int T, int BG, edtr_stream_t stream);
/* dst[p][y][x] = src[p][y][x] for planes x rows x cols fp32 elements with independent plane / row strides (elements).
 * replaces: the tile slicing `z[:, :, y1:y2, x1:x2]` and crop_valid_region placement, utils/tilevae/tilevae.py:218-229,468,548. */
int edtr_copy3d_f32(const float* src, int64_t src_plane_stride, int64_t src_row_stride, float* dst,
                    int64_t dst_plane_stride, int64_t dst_row_stride, int planes, int rows, int cols,
                    edtr_stream_t stream);

/* One level of the wavelet colour fix on fp32 NCHW planes: low = blur3x3(in; dilation radius, replicate pad) and, when
 * high_accum != NULL, high_accum += in - low.  `low` must not alias `in`.
 * replaces: wavelet_blur / wavelet_decomposition, reference utils/common.py:99-133 (runs right after vae_decode). */
int edtr_wavelet_level(const float* in, float* low, float* high_accum, int planes, int H, int W, int radius,
                       edtr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * hipGraph capture of a launch sequence issued on `stream` (one denoise step, or a whole batch).
 * ---------------------------------------------------------------------------------------- */
int edtr_graph_begin(edtr_stream_t stream);
int edtr_graph_end(edtr_stream_t stream, void** graph_exec_out);
int edtr_graph_launch(void* graph_exec, edtr_stream_t stream);
int edtr_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* EDTR_HIP_H */
