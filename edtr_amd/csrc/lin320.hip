// edtr_lin320 (round 6): the K = 320 linear layers of the 64 x 64-latent transformer blocks as a ROW-RESIDENT product, optionally with
// the LayerNorm in front of them — reference model/attention.py:171 (to_q of norm2(x)), :195 (to_out), :283-302 (proj_in / proj_out).
//
// Why.  These launches (M = 32768 at batch 8, N = K = 320) are five K-tiles long: the implicit-GEMM tiles spend a third of a
// workgroup's life in the multiply loop, the rest in the pipeline ramp and the epilogue's row loop (20.8 us per launch, 0.32 PFLOP/s;
// DESIGN.md section 6), and a LayerNorm in front is a launch of its own that reads and writes the tensor once more.
// Here a wave owns 32 token rows for the kernel's life, IN REGISTERS: its 32 x 320 16-bit values are the 20 B-operand fragments of
//     O^T[col][token] = W . X^T            (v_mfma_f32_32x32x16: lane = token l31, k = 16 s + 8 lh .. + 7)
// loaded once, straight from memory, normalised in place where a LayerNorm is asked for (two-pass statistics, (x - mean) rstd rounded
// to 16 bits: what the LayerNorm launch hands its GEMM; gamma is folded into W's columns, W beta into the additive row — ops.py).
// LDS holds only the weight stream: chunks of 32 output columns = 20 fragments of 1 KiB in the order the MFMA reads them (the host
// packs W that way: a DMA instruction of one wave IS one fragment, a fragment read is lane x 16 bytes: conflict-free), double
// buffered, 5 DMAs per wave and chunk, one barrier per chunk.  Two partial accumulators per chunk (even / odd k-steps) keep the
// matrix pipe from waiting on its own result.  Two chunks make a 64-column group: the finished fp32 values cross a wave-private LDS
// tile so that the epilogue (alpha, additive row, 16-bit residual, ONE rounding) works row-major and every store instruction writes
// eight whole 128-byte lines.  256-thread workgroups of 128 tokens, 78 KiB of LDS: two per CU, i.e. two waves per SIMD that run
// each other's non-matrix phases under their MFMAs.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int LK = 320;                        // K: the width this kernel is built for
constexpr int LKS = LK / 16;                   // 20 k-steps
constexpr int LBM = 128;                       // tokens per workgroup (4 waves x 32)
constexpr int LCH = 32;                        // output columns per chunk
constexpr int LCHB = LCH * LK * 2;             // 20 KiB: one chunk of W as 20 fragments
constexpr int LPITCH = 272;                    // bytes per token row of the fp32 staging tile (64 columns + 16 bytes)
constexpr int LVPITCH = 144;                   // ... and per CHANNEL row (32 tokens + 16 bytes) where a group is stored transposed (V^T)
constexpr int LSTGB = 64 * LVPITCH;            // 9 KiB per wave (>= 32 x LPITCH)
constexpr int L_STG = 2 * LCHB;                // staging tiles: 4 waves x LSTGB
constexpr int L_CV = L_STG + 4 * LSTGB;        // the additive row (fp32, N <= 1024)
constexpr int L_LDS = L_CV + 4096;             // 80 KiB: two workgroups per CU
static_assert(LSTGB >= 32 * LPITCH && 2 * L_LDS <= 160 * 1024, "LDS budget");

template <int N> __device__ __forceinline__ void lwait() {
    static_assert(N == 0 || N == 4 || N == 5 || N == 9, "lwait");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

template <typename T, bool LN, bool RES, bool GN>
__global__ void __launch_bounds__(256, 2) lin320_kernel(const edtr_lin320_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * LBM + 32 * wave;              // this wave's first token
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int nchunk = p.N / LCH;

    // weight chunk c -> buffer c & 1: fragments 5 wave .. 5 wave + 4 are this wave's (lane x 16 bytes each: the packed order)
    const u32x4 srdw = make_srd(p.w);
    auto issue_w = [&](int c) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
            dma16_buf((uint32_t)(lane * 16), srdw, (uint32_t)(c * LCHB + (5 * wave + i) * 1024), lds0 + (c & 1) * LCHB + (5 * wave + i) * 1024);
    };

    // ---- prologue: the token rows into registers, the first weight chunk in flight, the additive row into LDS
    U4 xf[LKS];
    {
        const uint16_t* xr = static_cast<const uint16_t*>(p.x) + (int64_t)(m0 + l31) * p.ldx + 8 * lh;
#pragma unroll
        for (int s = 0; s < LKS; ++s) xf[s] = ldg16(xr + 16 * s);
    }
    issue_w(0);
    for (int i = tid; i < (p.N >> 2); i += 256) {
        f32x4 v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (p.cvec) v = *reinterpret_cast<const f32x4*>(p.cvec + 4 * i);
        *reinterpret_cast<f32x4*>(smem + L_CV + 16 * i) = v;
    }
    if constexpr (GN) {
        // GroupNorm of the rows in the registers: x <- x scale[image][k] + shift[image][k] rounded to 16 bits — the (scale, shift) table
        // edtr_gn_table writes and edtr_igemm's a_gn applies, i.e. what the edtr_gn_apply launch in front of this projection stored.
        // The workgroup's 128 rows lie in one image: its table row (2.5 KiB) crosses LDS (the staging tiles are idle until the first group).
        const int img = (blockIdx.x * LBM) / p.rows_per_image;
        if (tid < LK / 2) *reinterpret_cast<f32x4*>(smem + L_STG + 16 * tid) = *reinterpret_cast<const f32x4*>(p.gn_table + (int64_t)img * LK * 2 + 4 * tid);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int tbo = L_STG + 64 * lh;
#pragma unroll
        for (int i = 0; i < LKS; ++i) {
            // (the fragment and its table offset are made to depend on the previous fragment's result: left alone, hipcc reads all 80
            //  table vectors and unpacks all 160 values up front and spills 88 - 156 registers)
            if (i > 0) asm volatile("" : "+v"(tbo), "+v"(xf[i].x), "+v"(xf[i].y), "+v"(xf[i].z), "+v"(xf[i].w) : "v"(xf[i - 1].x));
            const char* tb = smem + tbo + 128 * i;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb), t1 = *reinterpret_cast<const f32x4*>(tb + 16);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(tb + 32), t3 = *reinterpret_cast<const f32x4*>(tb + 48);
            float f[8];
            unpack8<T>(xf[i], f);
            f[0] = f[0] * t0[0] + t0[1]; f[1] = f[1] * t0[2] + t0[3];
            f[2] = f[2] * t1[0] + t1[1]; f[3] = f[3] * t1[2] + t1[3];
            f[4] = f[4] * t2[0] + t2[1]; f[5] = f[5] * t2[2] + t2[3];
            f[6] = f[6] * t3[0] + t3[1]; f[7] = f[7] * t3[2] + t3[3];
            xf[i] = pack8<T>(f);
            asm volatile("" : "+v"(xf[i].x), "+v"(xf[i].y), "+v"(xf[i].z), "+v"(xf[i].w));      // (finished HERE, not sunk into the loop preheader)
        }
    }
    if constexpr (LN) {
        // LayerNorm of this lane's token in the registers: two-pass statistics over the stored 16-bit values (reference nn.LayerNorm,
        // eps as given), x <- (x - mean) rstd rounded to 16 bits
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LKS; ++i) {
            float f[8];
            unpack8<T>(xf[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[j];
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / LK);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < LKS; ++i) {
            float f[8];
            unpack8<T>(xf[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = f[j] - mean; q = __builtin_fmaf(dlt, dlt, q); }
        }
        q += __shfl_xor(q, 32, 64);
        const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / LK) + p.eps);
        const float shift = -mean * rstd;
#pragma unroll
        for (int i = 0; i < LKS; ++i) {
            float f[8];
            unpack8<T>(xf[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = __builtin_fmaf(f[j], rstd, shift);
            xf[i] = pack8<T>(f);
        }
    }

    char* const stg = smem + L_STG + wave * LSTGB;             // this wave's staging tile: [32 tokens][64 columns] fp32 (or [64 channels][32 tokens])
    const float alpha = p.alpha;
    // epilogue geometry: lane -> rows (lane >> 3) + 8 i, columns 8 (lane & 7) .. + 7 of the group
    const int erow = lane >> 3, ecol = 8 * (lane & 7);
    const uint16_t* resp = RES ? static_cast<const uint16_t*>(p.residual) + (int64_t)(m0 + erow) * p.ldr + ecol : nullptr;
    uint16_t* const outp = static_cast<uint16_t*>(p.out) + (int64_t)(m0 + erow) * p.ldo + ecol;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // own fragments of chunk 0, own part of the additive row
#pragma unroll 1
    for (int g = 0; g < (nchunk >> 1); ++g) {
        U4 rv[4];
        const bool tgroup = p.vt_out != nullptr && 64 * g >= p.vt_col0;      // (wave-uniform; RES launches have no transposed part)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int c = 2 * g + hh;
            // this wave's fragments of chunk c have landed; younger than them: the residual loads of this group (odd chunks), the four
            // stores of the previous group's epilogue (even chunks; none in front of chunk 0, which the prologue waited for outright)
            if (hh == 0) lwait<4>();
            else if (RES) lwait<4>();
            else lwait<0>();
            __builtin_amdgcn_s_barrier();                      // ... everyone's; chunk c - 1 is multiplied: its buffer is free
            asm volatile("" ::: "memory");                     // (a raw barrier: __syncthreads() would drain the residual loads and the DMAs)
            if (c + 1 < nchunk) issue_w(c + 1);
            if (hh == 0 && RES) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[i] = ldg16(resp + (int64_t)(8 * i) * p.ldr + 64 * g);
            }
            const char* wb = smem + (c & 1) * LCHB + lane * 16;
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < LKS; s += 2) {
                const U4 w0 = *reinterpret_cast<const U4*>(wb + s * 1024), w1 = *reinterpret_cast<const U4*>(wb + (s + 1) * 1024);
                a0 = T::mfma(w0, xf[s], a0);
                a1 = T::mfma(w1, xf[s + 1], a1);
            }
            // D[col][token]: register 4 j + e = column 8 j + 4 lh + e of the chunk, token l31 -> four consecutive fp32 of the token's row
            // (a transposed group: one fp32 per register into the column's own row of 32 tokens)
            if (!tgroup) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = a0[4 * j + e] + a1[4 * j + e];
                    *reinterpret_cast<f32x4*>(stg + l31 * LPITCH + (32 * hh + 8 * j + 4 * lh) * 4) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<float*>(stg + (32 * hh + 8 * j + 4 * lh + e) * LVPITCH + l31 * 4) = a0[4 * j + e] + a1[4 * j + e];
            }
        }
        if (tgroup) {
            // ---- a group of V: channel rows of this wave's 32 tokens -> vt_out[(image) * (N - vt_col0) + channel][token], 16 bytes
            // (8 tokens) per lane, four lanes per channel row (edtr_igemm's vt_out layout: what edtr_flash_attn64 reads)
            const int img = m0 / p.rows_per_image, tok0 = m0 - img * p.rows_per_image;
            uint16_t* const vb = static_cast<uint16_t*>(p.vt_out) + ((int64_t)img * (p.N - p.vt_col0) + (64 * g - p.vt_col0)) * p.vt_ld + tok0;
            const float va = p.vt_alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = (lane >> 2) + 16 * i, oct = lane & 3;
                const char* sp = stg + ch * LVPITCH + oct * 32;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 16);
                const float cc = reinterpret_cast<const float*>(smem + L_CV)[64 * g + ch];
                float f[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[e] = __builtin_fmaf(v0[e], va, cc); f[e + 4] = __builtin_fmaf(v1[e], va, cc); }
                stg16(vb + (int64_t)ch * p.vt_ld + 8 * oct, pack8<T>(f));
            }
            continue;
        }
        // ---- the group's 64 columns, row-major: alpha, additive row, residual, one rounding, whole-line stores
        const float* cv = reinterpret_cast<const float*>(smem + L_CV) + 64 * g + ecol;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cv), c1 = *reinterpret_cast<const f32x4*>(cv + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* sp = stg + (erow + 8 * i) * LPITCH + ecol * 4;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 16);
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = __builtin_fmaf(v0[e], alpha, c0[e]); f[e + 4] = __builtin_fmaf(v1[e], alpha, c1[e]); }
            if constexpr (RES) {
                float rf[8];
                unpack8<T>(rv[i], rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += rf[e];
            }
            stg16(outp + (int64_t)(8 * i) * p.ldo + 64 * g, pack8<T>(f));
        }
    }
}

template <typename T>
int launch_lin320(const edtr_lin320_params& p, hipStream_t stream) {
    static EdtrLdsOnce once[6];
    const int v = p.gn_table ? 4 + (p.residual ? 1 : 0) : (p.ln ? 2 : 0) | (p.residual ? 1 : 0);
    const dim3 grid((unsigned)(p.M / LBM)), block(256);
    auto go = [&](auto kern) -> int {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(kern), L_LDS, once[v])) return rc_;
        hipLaunchKernelGGL(kern, grid, block, L_LDS, stream, p);
        EDTR_LAUNCH_CHECK();
        return EDTR_OK;
    };
    switch (v) {
        case 0: return go(&lin320_kernel<T, false, false, false>);
        case 1: return go(&lin320_kernel<T, false, true, false>);
        case 2: return go(&lin320_kernel<T, true, false, false>);
        case 3: return go(&lin320_kernel<T, true, true, false>);
        case 4: return go(&lin320_kernel<T, false, false, true>);
        default: return go(&lin320_kernel<T, false, true, true>);
    }
}

int check_lin320(const edtr_lin320_params& p) {
    if (!p.x || !p.w || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.K != LK || p.M <= 0 || p.N <= 0) return EDTR_E_SHAPE;
    if ((p.M % LBM) || (p.N % 64) || p.N > 1024) return EDTR_E_UNSUPPORTED;
    if (p.ldx < LK || (p.residual && p.ldr < p.N)) return EDTR_E_SHAPE;
    if ((p.ldx & 7) || (p.ldo & 7) || (p.residual && (p.ldr & 7))) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.out) || (p.residual && !aligned16(p.residual)) || (p.cvec && !aligned16(p.cvec))) return EDTR_E_ALIGN;
    if (p.x == p.out) return EDTR_E_UNSUPPORTED;               // (another workgroup may still read the rows this one writes)
    if (p.gn_table) {    // a GroupNorm of the rows applied in registers: one image per wave, not together with the LayerNorm
        if (p.ln || p.rows_per_image <= 0 || (p.rows_per_image % LBM) || (p.M % p.rows_per_image)) return EDTR_E_UNSUPPORTED;
        if (!aligned16(p.gn_table)) return EDTR_E_ALIGN;
    }
    if (p.vt_out) {      // the columns from vt_col0 on leave transposed: whole 64-column groups, a wave's 32 rows inside one image
        if (p.residual || p.vt_col0 <= 0 || p.vt_col0 >= p.N || (p.vt_col0 & 63) || p.rows_per_image <= 0 || (p.rows_per_image & 31) ||
            (p.M % p.rows_per_image) || p.vt_ld < p.rows_per_image)
            return EDTR_E_UNSUPPORTED;
        if ((p.vt_ld & 7) || !aligned16(p.vt_out)) return EDTR_E_ALIGN;
        if (p.ldo < p.vt_col0) return EDTR_E_SHAPE;
    } else if (p.ldo < p.N) {
        return EDTR_E_SHAPE;
    }
    if ((int64_t)p.N * LK * 2 >= 0xF0000000LL) return EDTR_E_UNSUPPORTED;
    return EDTR_OK;
}

}  // namespace

extern "C" int edtr_lin320(const edtr_lin320_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    if (int e = check_lin320(*pp)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return pp->dtype == EDTR_BF16 ? launch_lin320<BF16>(*pp, s) : launch_lin320<F16>(*pp, s);
}

// every check of the launch, no HIP call: EDTR_OK or the error edtr_lin320 would return
extern "C" int edtr_lin320_plan(const edtr_lin320_params* pp) {
    if (!pp) return EDTR_E_NULL;
    return check_lin320(*pp);
}
