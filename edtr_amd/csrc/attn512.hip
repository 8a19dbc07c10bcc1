// Fused single-head attention with head width 512 for gfx950 (include/edtr_hip.h: edtr_flash_attn512) — the VAE's AttnBlock,
// reference model/vae.py:279-308: softmax(q k^T / sqrt(512)) v over the 64 x 64 (128 x 128, tile-local 40 x 40 ...) latent positions.
//
// Until round 5 this ran as QK^T GEMM -> fp32 score matrix in HBM -> edtr_softmax_rows -> PV GEMM: 4096^2 x 4 B x 8 images
// written, read, 2 B written, read per call (1.25 GB fetched + written by the softmax launch alone, profiles/r04/pmc_hbm_traffic.json)
// and a 1-GiB score matrix per image at 1024^2.  Here the scores never leave the CU.
//
// d = 512 does not fit the d = 64 kernels' plan (a wave would hold 32 queries x 512 channels of O = 256 registers): the
// workgroup's EIGHT waves split the two products differently and meet in LDS:
//   * O^T for 128 queries x 512 channels lives in the workgroup's accumulators (128 registers per lane): wave w owns channels
//     64 w .. 64 w + 63 of all 128 queries;
//   * S: wave w owns queries 16 w .. 16 w + 15; its Q rows stay in registers for the whole kernel (16 queries x 512 = 64
//     registers), the 32-key K tile is read from LDS: S^T = K Q^T, 32 x v_mfma_16x16x32 per tile (keys on the MFMA rows, K rows
//     placed in LDS in an order that leaves a lane with EIGHT CONSECUTIVE keys of its query = the B fragment of the next product);
//   * online softmax by the S owner (in-lane over 8 keys, two lane exchanges over the 4 lanes of a query), exp2 domain, running
//     maximum raised only when it grows by more than 2^8 (deferred rescale), P (16-bit) and the per-query rescale factor go to LDS;
//   * O^T += V^T P^T: 32 MFMAs per wave and tile (4 channel blocks x 8 query blocks), 12 fragment reads.
// K / V^T tiles (32 keys: 32 + 32 KiB) go global -> LDS by buffer-addressed LDS-DMA, double buffered, two barriers per tile.
// V^T rows sit in LDS in the permuted order of halo512.hip, so a lane ends with 16 consecutive output channels of its query.
// Work per call at B = 8, N = 4096: 275 GFLOP in 256 workgroups (one per CU); images map to XCDs, so an image's K / V^T (4 + 4 MiB)
// are served by one L2.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr uint32_t kOob = 0xFFFFFF00u;

__device__ __forceinline__ u32x4 srd_of(const void* base) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4 srd;
    srd.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    srd.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    srd.z = 0xFFFFFF00u;
    srd.w = 0x00020000u;
    return srd;
}

__device__ __forceinline__ void dma(uint32_t voff, const u32x4& srd, uint32_t soff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :
                 : "v"(voff), "s"(srd), "s"(lds_addr), "s"(soff)
                 : "memory");
}

// 2-bit XOR key of a 64-byte LDS row by row index mod 8 (halo512.hip: conflict-free ds_read_b128 of 16 consecutive 64-byte rows)
__device__ __forceinline__ int key8(int x) {
    constexpr uint32_t kKey = 3u | (3u << 2) | (0u << 4) | (1u << 6) | (0u << 8) | (1u << 10) | (3u << 12) | (2u << 14);
    return (int)((kKey >> (2 * (x & 7))) & 3u);
}

constexpr int BQ = 128, TK = 32, D = 512;
constexpr int KTILE = TK * D * 2, VTILE = D * TK * 2;      // 32 KiB each
constexpr int K_BASE = 0, V_BASE = 2 * KTILE, P_BASE = V_BASE + 2 * VTILE, A_BASE = P_BASE + BQ * TK * 2;
constexpr int LDS_BYTES = A_BASE + 1024;                    // 137 KiB
constexpr float kDeferLog2 = 8.0f;

template <typename T>
__global__ void __launch_bounds__(512, 1) flash_attn512_kernel(const edtr_attn_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int nqb = (p.Nq + BQ - 1) / BQ;
    int bid = blockIdx.x;
    {       // an XCD (blockIdx mod 8) owns a contiguous range of (image, query block) pairs: an image's K / V^T stay in one L2
        const int nblk = p.B * nqb, qq = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + j;
    }
    const int b = bid / nqb, q0 = (bid - b * nqb) * BQ;
    const int NT = p.Nk / TK;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const uint16_t* kp = static_cast<const uint16_t*>(p.k) + (int64_t)b * p.k_bs;
    const uint16_t* vp = static_cast<const uint16_t*>(p.vt) + (int64_t)b * p.vt_bs;
    const u32x4 srd_k = srd_of(kp), srd_v = srd_of(vp);

    // ---- staging geometry.  K piece = one LDS row L = wave + 8 j (1 KiB = one key, 64 chunks of 16 B): L = 16 kb + r holds key
    // 8 (r >> 2) + 4 kb + (r & 3) of the tile; LDS chunk i holds the key's chunk i ^ (L & 15).  V^T piece = 16 LDS rows of 64 B:
    // LDS row 64 ww + 16 cb + r holds channel 64 ww + 16 (r >> 2) + 4 cb + (r & 3), slot s holds key chunk s ^ key8(row).
    // Three lane offsets serve the eight pieces of a wave and tile (eight loop-invariant registers are the first thing the register
    // allocator spills here): K rows w and w + 8 differ in their swizzle (one offset each), rows L and L + 16 only by four keys (a
    // scalar offset); the V^T pieces of a wave are the same 16 rows 128 channels apart (scalar).
    uint32_t voff_k[2], voff_v;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = wave + 8 * h;                          // LDS row r (kb = 0) and r + 16 (kb = 1): keys 8 (r >> 2) + (r & 3) (+ 4)
        voff_k[h] = (uint32_t)(((8 * (r >> 2) + (r & 3)) * p.k_ld + ((lane ^ r) * 8)) * 2);
    }
    {
        const int Lr = wave * 16 + (lane >> 2), slot = lane & 3;          // piece `wave` (j = 0); piece wave + 8 j = 128 j channels on
        const int rr = Lr & 15, cb = (Lr >> 4) & 3;
        const int chan = (Lr & ~63) + 16 * (rr >> 2) + 4 * cb + (rr & 3);
        voff_v = (uint32_t)((chan * p.vt_ld + ((slot ^ key8(Lr)) * 8)) * 2);
    }
    // piece j of a wave and tile (0..3 K, 4..7 V^T): K piece j = LDS row (wave + 8 (j & 1)) + 16 (j >> 1)
    auto stage_piece = [&](int t, int buf, int j8) {
        const uint32_t sk = (uint32_t)(t * TK) * (uint32_t)p.k_ld * 2u, sv = (uint32_t)(t * TK * 2);
        const int j = j8 & 3;
        if (j8 < 4) dma(voff_k[j & 1], srd_k, sk + (uint32_t)((j >> 1) * 4 * p.k_ld * 2), smem_base + K_BASE + buf * KTILE + (wave + 8 * (j & 1) + 16 * (j >> 1)) * 1024);
        else dma(voff_v, srd_v, sv + (uint32_t)(128 * j) * (uint32_t)p.vt_ld * 2u, smem_base + V_BASE + buf * VTILE + (wave + 8 * j) * 1024);
    };
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) stage_piece(t, buf, j8);
    };
    stage(0, 0);

    // ---- this wave's Q rows: B fragments of S^T = K Q^T, lane (query l15, lq) holds d = 32 s + 8 lq .. + 7 for s = 0 .. 15
    U4 qf[16];
    {
        const int qrow = q0 + 16 * wave + l15;
        const uint16_t* qp = static_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qrow * p.q_ld + 8 * lq;
#pragma unroll
        for (int s = 0; s < 16; ++s) qf[s] = qrow < p.Nq ? ldg16(qp + 32 * s) : zero16();
    }
    // fragment read addresses (lane part; tile buffer, block and step are immediates)
    const int j4 = l15 >> 2;
    int koff[4];                  // K chunk (4 s + lq) ^ l15 = 4 (s ^ j4) | (lq ^ (l15 & 3)): the four values of (s & 3) ^ j4
#pragma unroll
    for (int m = 0; m < 4; ++m) koff[m] = K_BASE + l15 * 1024 + 64 * (m ^ j4) + 16 * (lq ^ (l15 & 3));
    const int swz = (lq ^ key8(l15)) << 4;
    const int v_rd = V_BASE + (64 * wave + l15) * 64 + swz;              // + buf VTILE + cb KiB
    const int p_rd = P_BASE + l15 * 64 + swz;                            // + qb8 KiB
    const int p_wr = P_BASE + (16 * wave + l15) * 64 + swz;
    const float sc = p.scale * 1.4426950408889634f;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -__builtin_inff(), l_run = 0.0f;

    auto tile = [&](int t, auto BUFc) {
        constexpr int BUF = decltype(BUFc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // tile t landed for every wave; every wave is past tile t - 1
        const bool more = t + 1 < NT;                       // tile t + 1's eight pieces are requested one per MFMA group of the S phase below: up
                                                            // front they cost ~500 cycles of DMA issue with no MFMA of any wave beside them
        // ---- S^T block of this wave: 32 keys x 16 queries over d = 512
        f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f}, s2 = {0.0f, 0.0f, 0.0f, 0.0f}, s3 = {0.0f, 0.0f, 0.0f, 0.0f};      // two chains per key block
#pragma unroll
        for (int sh = 0; sh < 8; ++sh) {                    // two steps at a time: 4 fragment reads in flight (8 would spill)
            U4 kf[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    kf[m][kb] = *reinterpret_cast<const U4*>(smem + koff[2 * (sh & 1) + m] + BUF * KTILE + kb * 16384 + (sh >> 1) * 256);
            if (more) stage_piece(t + 1, BUF ^ 1, sh);
            s0 = T::mfma16(kf[0][0], qf[2 * sh], s0);
            s1 = T::mfma16(kf[0][1], qf[2 * sh], s1);
            s2 = T::mfma16(kf[1][0], qf[2 * sh + 1], s2);
            s3 = T::mfma16(kf[1][1], qf[2 * sh + 1], s3);
        }
        s0 += s2;
        s1 += s3;
        // ---- online softmax of this lane's query over its 8 keys (8 lq .. 8 lq + 7), exp2 domain
        float sv[8] = {s0[0] * sc, s0[1] * sc, s0[2] * sc, s0[3] * sc, s1[0] * sc, s1[1] * sc, s1[2] * sc, s1[3] * sc};
        float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = tmax > m_run + kDeferLog2 ? tmax : m_run;     // (first tile: m_run = -inf)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);           // 1 when unchanged, 0 on the first tile
        float ps = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sv[i] = __builtin_amdgcn_exp2f(sv[i] - m_new); ps += sv[i]; }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        *reinterpret_cast<U4*>(smem + p_wr) = pack8<T>(sv);
        if (lq == 0) *reinterpret_cast<float*>(smem + A_BASE + (16 * wave + l15) * 4) = alpha;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // P and the rescale factors of all 128 queries are visible
        // ---- O^T (this wave's 64 channels x 128 queries) += V^T P^T
        bool any = false;
#pragma unroll
        for (int qb = 0; qb < 8; ++qb) any |= *reinterpret_cast<const float*>(smem + A_BASE + (16 * qb + l15) * 4) != 1.0f;
        U4 vf[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) vf[cb] = *reinterpret_cast<const U4*>(smem + v_rd + BUF * VTILE + cb * 1024);
        if (__any(any)) {            // rare (the running maximum moved by more than 2^8 somewhere): the factors are read again here, not kept
#pragma unroll
            for (int qb = 0; qb < 8; ++qb) {
                const float al = *reinterpret_cast<const float*>(smem + A_BASE + (16 * qb + l15) * 4);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb][qb] *= al;
            }
        }
#pragma unroll
        for (int qh = 0; qh < 4; ++qh) {                    // two query blocks at a time
            U4 pf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) pf[i] = *reinterpret_cast<const U4*>(smem + p_rd + (2 * qh + i) * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb][2 * qh + i] = T::mfma16(vf[cb], pf[i], acc[cb][2 * qh + i]);
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < NT) tile(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: 1 / row sum through LDS, then O rows: lane (query l15 of block qb, lq) holds channels 64 w + 16 lq + 4 cb + i
    // (lane-derived values of the epilogue come from an opaque copy of the lane index: computed at kernel entry they are carried —
    //  spilled — through the tile loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15e = lane_e & 15, lqe = lane_e >> 4;
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __builtin_amdgcn_s_barrier();                           // every wave is past its last read of the rescale factors
    if (lqe == 0) *reinterpret_cast<float*>(smem + A_BASE + (16 * wave + l15e) * 4) = 1.0f / l_run;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int qb = 0; qb < 8; ++qb) {
        const float inv = *reinterpret_cast<const float*>(smem + A_BASE + (16 * qb + l15e) * 4);
        const int row = q0 + 16 * qb + l15e;
        float f[16];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[4 * cb + i] = acc[cb][qb][i] * inv;
        if (row < p.Nq) {
            const int64_t oidx = (int64_t)b * p.o_bs + (int64_t)row * p.o_ld + 64 * wave + 16 * lqe;
            if (p.out_f32) {
                float* o = static_cast<float*>(p.out) + oidx;
#pragma unroll
                for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(o + 4 * v) = f32x4{f[4 * v], f[4 * v + 1], f[4 * v + 2], f[4 * v + 3]};
            } else {
                uint16_t* o = static_cast<uint16_t*>(p.out) + oidx;
                float g0[8], g1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { g0[i] = f[i]; g1[i] = f[8 + i]; }
                stg16(o, pack8<T>(g0));
                stg16(o + 8, pack8<T>(g1));
            }
        }
    }
}

template <typename T>
int launch512(const edtr_attn_params& p, hipStream_t s) {
    static EdtrLdsOnce once;
    if (int rc = edtr_lds_attr(reinterpret_cast<const void*>(&flash_attn512_kernel<T>), LDS_BYTES, once)) return rc;
    const int nqb = (p.Nq + BQ - 1) / BQ;
    hipLaunchKernelGGL(flash_attn512_kernel<T>, dim3((unsigned)(p.B * nqb)), dim3(512), LDS_BYTES, s, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

}  // namespace

extern "C" int edtr_flash_attn512(const edtr_attn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_attn_params& p = *pp;
    if (!p.q || !p.k || !p.vt || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H != 1 || p.Nq <= 0 || p.Nk <= 0) return EDTR_E_SHAPE;
    if (p.Nk % TK) return EDTR_E_UNSUPPORTED;                 // whole 32-key tiles (the caller keeps the three-launch form otherwise)
    if (p.causal || p.q_prescaled || p.q_lo || p.k_lo || p.vt_lo) return EDTR_E_UNSUPPORTED;
    if ((p.q_ld & 7) || (p.k_ld & 7) || (p.vt_ld & 7) || (p.o_ld & (p.out_f32 ? 3 : 7)) || p.q_ld < D || p.k_ld < D || p.vt_ld < p.Nk || p.o_ld < D)
        return EDTR_E_ALIGN;
    if ((p.q_bs & 7) || (p.k_bs & 7) || (p.vt_bs & 7) || (p.o_bs & (p.out_f32 ? 3 : 7))) return EDTR_E_ALIGN;
    if (!aligned16(p.q) || !aligned16(p.k) || !aligned16(p.vt) || !aligned16(p.out)) return EDTR_E_ALIGN;
    // 32-bit byte offsets inside one image's K rows / V^T rows (buffer-addressed DMA)
    if ((int64_t)p.Nk * p.k_ld * 2 >= 0xF0000000LL || (int64_t)D * p.vt_ld * 2 >= 0xF0000000LL) return EDTR_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == EDTR_BF16 ? launch512<BF16>(p, s) : launch512<F16>(p, s);
}
