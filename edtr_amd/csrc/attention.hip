// Fused attention, head width 64, for gfx950 (see include/edtr_hip.h: edtr_flash_attn64).
//
// One workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.
// Per 64-key tile a wave computes S^T = K Q^T (keys on the MFMA rows, queries on the lanes), so a lane holds 32 of
// the 64 scores of ITS query in registers: the row max / row sum are in-lane reductions plus one exchange with
// lane^32 (wavefront shuffle), never through LDS.  The probabilities are converted to 16-bit in registers (one
// v_cvt_pk per pair) and fed straight back as the B operand of O^T += V^T P^T ("accumulator tile as the next MFMA's
// operand"): K rows are read from LDS in an order with bits 2,3 of the row swapped, which makes registers 8s..8s+7
// of a score accumulator exactly keys 16s+8h..16s+8h+7 — the natural k order of the V^T fragment (one ds_read_b128).
// V arrives key-major (V^T, written that way by the projection GEMM), so no transposed read is needed.
//
// K / V^T tiles go global -> LDS by buffer-addressed LDS-DMA (no VGPR staging, no ds_write; keys >= Nk fail the
// buffer range check and land as zeros), double buffered: tile t+1 is in flight while tile t is consumed
// (counted s_waitcnt vmcnt + raw s_barrier).  The softmax is VALU-bound at d = 64, so the instruction count per
// score is kept minimal: exp2 domain (one FMA + one v_exp per score), v_max3 reductions, and the running max is only
// raised when it grows by more than 2^8 (deferred rescale: O and l are rescaled on <1 % of the tiles).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int KV = 64;            // keys per tile
constexpr int TILE_BYTES = 64 * 64 * 2;
constexpr float kDeferLog2 = 8.0f;   // P <= 2^8 before a rescale is forced (exact in bf16/fp16: only the exponent grows)

constexpr uint32_t kOob = 0xFFFFFF00u;

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__device__ __forceinline__ u32x4 srd_of(const void* base) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4 srd;
    srd.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    srd.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    srd.z = 0xFFFFFF00u;
    srd.w = 0x00020000u;
    return srd;
}

__device__ __forceinline__ void dma16(uint32_t voff, const u32x4& srd, uint32_t soff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(srd), "s"(lds_addr), "s"(soff)
                 : "memory");
}

// ---- coalesced store of a wave's O tile [32 queries][64 channels] (round 4) ------------------------------------------------
// The accumulator layout of O^T = V^T P^T gives lane (query l31, half lh) the channels db*32 + 8g + 4*lh + (0..3): written straight
// to memory that is 8 stores of 8 bytes per lane, each instruction touching 32 different rows with two adjacent 8-byte pieces
// per row.  Measured on the cross-attention launches (tools/exp/r04_attn_shapes.py with the stores compiled out): 7 of 18 us at
// 64x64 latents, 3 of 10.7 at 32x32 — the L2 sees 16-byte partial-line writes at its request rate, not 21 MB at its bandwidth.
// Here the wave transposes the tile through a private 4.5 KiB LDS slab (144-byte rows: 16-byte aligned for the read-back) and
// stores 16 bytes per lane with 8 lanes covering one row's 128 contiguous bytes: 4 store instructions of 8 whole row segments each.
constexpr int OPITCH = 144, OSLAB = 32 * OPITCH;

template <typename T, typename ACC>
__device__ __forceinline__ void store_o_tile(char* slab, const ACC& o0, const ACC& o1, float inv, uint16_t* out_rows, int64_t o_ld,
                                             int q0, int Nq, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint2 v;
        v.x = pack2<T>(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv);
        v.y = pack2<T>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(slab + l31 * OPITCH + (8 * g + 4 * lh) * 2) = v;
        v.x = pack2<T>(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv);
        v.y = pack2<T>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(slab + l31 * OPITCH + (32 + 8 * g + 4 * lh) * 2) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // same wave: the LDS queue keeps the order, the compiler must too
    const int r8 = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + r8;
        const U4 v = *reinterpret_cast<const U4*>(slab + row * OPITCH + ch * 16);
        if (q0 + row < Nq) stg16(out_rows + (int64_t)row * o_ld + ch * 8, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slab may be rewritten by this wave's next tile
}

template <typename T>
__global__ void __launch_bounds__(kThreads, 2) flash_attn64_kernel(const edtr_attn_params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [2 buffers][K tile | V^T tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    // block -> (image, head, 128-query block).  The query blocks of ONE (image, head) read the same K / V^T: dealt round-robin over
    // the eight XCDs (blocks b and b + 8 share one) they made every L2 fetch every head's keys and values — 102 MB per launch at
    // N = 1024 for 42 MB of operands (profiles/r05/pmc_hbm_traffic.json).  With B H a multiple of 8 the heads are dealt to the XCDs
    // instead and a head's query blocks follow each other on its XCD (the mapping of the v3 kernel below).
    const int nqb = (p.Nq + 127) / 128, BH = p.B * p.H;
    int bh, qblk;
    if ((BH & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = xcd + 8 * (j / nqb);
        qblk = j % nqb;
    } else {
        bh = blockIdx.x / nqb;
        qblk = blockIdx.x % nqb;
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q_row = qblk * 128 + wave * 32 + l31;
    const bool q_ok = q_row < p.Nq;

    const uint16_t* qp = static_cast<const uint16_t*>(p.q) + b * p.q_bs + h * 64;
    const uint16_t* kp = static_cast<const uint16_t*>(p.k) + b * p.k_bs + h * 64;
    const uint16_t* vp = static_cast<const uint16_t*>(p.vt) + b * p.vt_bs + (int64_t)h * 64 * p.vt_ld;

    // Q^T fragments (B operand of S^T = K Q^T): lane (q, half) holds Q[q][16ks + 8*half .. +7]
    U4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = zero16();
        if (q_ok) qf[ks] = ldg16(qp + (int64_t)q_row * p.q_ld + ks * 16 + lh * 8);
    }

    // ---- tile staging by LDS-DMA: one instruction = 8 tile rows of 128 bytes; a wave owns rows wave*16 + 8j + (lane>>3)
    const u32x4 srd_k = srd_of(kp), srd_v = srd_of(vp);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem));
    const int rsub = lane >> 3, slot = lane & 7;
    uint32_t koff[2], voff[2];     // constant per-lane byte offsets (tile 0); a tile step only moves the scalar offset
    int vkey[2], krow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 16 + 8 * j + rsub;
        const int chunk = slot ^ ((row >> 1) & 7);
        krow[j] = row;                                      // key index inside the tile
        koff[j] = (uint32_t)(((int64_t)row * p.k_ld + chunk * 8) * 2);
        vkey[j] = chunk * 8;                                // first key of this chunk inside the tile
        voff[j] = (uint32_t)(((int64_t)row * p.vt_ld + chunk * 8) * 2);
    }
    auto issue_tile = [&](int t, int buf) {
        const int kv0 = t * KV;
        const uint32_t sk = smem_base + buf * 2 * TILE_BYTES + wave * (16 * 128);
        const uint32_t sv = sk + TILE_BYTES;
        const uint32_t soff_k = (uint32_t)kv0 * (uint32_t)p.k_ld * 2u, soff_v = (uint32_t)kv0 * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dma16(kv0 + krow[j] < p.Nk ? koff[j] : kOob, srd_k, soff_k, sk + j * 1024);
            dma16(kv0 + vkey[j] < p.Nk ? voff[j] : kOob, srd_v, soff_v, sv + j * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
    float m_run = -1e30f, l_run = 0.0f;
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;  // scores are kept raw; p = exp2(c*s - c*m)
    float mc = m_run * c;

    const int nt = (p.Nk + KV - 1) / KV;
    issue_tile(0, 0);
    const int krd = swap23(l31);  // LDS row of the K tile feeding MFMA row l31

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            issue_tile(t + 1, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's 4 DMAs of tile t have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sk = smem + cur * 2 * TILE_BYTES;
        const char* sv = sk + TILE_BYTES;

        // ---- S^T[key][q] for 2 key blocks of 32
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const U4 kf = *reinterpret_cast<const U4*>(sk + tile_off(kb * 32 + krd, 2 * ks + lh));
                s[kb] = T::mfma(kf, qf[ks], s[kb]);
            }
        }
        // register r of block kb at lane half lh is key  t*64 + kb*32 + 16*(r>>3) + 8*lh + (r&7)
        if ((t + 1) * KV > p.Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KV + kb * 32 + 16 * (r >> 3) + 8 * lh + (r & 7);
                    if (key >= p.Nk) s[kb][r] = -1e30f;
                }
        }
        if (p.causal && (t + 1) * KV > qblk * 128 + wave * 32) {   // this tile reaches past some query of the wave
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KV + kb * 32 + 16 * (r >> 3) + 8 * lh + (r & 7);
                    if (key > q_row) s[kb][r] = -1e30f;
                }
        }
        // ---- online softmax (per lane = per query; partner lane^32 holds the other 32 keys)
        float mt = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        // deferred rescale: keep the old running max unless some row's max grew by more than 2^kDeferLog2
        if (!__all((mt - m_run) * c <= kDeferLog2)) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            mc = m_new * c;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        float psum = 0.0f;
        U4 pf[2][2];  // [key block][16-key step]: B operand fragments of P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c, -mc));
                psum += pr[r];
            }
            pf[kb][0].x = pack2<T>(pr[0], pr[1]);   pf[kb][0].y = pack2<T>(pr[2], pr[3]);
            pf[kb][0].z = pack2<T>(pr[4], pr[5]);   pf[kb][0].w = pack2<T>(pr[6], pr[7]);
            pf[kb][1].x = pack2<T>(pr[8], pr[9]);   pf[kb][1].y = pack2<T>(pr[10], pr[11]);
            pf[kb][1].z = pack2<T>(pr[12], pr[13]); pf[kb][1].w = pack2<T>(pr[14], pr[15]);
        }
        l_run += psum;
        // ---- O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const U4 vf = *reinterpret_cast<const U4*>(sv + tile_off(db * 32 + l31, kb * 4 + 2 * st + lh));
                    o[db] = T::mfma(vf, pf[kb][st], o[db]);
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // buffer `cur` is refilled by the next iteration's DMA
        asm volatile("" ::: "memory");
    }

    // ---- normalise and store: lane (q, half) holds d = db*32 + 8g + 4*half + (0..3) in regs 4g..4g+3; the tile goes through a
    // wave-private slab of the (now dead: the loop's last barrier has passed) K / V^T buffers to whole-row stores
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q0 = qblk * 128 + wave * 32;
    store_o_tile<T>(smem + wave * OSLAB, o[0], o[1], inv, static_cast<uint16_t*>(p.out) + b * p.o_bs + (int64_t)q0 * p.o_ld + h * 64,
                    p.o_ld, q0, p.Nq, lane);
}


// =====================================================================================================================
// Split-operand kernel (round 4): the ROBUST parity mode's attention.  On the heavy-tailed weight set (sharp attention) the fp16
// rounding of q and k ALONE puts a denoiser evaluation 2.7e-3 from the reference (tests/heavy_attention_budget.py: fp32 oracle
// with roundings injected; p 4.8e-4, v 4.4e-4, everything else exact 6e-6) — it bounded every GPU mode of rounds 1-3, whose
// attention operands were one fp16 part.  Here the operands arrive as fp16 hi + lo PAIRS (x = hi + lo to ~22 bits, written by
// edtr_split_operand from the fp32 projections) and each product runs as three MFMA products, the trick the GEMMs of the parity
// modes use:   S = Qh Kh^T + Ql Kh^T + Qh Kl^T     and, with PV,     O = Ph Vh + Pl Vh + Ph Vl   (P split in registers).
// Same structure as the generic kernel (one workgroup = 4 waves = 128 queries, 64-key tiles by LDS-DMA, online softmax in the exp2
// domain, deferred rescale); 3 or 4 tiles per stage (K hi, K lo, V^T hi, V^T lo) and three times the MFMAs.  Correctness first:
// this is the `precision="high"` path, not the throughput path.
// =====================================================================================================================
template <typename T, bool PV>
__global__ void __launch_bounds__(kThreads, 1) flash_attn64_split_kernel(const edtr_attn_params p) {
    constexpr int NTL = PV ? 4 : 3;                                      // tiles per stage: K hi | K lo | V^T hi | (V^T lo)
    __shared__ __attribute__((aligned(16))) char smem[2 * NTL * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_row = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = q_row < p.Nq;

    const int64_t qo = b * p.q_bs + h * 64, ko = b * p.k_bs + h * 64, vo = b * p.vt_bs + (int64_t)h * 64 * p.vt_ld;
    const uint16_t* qph = static_cast<const uint16_t*>(p.q) + qo;
    const uint16_t* qpl = static_cast<const uint16_t*>(p.q_lo) + qo;
    U4 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qh[ks] = zero16();
        ql[ks] = zero16();
        if (q_ok) {
            qh[ks] = ldg16(qph + (int64_t)q_row * p.q_ld + ks * 16 + lh * 8);
            ql[ks] = ldg16(qpl + (int64_t)q_row * p.q_ld + ks * 16 + lh * 8);
        }
    }
    const u32x4 srd_kh = srd_of(static_cast<const uint16_t*>(p.k) + ko), srd_kl = srd_of(static_cast<const uint16_t*>(p.k_lo) + ko);
    const u32x4 srd_vh = srd_of(static_cast<const uint16_t*>(p.vt) + vo);
    const u32x4 srd_vl = srd_of(static_cast<const uint16_t*>(PV ? p.vt_lo : p.vt) + vo);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem));
    const int rsub = lane >> 3, slot = lane & 7;
    uint32_t koff[2], voff[2];
    int vkey[2], krow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 16 + 8 * j + rsub;
        const int chunk = slot ^ ((row >> 1) & 7);
        krow[j] = row;
        koff[j] = (uint32_t)(((int64_t)row * p.k_ld + chunk * 8) * 2);
        vkey[j] = chunk * 8;
        voff[j] = (uint32_t)(((int64_t)row * p.vt_ld + chunk * 8) * 2);
    }
    auto issue_tile = [&](int t, int buf) {
        const int kv0 = t * KV;
        const uint32_t s0 = smem_base + buf * NTL * TILE_BYTES + wave * (16 * 128);
        const uint32_t soff_k = (uint32_t)kv0 * (uint32_t)p.k_ld * 2u, soff_v = (uint32_t)kv0 * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t ko_ = kv0 + krow[j] < p.Nk ? koff[j] : kOob, vo_ = kv0 + vkey[j] < p.Nk ? voff[j] : kOob;
            dma16(ko_, srd_kh, soff_k, s0 + j * 1024);
            dma16(ko_, srd_kl, soff_k, s0 + TILE_BYTES + j * 1024);
            dma16(vo_, srd_vh, soff_v, s0 + 2 * TILE_BYTES + j * 1024);
            if constexpr (PV) dma16(vo_, srd_vl, soff_v, s0 + 3 * TILE_BYTES + j * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
    float m_run = -1e30f, l_run = 0.0f;
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
    float mc = m_run * c;
    const int nt = (p.Nk + KV - 1) / KV;
    issue_tile(0, 0);
    const int krd = swap23(l31);

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            issue_tile(t + 1, cur ^ 1);
            if constexpr (PV) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* skh = smem + cur * NTL * TILE_BYTES;
        const char* skl = skh + TILE_BYTES;
        const char* svh = skh + 2 * TILE_BYTES;
        const char* svl = skh + 3 * TILE_BYTES;

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = tile_off(kb * 32 + krd, 2 * ks + lh);
                const U4 kfh = *reinterpret_cast<const U4*>(skh + off), kfl = *reinterpret_cast<const U4*>(skl + off);
                s[kb] = T::mfma(kfl, qh[ks], s[kb]);        // the small products first
                s[kb] = T::mfma(kfh, ql[ks], s[kb]);
                s[kb] = T::mfma(kfh, qh[ks], s[kb]);
            }
        }
        if ((t + 1) * KV > p.Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KV + kb * 32 + 16 * (r >> 3) + 8 * lh + (r & 7);
                    if (key >= p.Nk) s[kb][r] = -1e30f;
                }
        }
        if (p.causal && (t + 1) * KV > blockIdx.x * 128 + wave * 32) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KV + kb * 32 + 16 * (r >> 3) + 8 * lh + (r & 7);
                    if (key > q_row) s[kb][r] = -1e30f;
                }
        }
        float mt = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (!__all((mt - m_run) * c <= kDeferLog2)) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            mc = m_new * c;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        float psum = 0.0f;
        U4 pfh[2][2], pfl[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float pr[16], pl[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c, -mc));
                psum += pr[r];
                pl[r] = pr[r] - T::to_f32(T::from_f32(pr[r]));          // the part the 16-bit rounding drops
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                pfh[kb][st].x = pack2<T>(pr[8 * st + 0], pr[8 * st + 1]); pfh[kb][st].y = pack2<T>(pr[8 * st + 2], pr[8 * st + 3]);
                pfh[kb][st].z = pack2<T>(pr[8 * st + 4], pr[8 * st + 5]); pfh[kb][st].w = pack2<T>(pr[8 * st + 6], pr[8 * st + 7]);
                if constexpr (PV) {
                    pfl[kb][st].x = pack2<T>(pl[8 * st + 0], pl[8 * st + 1]); pfl[kb][st].y = pack2<T>(pl[8 * st + 2], pl[8 * st + 3]);
                    pfl[kb][st].z = pack2<T>(pl[8 * st + 4], pl[8 * st + 5]); pfl[kb][st].w = pack2<T>(pl[8 * st + 6], pl[8 * st + 7]);
                }
            }
        }
        l_run += psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int off = tile_off(db * 32 + l31, kb * 4 + 2 * st + lh);
                    const U4 vfh = *reinterpret_cast<const U4*>(svh + off);
                    if constexpr (PV) {
                        const U4 vfl = *reinterpret_cast<const U4*>(svl + off);
                        o[db] = T::mfma(vfl, pfh[kb][st], o[db]);
                        o[db] = T::mfma(vfh, pfl[kb][st], o[db]);
                    }
                    o[db] = T::mfma(vfh, pfh[kb][st], o[db]);
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        // the output feeds a multi-part GEMM (attn.out): fp32 when out_f32, else one 16-bit part
        if (p.out_f32) {
            float* op = static_cast<float*>(p.out) + b * p.o_bs + (int64_t)q_row * p.o_ld + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    v[0] = o[db][4 * g + 0] * inv; v[1] = o[db][4 * g + 1] * inv; v[2] = o[db][4 * g + 2] * inv; v[3] = o[db][4 * g + 3] * inv;
                    *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * lh) = v;
                }
        } else {
            uint16_t* op = static_cast<uint16_t*>(p.out) + b * p.o_bs + (int64_t)q_row * p.o_ld + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v;
                    v.x = pack2<T>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
                    v.y = pack2<T>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                    *reinterpret_cast<uint2*>(op + db * 32 + 8 * g + 4 * lh) = v;
                }
        }
    }
}


// =====================================================================================================================
// Small-Nk kernel (round 4): the cross-attention of every transformer block — Nk = 77 context tokens (any Nk <= 128) — 92 of the
// 184 attention launches of a pass.  The generic kernel above treats it like any other shape: a workgroup per 128 queries stages
// its two key tiles, runs the ONLINE softmax machinery (running maximum, deferred rescale, two barriers per tile) over them and
// leaves; at 64x64 latents that is 1280 workgroups whose fixed costs (tile flight, barriers) exceed their 24 MFMAs per wave:
// 18 us for a launch whose operands are 42 MB (7 us at the HBM rate).
// Here K and V^T of the (image, head) are staged ONCE per workgroup (two 64-key tiles each, 32 KiB of LDS, one wait, one barrier)
// and stay resident while every wave walks its 32-query blocks: all <= 128 scores of a query sit in registers, so the softmax is
// the plain two-pass form (max, then exp2 / sum) with no running state, and a wave's loop has no barrier at all — the next block's
// Q rows are requested (coalesced) before the current block's products.  Key blocks that lie entirely beyond Nk are skipped (77 keys:
// 3 of 4 blocks of 32; 5 of 8 key steps of the P.V product).
// Measured (tools/exp/r04_attn_shapes.py, batch 8, pipeline layout, profiles/r04/attn_shapes.log): 18.3 -> 15.3 us at 64x64 latents,
// 10.8 -> 9.3 at 32x32, 7.8 -> 6.9 at 16x16 — most of it from the whole-row stores (store_o_tile) and the coalesced Q loads, little
// from the resident K / V^T: the launch is bound by ~2000 issue cycles per 32-query block (5120 blocks on 1024 SIMDs) plus the
// first tiles' flight, not by its 42 MB of operands.
// =====================================================================================================================
template <typename T>
__global__ void __launch_bounds__(kThreads, 2) flash_attn64_smallk_kernel(const edtr_attn_params p, int blocks_per_wg) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES + 4 * OSLAB];  // [key tile t][K tile | V^T tile], then a store slab per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int nblk = (p.Nq + 31) >> 5;                                   // 32-query blocks of this (image, head)
    const int blk0 = blockIdx.x * blocks_per_wg;

    const uint16_t* qp = static_cast<const uint16_t*>(p.q) + b * p.q_bs + h * 64;
    const uint16_t* kp = static_cast<const uint16_t*>(p.k) + b * p.k_bs + h * 64;
    const uint16_t* vp = static_cast<const uint16_t*>(p.vt) + b * p.vt_bs + (int64_t)h * 64 * p.vt_ld;
    uint16_t* ob = static_cast<uint16_t*>(p.out) + b * p.o_bs + h * 64;

    // ---- K / V^T of this (image, head): both key tiles at once (keys >= Nk fail the range check and land as zeros)
    const int nt = (p.Nk + KV - 1) / KV;                                  // 1 or 2
    {
        const u32x4 srd_k = srd_of(kp), srd_v = srd_of(vp);
        const uint32_t smem_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem));
        const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nt) {
                const int kv0 = t * KV;
                const uint32_t sk = smem_base + t * 2 * TILE_BYTES + wave * (16 * 128), sv = sk + TILE_BYTES;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wave * 16 + 8 * j + rsub, chunk = slot ^ ((row >> 1) & 7);
                    const uint32_t koff = (uint32_t)(((int64_t)(kv0 + row) * p.k_ld + chunk * 8) * 2);
                    const uint32_t voff = (uint32_t)(((int64_t)row * p.vt_ld + kv0 + chunk * 8) * 2);
                    dma16(kv0 + row < p.Nk ? koff : kOob, srd_k, 0u, sk + j * 1024);
                    dma16(kv0 + chunk * 8 < p.Nk ? voff : kOob, srd_v, 0u, sv + j * 1024);
                }
            }
        }
    }
    // Q rows are LOADED coalesced (lane = (row, 16-byte chunk): 8 lanes cover a row's 128 bytes, one block ahead of its use) and
    // turned into the MFMA's B-operand fragments — lane (query l31, half lh) holds Q[q][16 ks + 8 lh .. +7] — through the wave's
    // store slab; the fragment layout read straight from memory is 32 rows x 32-byte pieces per instruction (2.7 of 18 us at
    // 64x64 latents with the loads compiled out)
    char* slab = smem + 4 * TILE_BYTES + wave * OSLAB;
    const int r8 = lane >> 3, ch8 = lane & 7;
    auto load_q = [&](int blk, U4 (&qr)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int q_row = blk * 32 + ps * 8 + r8;
            qr[ps] = zero16();
            if (blk < nblk && q_row < p.Nq) qr[ps] = ldg16(qp + (int64_t)q_row * p.q_ld + ch8 * 8);
        }
    };
    auto to_fragments = [&](const U4 (&qr)[4], U4 (&qf)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<U4*>(slab + (ps * 8 + r8) * OPITCH + ch8 * 16) = qr[ps];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const U4*>(slab + l31 * OPITCH + (2 * ks + lh) * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    U4 qf[4], qn[4];
    int blk = blk0 + wave;
    load_q(blk, qn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
    const int krd = swap23(l31);
    const int blk_end = min(blk0 + blocks_per_wg, nblk);
    for (; blk < blk_end; blk += 4) {
        to_fragments(qn, qf);
        load_q(blk + 4 < blk_end ? blk + 4 : nblk, qn);                   // (out-of-range block index: zeros, no load)
        // ---- S^T[key][q]: up to 4 key blocks of 32
        f32x16 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = -1e30f;
            if (kb * 32 < p.Nk) {
                const char* sk = smem + (kb >> 1) * 2 * TILE_BYTES;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const U4 kf = *reinterpret_cast<const U4*>(sk + tile_off((kb & 1) * 32 + krd, 2 * ks + lh));
                    s[kb] = T::mfma(kf, qf[ks], s[kb]);
                }
                if (kb * 32 + 32 > p.Nk) {      // register r at lane half lh is key kb*32 + 16*(r>>3) + 8*lh + (r&7)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + 16 * (r >> 3) + 8 * lh + (r & 7) >= p.Nk) s[kb][r] = -1e30f;
                }
            }
        }
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mc = mt * c;
        float psum = 0.0f;
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb * 32 < p.Nk) {
                float pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c, -mc));
                    psum += pr[r];
                }
                U4 pf[2];
                pf[0].x = pack2<T>(pr[0], pr[1]);   pf[0].y = pack2<T>(pr[2], pr[3]);
                pf[0].z = pack2<T>(pr[4], pr[5]);   pf[0].w = pack2<T>(pr[6], pr[7]);
                pf[1].x = pack2<T>(pr[8], pr[9]);   pf[1].y = pack2<T>(pr[10], pr[11]);
                pf[1].z = pack2<T>(pr[12], pr[13]); pf[1].w = pack2<T>(pr[14], pr[15]);
                const char* sv = smem + (kb >> 1) * 2 * TILE_BYTES + TILE_BYTES;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    if (kb * 32 + st * 16 < p.Nk) {
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const U4 vf = *reinterpret_cast<const U4*>(sv + tile_off(db * 32 + l31, (kb & 1) * 4 + 2 * st + lh));
                            o[db] = T::mfma(vf, pf[st], o[db]);
                        }
                    }
                }
            }
        }
        const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));
        store_o_tile<T>(slab, o[0], o[1], inv, ob + (int64_t)blk * 32 * p.o_ld, p.o_ld, blk * 32, p.Nq, lane);
    }
}


constexpr int QB2 = 256;                  // queries per workgroup of the large-N kernel
constexpr int NSTAGE = 4;

// =====================================================================================================================
// v3: the large-N self-attention kernel (64x64 latents: N = 4096 keys; any Nq >= 2048 with Nk % 256 == 0).
//
// At head width 64 the softmax needs more vector-ISSUE cycles than the two matrix products need matrix-pipe cycles, so the
// structure deletes issue slots instead of re-ordering them:
//   * one workgroup = 4 waves = 256 queries of one (image, head); a wave owns TWO 32-query blocks (A, B) and the whole
//     512-entry register file (one wave per SIMD): a K / V^T tile staged by 4 LDS-DMA pieces per wave feeds 32 MFMAs per wave
//     (16 in v1), and every K / V fragment read from LDS serves both blocks' products;
//   * scores arrive pre-scaled (q_prescaled) and the running maximum enters through the MFMA's C operand (a register vector
//     holding -max): ONE v_exp per score, no v_fma, and no per-score maximum — the row max is taken on the first tile; later
//     a half-row tile sum above 2^14 (one wave vote per block and tile) sends the block through an out-of-line path that
//     raises the maximum, rescales O and l and re-forms the tile (fp32 accumulators and the exponent range of bf16 / fp16 P
//     make 2^14 safe);
//   * the blocks are software-pipelined against each other in ONE instruction stream:
//        phase 1:  softmax(A, t)  beside  S_B(t) = K(t) Q_B      and  O_B += V(t-1) P_B(t-1)   (fragments kept in registers)
//        phase 2:  softmax(B, t)  beside  S_A(t+1) = K(t+1) Q_A  and  O_A += V(t) P_A(t)       (16 ds_read_b128)
//     one s_barrier per tile, tiles travel two ahead through a 4-stage LDS ring.
// hipcc's scheduler clusters the MFMAs of such a loop, shuttles accumulators between the register files around every branch
// and packs the f32 adds (the C++ form of this algorithm ran at 447 TFLOP/s where v1 runs at 711), so the whole tile loop is
// ONE generated inline-asm statement (tools/gen_attn_v2.py -> attn_v3_loop.inc) that owns its registers; the C++ around it
// computes addresses, loads Q and stores O.  Measured (MI355X, B 8 x 5 heads x 4096^2, bf16): 201 us = 854 TFLOP/s;
// in-kernel stamps (tools/exp/attn_variants.py): ~2010 cycles per tile and wave = 32 MFMA slots of ~44 issue cycles (v_exp
// ~13 each), 255 for the 4 LDS-DMA pieces, 95 at the barrier — the wave is ISSUE-bound, the matrix pipe is busy 51 %.
// =====================================================================================================================
#ifndef EDTR_ATTN_V3_INC
#define EDTR_ATTN_V3_INC "attn_v3_loop.inc"
#endif
#include EDTR_ATTN_V3_INC
#ifdef EDTR_STAMPS      // diagnostic build (tools/exp/attn_variants.py): per-wave cycle sums of the loop's segments
__device__ unsigned g_attn_stamps[1 << 16];
#endif

template <typename T>
__global__ void __launch_bounds__(kThreads, 1) flash_attn64_v3_kernel(const edtr_attn_params p) {
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * 2 * TILE_BYTES];  // [stage][K tile | V^T tile] = 64 KiB

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int nqb = (p.Nq + QB2 - 1) / QB2;
    const int BH = p.B * p.H;
    int bh, qb;
    {
        const int bid = blockIdx.x;
        if ((BH & 7) == 0) {
            const int xcd = bid & 7, j = bid >> 3;
            bh = xcd + 8 * (j / nqb);
            qb = j % nqb;
        } else {
            bh = bid / nqb;
            qb = bid % nqb;
        }
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q_rowA = qb * QB2 + wave * 64 + l31, q_rowB = q_rowA + 32;

    const uint16_t* qp = static_cast<const uint16_t*>(p.q) + b * p.q_bs + h * 64;
    const uint16_t* kp = static_cast<const uint16_t*>(p.k) + b * p.k_bs + h * 64;
    const uint16_t* vp = static_cast<const uint16_t*>(p.vt) + b * p.vt_bs + (int64_t)h * 64 * p.vt_ld;

    U4 qA[4], qB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qA[ks] = zero16();
        qB[ks] = zero16();
        if (q_rowA < p.Nq) qA[ks] = ldg16(qp + (int64_t)q_rowA * p.q_ld + ks * 16 + lh * 8);
        if (q_rowB < p.Nq) qB[ks] = ldg16(qp + (int64_t)q_rowB * p.q_ld + ks * 16 + lh * 8);
    }
    const u32x4 srd_k = srd_of(kp), srd_v = srd_of(vp);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem));
    const uint32_t ldsb = smem_base + (uint32_t)wave * (16 * 128);
    const uint32_t stepk = (uint32_t)KV * (uint32_t)p.k_ld * 2u;
    const int nt = p.Nk / KV;
    // DMA lane offsets: pieces 0/1 = K rows wave*16 + 8j + (lane>>3), pieces 2/3 = V^T rows likewise (as v1 / v2)
    uint32_t doff[4];
    {
        const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wave * 16 + 8 * j + rsub;
            const int chunk = slot ^ ((row >> 1) & 7);
            doff[j] = (uint32_t)(((int64_t)row * p.k_ld + chunk * 8) * 2);
            doff[2 + j] = (uint32_t)(((int64_t)row * p.vt_ld + chunk * 8) * 2);
        }
    }
    // LDS fragment addresses (byte address in LDS space, stage / key-block / d-block offsets are immediates in the stream)
    const int krd = swap23(l31);
    int kad[4], vad[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kad[ks] = (int)smem_base + tile_off(krd, 2 * ks + lh);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st) vad[kb * 2 + st] = (int)smem_base + tile_off(l31, kb * 4 + 2 * st + lh);

    float oa0[16], oa1[16], ob0[16], ob1[16], la, lb;
#ifdef EDTR_STAMPS
    unsigned stamps[4];
#endif
    if constexpr (__is_same(T, BF16)) {
        asm volatile(EDTR_ATTN_V3_ASM("v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32") : EDTR_ATTN_V3_OUTS : EDTR_ATTN_V3_INS : EDTR_ATTN_V3_CLOBBERS);
    } else {
        asm volatile(EDTR_ATTN_V3_ASM("v_mfma_f32_32x32x16_f16", "v_cvt_pk_f16_f32") : EDTR_ATTN_V3_OUTS : EDTR_ATTN_V3_INS : EDTR_ATTN_V3_CLOBBERS);
    }

#ifdef EDTR_STAMPS
    if (lane == 0 && blockIdx.x < 4096) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g_attn_stamps[(blockIdx.x * 4 + wave) * 4 + i] = stamps[i];
    }
#endif
    const float invA = 1.0f / (la + __shfl_xor(la, 32, 64)), invB = 1.0f / (lb + __shfl_xor(lb, 32, 64));
    uint16_t* ob = static_cast<uint16_t*>(p.out) + b * p.o_bs + h * 64;
    // whole-row stores through a wave-private slab of the K / V^T ring (dead once every wave has left the tile loop)
    __syncthreads();
    const int q0 = qb * QB2 + wave * 64;
    store_o_tile<T>(smem + wave * OSLAB, oa0, oa1, invA, ob + (int64_t)q0 * p.o_ld, p.o_ld, q0, p.Nq, lane);
    store_o_tile<T>(smem + wave * OSLAB, ob0, ob1, invB, ob + (int64_t)(q0 + 32) * p.o_ld, p.o_ld, q0 + 32, p.Nq, lane);
}

}  // namespace

#ifdef EDTR_STAMPS
extern "C" int edtr_attn_stamps_read(unsigned* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_stamps), (size_t)n * 4);
}
#endif

extern "C" int edtr_flash_attn64(const edtr_attn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_attn_params& p = *pp;
    if (!p.q || !p.k || !p.vt || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0) return EDTR_E_SHAPE;
    if (p.causal && p.Nq != p.Nk) return EDTR_E_SHAPE;
    if ((p.q_ld & 7) || (p.k_ld & 7) || (p.vt_ld & 7) || (!p.out_f32 && (p.o_ld & 7)) || (p.q_bs & 7) || (p.k_bs & 7) ||
        (p.vt_bs & 7) || (!p.out_f32 && (p.o_bs & 7)))
        return EDTR_E_ALIGN;
    if (p.vt_ld < ((p.Nk + 7) & ~7)) return EDTR_E_SHAPE;
    if (!aligned16(p.q) || !aligned16(p.k) || !aligned16(p.vt) || !aligned16(p.out)) return EDTR_E_ALIGN;
    // 32-bit byte offsets inside one (image, head) slice of K and of V^T (buffer addressing)
    if ((int64_t)(p.Nk + 64) * p.k_ld * 2 >= 0xF0000000LL || (int64_t)64 * p.vt_ld * 2 >= 0xF0000000LL) return EDTR_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((p.q_lo != nullptr) != (p.k_lo != nullptr)) return EDTR_E_NULL;        // q and k are split together
    if (p.vt_lo && !p.q_lo) return EDTR_E_UNSUPPORTED;
    if (p.out_f32 && !p.q_lo) return EDTR_E_UNSUPPORTED;                        // fp32 output: the split-operand kernel only
    if (p.q_lo) {      // hi + lo operand pairs: three MFMA products per matrix product (the robust parity mode)
        if (!aligned16(p.q_lo) || !aligned16(p.k_lo) || (p.vt_lo && !aligned16(p.vt_lo))) return EDTR_E_ALIGN;
        if (p.out_f32 && ((p.o_ld & 3) || (p.o_bs & 3))) return EDTR_E_ALIGN;
        dim3 grid((p.Nq + 127) / 128, p.H, p.B);
        if (p.dtype == EDTR_BF16) {
            if (p.vt_lo) hipLaunchKernelGGL((flash_attn64_split_kernel<BF16, true>), grid, dim3(kThreads), 0, s, p);
            else hipLaunchKernelGGL((flash_attn64_split_kernel<BF16, false>), grid, dim3(kThreads), 0, s, p);
        } else {
            if (p.vt_lo) hipLaunchKernelGGL((flash_attn64_split_kernel<F16, true>), grid, dim3(kThreads), 0, s, p);
            else hipLaunchKernelGGL((flash_attn64_split_kernel<F16, false>), grid, dim3(kThreads), 0, s, p);
        }
        EDTR_LAUNCH_CHECK();
        return EDTR_OK;
    }
    static int v2_min_nq = -1;      // the large-N kernel takes over at this many queries (EDTR_ATTN_V3_MIN_NQ, 0 = never)
    if (v2_min_nq < 0) {
        const char* e = getenv("EDTR_ATTN_V3_MIN_NQ");
        v2_min_nq = e ? atoi(e) : 2048;
    }
    static int v3_on = -1;          // EDTR_ATTN_V3=0: every shape runs the v1 kernel (A/B runs on one device)
    if (v3_on < 0) {
        const char* e = getenv("EDTR_ATTN_V3");
        v3_on = (e && e[0] == '0') ? 0 : 1;
    }
    if (v3_on && p.q_prescaled && !p.causal && v2_min_nq > 0 && p.Nq >= v2_min_nq && p.Nk >= 256 && (p.Nk & 255) == 0) {
        dim3 grid3(((p.Nq + QB2 - 1) / QB2) * p.H * p.B);
        if (p.dtype == EDTR_BF16) hipLaunchKernelGGL((flash_attn64_v3_kernel<BF16>), grid3, dim3(kThreads), 0, s, p);
        else hipLaunchKernelGGL((flash_attn64_v3_kernel<F16>), grid3, dim3(kThreads), 0, s, p);
        EDTR_LAUNCH_CHECK();
        return EDTR_OK;
    }
    static int smallk_on = -1;      // EDTR_ATTN_SMALLK=0: the generic kernel for the small-Nk (cross-attention) shapes too (A/B runs)
    if (smallk_on < 0) {
        const char* e = getenv("EDTR_ATTN_SMALLK");
        smallk_on = (e && e[0] == '0') ? 0 : 1;
    }
    if (smallk_on && !p.causal && p.Nk <= 128) {
        // 32-query blocks per workgroup (4 waves): one per wave while that still gives every CU two workgroups, up to 4 per wave
        const int64_t total = (int64_t)p.B * p.H * ((p.Nq + 31) / 32);
        int per_wave = (int)(total / (4 * 512));
        per_wave = per_wave < 1 ? 1 : (per_wave > 4 ? 4 : per_wave);
        const int bpw = 4 * per_wave;
        dim3 gridk(((p.Nq + 31) / 32 + bpw - 1) / bpw, p.H, p.B);
        if (p.dtype == EDTR_BF16) hipLaunchKernelGGL((flash_attn64_smallk_kernel<BF16>), gridk, dim3(kThreads), 0, s, p, bpw);
        else hipLaunchKernelGGL((flash_attn64_smallk_kernel<F16>), gridk, dim3(kThreads), 0, s, p, bpw);
        EDTR_LAUNCH_CHECK();
        return EDTR_OK;
    }
    dim3 grid(((p.Nq + 127) / 128) * p.H * p.B);        // one dimension: the kernel deals heads, not query blocks, to the XCDs
    if (p.dtype == EDTR_BF16) hipLaunchKernelGGL((flash_attn64_kernel<BF16>), grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((flash_attn64_kernel<F16>), grid, dim3(kThreads), 0, s, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}
