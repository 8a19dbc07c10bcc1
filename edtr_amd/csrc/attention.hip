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

// LSUM (staged experiment, EDTR_ATTN_LSUM_MFMA=1, not yet validated on hardware): the softmax row sums come from the matrix
// core — a constant all-ones A fragment multiplies the same P^T fragments that feed O^T += V^T P^T — instead of 32 v_add_f32 per
// lane and tile: the kernel is bound by vector ISSUE slots (v_exp 8 cycles, other VALU 4), an extra MFMA costs 8 of them.
template <typename T, bool LSUM = false>
__global__ void __launch_bounds__(kThreads, 2) flash_attn64_kernel(const edtr_attn_params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [2 buffers][K tile | V^T tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_row = blockIdx.x * 128 + wave * 32 + l31;
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
    f32x16 ol;                     // LSUM: every row of this accumulator is the running row sum of the lane's query
    U4 ones;
    if constexpr (LSUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ol[r] = 0.0f;
        ones.x = ones.y = ones.z = ones.w = T::kOnePair;
    }
    const float c = p.scale * 1.4426950408889634f;  // scores are kept raw; p = exp2(c*s - c*m)
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
        if (p.causal && (t + 1) * KV > blockIdx.x * 128 + wave * 32) {   // this tile reaches past some query of the wave
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
            if constexpr (LSUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ol[r] *= alpha;
            }
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
                if constexpr (!LSUM) psum += pr[r];
            }
            pf[kb][0].x = pack2<T>(pr[0], pr[1]);   pf[kb][0].y = pack2<T>(pr[2], pr[3]);
            pf[kb][0].z = pack2<T>(pr[4], pr[5]);   pf[kb][0].w = pack2<T>(pr[6], pr[7]);
            pf[kb][1].x = pack2<T>(pr[8], pr[9]);   pf[kb][1].y = pack2<T>(pr[10], pr[11]);
            pf[kb][1].z = pack2<T>(pr[12], pr[13]); pf[kb][1].w = pack2<T>(pr[14], pr[15]);
        }
        if constexpr (!LSUM) l_run += psum;
        if constexpr (LSUM) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) ol = T::mfma(ones, pf[kb][st], ol);
        }
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

    // ---- normalise and store: lane (q, half) holds d = db*32 + 8g + 4*half + (0..3) in regs 4g..4g+3
    float l_tot;
    if constexpr (LSUM) l_tot = ol[0];      // the MFMA already summed over both lane halves' keys
    else l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
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

}  // namespace

extern "C" int edtr_flash_attn64(const edtr_attn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_attn_params& p = *pp;
    if (!p.q || !p.k || !p.vt || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0) return EDTR_E_SHAPE;
    if (p.causal && p.Nq != p.Nk) return EDTR_E_SHAPE;
    if ((p.q_ld & 7) || (p.k_ld & 7) || (p.vt_ld & 7) || (p.o_ld & 7) || (p.q_bs & 7) || (p.k_bs & 7) ||
        (p.vt_bs & 7) || (p.o_bs & 7))
        return EDTR_E_ALIGN;
    if (p.vt_ld < ((p.Nk + 7) & ~7)) return EDTR_E_SHAPE;
    if (!aligned16(p.q) || !aligned16(p.k) || !aligned16(p.vt) || !aligned16(p.out)) return EDTR_E_ALIGN;
    // 32-bit byte offsets inside one (image, head) slice of K and of V^T (buffer addressing)
    if ((int64_t)(p.Nk + 64) * p.k_ld * 2 >= 0xF0000000LL || (int64_t)64 * p.vt_ld * 2 >= 0xF0000000LL) return EDTR_E_UNSUPPORTED;
    dim3 grid((p.Nq + 127) / 128, p.H, p.B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    static int lsum = -1;           // staged experiment switch, read once
    if (lsum < 0) {
        const char* e = getenv("EDTR_ATTN_LSUM_MFMA");
        lsum = (e && e[0] == '1') ? 1 : 0;
    }
    if (lsum) {
        if (p.dtype == EDTR_BF16) hipLaunchKernelGGL((flash_attn64_kernel<BF16, true>), grid, dim3(kThreads), 0, s, p);
        else hipLaunchKernelGGL((flash_attn64_kernel<F16, true>), grid, dim3(kThreads), 0, s, p);
    } else if (p.dtype == EDTR_BF16)
        hipLaunchKernelGGL((flash_attn64_kernel<BF16, false>), grid, dim3(kThreads), 0, s, p);
    else
        hipLaunchKernelGGL((flash_attn64_kernel<F16, false>), grid, dim3(kThreads), 0, s, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}
