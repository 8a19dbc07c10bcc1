// Fused attention, head width 64, for gfx950 (see include/edtr_hip.h: edtr_flash_attn64).
//
// One workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.
// Per 64-key tile a wave computes S^T = K Q^T (keys on the MFMA rows, queries on the lanes), so a
// lane holds 32 of the 64 scores of ITS query in registers: the row max / row sum are in-lane
// reductions plus one exchange with lane^32 (wavefront shuffle), never through LDS.
// The probabilities are converted to 16-bit in registers and fed straight back as the B operand of
// O^T += V^T P^T ("accumulator tile as the next MFMA's operand"): K rows are read from LDS in an
// order with bits 2,3 of the row swapped, which makes registers 8s..8s+7 of a score accumulator
// exactly keys 16s+8h..16s+8h+7 — the natural k order of the V^T fragment (one ds_read_b128).
// V arrives key-major (V^T, written that way by the projection GEMM), so no transposed read is
// needed.  K / V^T tiles are staged global -> registers -> LDS, double buffered, one barrier per tile.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int KV = 64;            // keys per tile
constexpr int TILE_BYTES = 64 * 64 * 2;

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <typename T>
__global__ void __launch_bounds__(kThreads, 2) flash_attn64_kernel(const edtr_attn_params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [2 buffers][K tile | V^T tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // tile staging: thread -> chunk column kc, rows (tid>>3) and (tid>>3)+32
    const int kc = tid & 7, r0 = tid >> 3;
    U4 rk[2], rv[2];
    auto load_tile = [&](int t) {
        const int kv0 = t * KV;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + 32 * i;
            rk[i] = zero16();
            if (kv0 + row < p.Nk) rk[i] = ldg16(kp + (int64_t)(kv0 + row) * p.k_ld + kc * 8);
            rv[i] = zero16();
            if (kv0 + kc * 8 < p.Nk) rv[i] = ldg16(vp + (int64_t)row * p.vt_ld + kv0 + kc * 8);
        }
    };
    auto store_tile = [&](int buf) {
        char* sk = smem + buf * 2 * TILE_BYTES;
        char* sv = sk + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<U4*>(sk + tile_off(r0 + 32 * i, kc)) = rk[i];
            *reinterpret_cast<U4*>(sv + tile_off(r0 + 32 * i, kc)) = rv[i];
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
    float m_run = -1e30f, l_run = 0.0f;
    const float c = p.scale * 1.4426950408889634f;  // scores are kept raw; exp2(c*s - c*m)

    const int nt = (p.Nk + KV - 1) / KV;
    load_tile(0);
    store_tile(0);
    if (nt > 1) load_tile(1);
    __syncthreads();

    const int krow = swap23(l31);  // LDS row of the K tile feeding MFMA row l31

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
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
                const U4 kf = *reinterpret_cast<const U4*>(sk + tile_off(kb * 32 + krow, 2 * ks + lh));
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
        // ---- online softmax (per lane = per query; partner lane^32 holds the other 32 keys)
        float mt = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
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
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
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

        if (t + 1 < nt) {
            store_tile(cur ^ 1);
            if (t + 2 < nt) load_tile(t + 2);
        }
        __syncthreads();
    }

    // ---- normalise and store: lane (q, half) holds d = db*32 + 8g + 4*half + (0..3) in regs 4g..4g+3
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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
    if ((p.q_ld & 7) || (p.k_ld & 7) || (p.vt_ld & 7) || (p.o_ld & 7) || (p.q_bs & 7) || (p.k_bs & 7) ||
        (p.vt_bs & 7) || (p.o_bs & 7))
        return EDTR_E_ALIGN;
    if (p.vt_ld < ((p.Nk + 7) & ~7)) return EDTR_E_SHAPE;
    if (!aligned16(p.q) || !aligned16(p.k) || !aligned16(p.vt) || !aligned16(p.out)) return EDTR_E_ALIGN;
    dim3 grid((p.Nq + 127) / 128, p.H, p.B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p.dtype == EDTR_BF16)
        hipLaunchKernelGGL(flash_attn64_kernel<BF16>, grid, dim3(kThreads), 0, s, p);
    else
        hipLaunchKernelGGL(flash_attn64_kernel<F16>, grid, dim3(kThreads), 0, s, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}
