// SwinIR kernels for gfx950 (see include/edtr_hip.h: edtr_window_attn, edtr_pixel_unshuffle).
//
// edtr_window_attn — one wavefront per (window, head), four per workgroup, no workgroup barrier.
// The 8 x 8 window is 64 queries x 64 keys at head width <= 32 (zero padded to 32 by the packed projection), i.e. per wave
//   S^T = K Q^T : 2 x 2 blocks of 32 x 32, K-depth 32  ->  8 MFMA 32x32x16
//   O^T = V^T P^T : 1 x 2 blocks,          K-depth 64  ->  8 MFMA
// so the kernel is pure latency / HBM traffic (12 KiB in, 4 KiB out per wave); what matters is that nothing but q, k, v
// and the output ever touches memory: the cyclic shift and the window gather / scatter are address arithmetic on the token
// row, the bias is a 96 KiB L2-resident table, the region mask is one 8-byte load per 8 keys.
// Register layout as in attention.hip: keys sit on the MFMA rows of S^T and are fed with bits 2,3 of the row swapped, which
// makes accumulator registers 8s..8s+7 of lane (query, half) hold keys 16s + 8*half + 0..7 — exactly the k order of the
// next MFMA's B operand, so the probabilities go straight back into the matrix core after one v_cvt_pk per pair.
// V is token-major in memory but the V^T operand needs 8 consecutive keys of one channel per lane: each wave transposes its
// 64 x 32 V block through a private 4.5 KiB LDS slab (ds_write_b16 columns, ds_read_b128 rows).
#include "common.h"

namespace {

constexpr int WS = 8, NTOK = 64, HP = 32;          // window edge, tokens per window, padded head width
constexpr int VT_PITCH = 72;                        // 16-bit elements per V^T row in LDS (64 keys + 8 pad: 144-byte rows)

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <typename T>
__global__ void __launch_bounds__(256) window_attn_kernel(const edtr_window_attn_params p, int total_tasks) {
    __shared__ __attribute__((aligned(16))) uint16_t vt_all[4][HP * VT_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int task = blockIdx.x * 4 + wave;
    if (task >= total_tasks) return;                 // wave-uniform; there is no workgroup barrier below
    const int h = task % p.heads;
    int win = task / p.heads;
    const int nwx = p.W / WS, nwy = p.H / WS;
    const int wx = win % nwx; win /= nwx;
    const int wy = win % nwy;
    const int b = win / nwy;

    // token n of this window -> row of the unshifted token matrix
    auto token_row = [&](int n) -> int64_t {
        int y = wy * WS + (n >> 3) + p.shift, x = wx * WS + (n & 7) + p.shift;
        if (y >= p.H) y -= p.H;
        if (x >= p.W) x -= p.W;
        return ((int64_t)b * p.H + y) * p.W + x;
    };
    const uint16_t* base = static_cast<const uint16_t*>(p.qkv);
    const int qcol = h * HP, kcol = (p.heads + h) * HP, vcol = (2 * p.heads + h) * HP;

    // ---- V: lane = key; transpose into LDS
    uint16_t* vt = vt_all[wave];
    {
        const uint16_t* vrow = base + token_row(lane) * p.ld_qkv + vcol;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const U4 v = ldg16(vrow + c * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt[(c * 8 + 2 * j) * VT_PITCH + lane] = (uint16_t)(w[j] & 0xffffu);
                vt[(c * 8 + 2 * j + 1) * VT_PITCH + lane] = (uint16_t)(w[j] >> 16);
            }
        }
    }
    // ---- Q^T (B operand) and K (A operand, rows in swap23 order) fragments straight from global memory
    U4 qf[2][2], kf[2][2];   // [block of 32][k-step of 16 channels]
    int64_t qrow[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        qrow[blk] = token_row(blk * 32 + l31);
        const uint16_t* qp = base + qrow[blk] * p.ld_qkv + qcol + lh * 8;
        const uint16_t* kp = base + token_row(blk * 32 + swap23(l31)) * p.ld_qkv + kcol + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[blk][ks] = ldg16(qp + ks * 16);
            kf[blk][ks] = ldg16(kp + ks * 16);
        }
    }

    // ---- scores: s[qb][kb] register r of lane (q = qb*32 + l31, half lh) is key kb*32 + 16*(r>>3) + 8*lh + (r&7)
    f32x16 s[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[qb][kb] = T::mfma(kf[kb][ks], qf[qb][ks], s[qb][kb]);
        }

    // ---- + relative-position bias, + region mask
    const float* bias_h = p.bias + (int64_t)h * NTOK * NTOK;
    uint32_t qlab[2] = {0u, 0u};
    if (p.labels) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int n = qb * 32 + l31;
            qlab[qb] = p.labels[(int64_t)(wy * WS + (n >> 3)) * p.W + wx * WS + (n & 7)];
        }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 2; ++g) {            // 8 keys = one row of the window
            const int key0 = kb * 32 + 16 * g + 8 * lh;
            uint64_t klab = 0;
            if (p.labels) klab = *reinterpret_cast<const uint64_t*>(p.labels + (int64_t)(wy * WS + (key0 >> 3)) * p.W + wx * WS);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float* bp = bias_h + (qb * 32 + l31) * NTOK + key0;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = __builtin_fmaf(s[qb][kb][8 * g + j], p.scale, bv[j]);
                    if (p.labels && (uint32_t)((klab >> (8 * j)) & 0xffu) != qlab[qb]) v -= 100.0f;
                    s[qb][kb][8 * g + j] = v;
                }
            }
        }

    // ---- softmax over the 64 keys of each query (32 in this lane, 32 in lane ^ 32), probabilities packed to 16 bits
    U4 pf[2][2][2];     // [qb][kb][16-key step]
    float inv_l[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float m = s[qb][0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, s[qb][0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, s[qb][1][r]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mc = m * 1.4426950408889634f;
        float l = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kb][r], 1.4426950408889634f, -mc));
                l += pr[r];
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                pf[qb][kb][st].x = pack2<T>(pr[8 * st + 0], pr[8 * st + 1]);
                pf[qb][kb][st].y = pack2<T>(pr[8 * st + 2], pr[8 * st + 3]);
                pf[qb][kb][st].z = pack2<T>(pr[8 * st + 4], pr[8 * st + 5]);
                pf[qb][kb][st].w = pack2<T>(pr[8 * st + 6], pr[8 * st + 7]);
            }
        }
        l += __shfl_xor(l, 32, 64);
        inv_l[qb] = 1.0f / l;
    }

    // ---- O^T[d][q] = V^T[d][key] P^T[key][q]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's V^T writes have landed (LDS ops of one wave are in order)
    __builtin_amdgcn_wave_barrier();
    f32x16 o[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const U4 vf = *reinterpret_cast<const U4*>(vt + l31 * VT_PITCH + kb * 32 + 16 * st + 8 * lh);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) o[qb] = T::mfma(vf, pf[qb][kb][st], o[qb]);
        }

    // ---- store: lane (q, half) holds channels d = 8g + 4*half + 0..3 in registers 4g..4g+3
    uint16_t* outp = static_cast<uint16_t*>(p.out);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        uint16_t* op = outp + qrow[qb] * p.ld_out + h * p.head_dim;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * lh;
            if (d + 1 < p.head_dim)
                *reinterpret_cast<uint32_t*>(op + d) = pack2<T>(o[qb][4 * g + 0] * inv_l[qb], o[qb][4 * g + 1] * inv_l[qb]);
            if (d + 3 < p.head_dim)
                *reinterpret_cast<uint32_t*>(op + d + 2) = pack2<T>(o[qb][4 * g + 2] * inv_l[qb], o[qb][4 * g + 3] * inv_l[qb]);
        }
        if (h == 0) {        // the head-0 wave also zeroes the pad columns of its 64 token rows
            uint16_t* zp = outp + qrow[qb] * p.ld_out;
            for (int c = p.heads * p.head_dim + 2 * lh; c < p.c_pad; c += 4) *reinterpret_cast<uint32_t*>(zp + c) = 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pixel-unshuffle front end: one thread = one (token, channel, dy) run of r source pixels
template <typename T>
__global__ void __launch_bounds__(256) pixel_unshuffle_kernel(const float* src, int B, int C, int H, int W, int r, const float* sub,
                                                             float scale, uint16_t* dst, int ld, int zero_pad_to) {
    const int tw = W / r, th = H / r;
    const int64_t total = (int64_t)B * C * H * tw;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % tw);
    int64_t t = i / tw;
    const int yy = (int)(t % H); t /= H;          // source row = y*r + dy
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const int y = yy / r, dy = yy - y * r;
    const float* sp = src + (((int64_t)b * C + c) * H + yy) * W + (int64_t)x * r;
    const float m = sub ? sub[c] : 0.0f;
    uint16_t* dp = dst + (((int64_t)b * th + y) * tw + x) * ld + (c * r + dy) * r;
    if (r == 8) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 4);
        const float f[8] = {(a0[0] - m) * scale, (a0[1] - m) * scale, (a0[2] - m) * scale, (a0[3] - m) * scale,
                            (a1[0] - m) * scale, (a1[1] - m) * scale, (a1[2] - m) * scale, (a1[3] - m) * scale};
        stg16(dp, pack8<T>(f));
    } else {
        for (int dx = 0; dx < r; ++dx) dp[dx] = T::from_f32((sp[dx] - m) * scale);
    }
    if (c == 0 && dy == 0) {
        uint16_t* row = dst + (((int64_t)b * th + y) * tw + x) * ld;
        for (int z = C * r * r; z < zero_pad_to; ++z) row[z] = 0;
    }
}

}  // namespace

extern "C" int edtr_window_attn(const edtr_window_attn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_window_attn_params& p = *pp;
    if (!p.qkv || !p.out || !p.bias) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.heads <= 0 || p.head_dim <= 0) return EDTR_E_SHAPE;
    if ((p.H % WS) || (p.W % WS)) return EDTR_E_SHAPE;
    if (p.head_dim > HP || (p.head_dim & 1)) return EDTR_E_UNSUPPORTED;
    if (p.shift < 0 || p.shift >= WS) return EDTR_E_SHAPE;
    if (p.shift > 0 && !p.labels) return EDTR_E_NULL;
    if (p.ld_qkv < 3 * p.heads * HP || p.c_pad < p.heads * p.head_dim || p.ld_out < p.c_pad) return EDTR_E_SHAPE;
    if ((p.ld_qkv & 7) || (p.ld_out & 7) || (p.c_pad & 3) || ((p.heads * p.head_dim) & 3)) return EDTR_E_ALIGN;
    if (!aligned16(p.qkv) || !aligned16(p.out) || !aligned16(p.bias)) return EDTR_E_ALIGN;
    if (p.labels && (reinterpret_cast<uintptr_t>(p.labels) & 7u)) return EDTR_E_ALIGN;
    const int64_t tasks = (int64_t)p.B * (p.H / WS) * (p.W / WS) * p.heads;
    if (tasks > 0x7fffffffLL) return EDTR_E_UNSUPPORTED;
    dim3 grid((unsigned)((tasks + 3) / 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p.dtype == EDTR_BF16)
        hipLaunchKernelGGL(window_attn_kernel<BF16>, grid, dim3(256), 0, s, p, (int)tasks);
    else
        hipLaunchKernelGGL(window_attn_kernel<F16>, grid, dim3(256), 0, s, p, (int)tasks);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_pixel_unshuffle(int dtype, const float* src, int B, int C, int H, int W, int r, const float* sub, float scale,
                                    void* dst, int ld, int zero_pad_to, edtr_stream_t stream) {
    if (!src || !dst) return EDTR_E_NULL;
    if (dtype != EDTR_BF16 && dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || r < 1 || r > 8) return EDTR_E_SHAPE;
    if ((H % r) || (W % r)) return EDTR_E_SHAPE;
    if (ld < C * r * r || zero_pad_to > ld) return EDTR_E_SHAPE;
    if ((ld & 7) || !aligned16(dst) || !aligned16(src)) return EDTR_E_ALIGN;
    if (r == 8 && (W & 3)) return EDTR_E_ALIGN;
    const int64_t total = (int64_t)B * C * H * (W / r);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EDTR_BF16)
        hipLaunchKernelGGL(pixel_unshuffle_kernel<BF16>, grid, dim3(256), 0, s, src, B, C, H, W, r, sub, scale,
                           static_cast<uint16_t*>(dst), ld, zero_pad_to);
    else
        hipLaunchKernelGGL(pixel_unshuffle_kernel<F16>, grid, dim3(256), 0, s, src, B, C, H, W, r, sub, scale,
                           static_cast<uint16_t*>(dst), ld, zero_pad_to);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}
