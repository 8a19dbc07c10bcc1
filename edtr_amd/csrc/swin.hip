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
#include <type_traits>

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
// edtr_swin_mlp — x + fc2(GELU(fc1(LayerNorm(x)))) of one Swin layer in ONE launch (reference model/swinir.py:24-37 Mlp,
// :281-283 `x = x + self.drop_path(self.mlp(self.norm2(x)))`).  The two GEMMs are K = 192 / K = 384 deep: as separate
// launches each is a 20 - 28 us round trip of the 12.6 MB token tensor (plus the 25 MB hidden tensor in between) through
// HBM / L2 for 2 - 3 us of matrix work.  Here a workgroup owns 128 tokens; the hidden activations never leave registers.
//
// Both products run TRANSPOSED so that the first one's accumulators are the second one's B operand:
//   h^T[u][t]   = sum_k  W1g[u][k] x[t][k]         A = W1g rows (LDS), B = x^T fragments (registers)
//   out^T[c][t] = sum_u  W2[c][u]  h[u][t]         A = W2 rows (LDS),  B = GELU(...) packed from the accumulators
// Both weight matrices are fed with bits 2,3 of the row index swapped (as attention.hip does with its keys), which makes
// accumulator registers 8s..8s+7 of lane (token, half) hold rows 16s + 8*half + 0..7: for h^T that is the k order of the
// next MFMA's B operand, for out^T it is the channel order of the x fragment the lane already holds — the residual comes
// from registers and the result goes back into the token tile's LDS image in place.
// LayerNorm is folded (include/edtr_hip.h, "LayerNorm folded into the GEMMs around it"): W1g = gamma . W1, the row mean and
// rstd come from the x fragments, pre = rstd (acc - mean c1[u]) + c2b[u].
//
// The kernel is as much VALU as MFMA work: the erf-GELU costs ~70 VALU cycles per hidden activation and wave (two
// quarter-rate transcendentals), 1.2k cycles per 32 x 32 hidden tile against 768 for the tile's 24 MFMAs.  So the eight
// waves PING-PONG: wave (t4, hg) owns tokens 32 t4 .. + 31 and hidden HALF hg (six tiles), the two waves of a SIMD are one
// of each half, and the workgroup runs in barrier-separated PERIODS in which half hg = t & 1 multiplies
// (second product of its tile j - 1, first product of tile j) while the other half evaluates the GELU of the tile it
// multiplied a period earlier: 14 periods of max(MFMA, VALU) instead of 6 steps of their sum (measured: DESIGN.md).
// Hidden halves rather than more tokens per workgroup so that 32768 tokens (batch 8) are 256 workgroups, one per CU.  The
// halves' partial outputs meet once, after the loop: each wave keeps three of the six 32-channel output tiles and hands the
// other three to its partner through LDS (fp32, lane-linear: conflict-free both ways).
// Everything arrives by LDS-DMA: the weights as pre-swizzled 12 KiB images (packed by the host, layout in the header), a
// "unit" = [W2 slice j - 1 | W1 tile j] of one half per period, two periods ahead, into that half's other buffer; the token
// tile with the same XOR swizzle applied on the source side, so that rows are fetched and stored as whole 384-byte runs.
constexpr int MLP_CT = 6, MLP_HT = 12;                       // 32-wide tiles of the token width 192 and of the hidden width 384
constexpr int MLP_IMG = 32 * 32 * MLP_CT * 2;                // 12288 B: one W1 tile (32 units x 192 k) or one W2 slice (192 c x 32 units)
constexpr int MLP_THREADS = 512, MLP_TOKENS = 128;
constexpr int MLP_WBUF = 4 * 2 * MLP_IMG;                    // [half][buffer][W2 slice | W1 tile]
constexpr int MLP_XT = MLP_TOKENS * 32 * MLP_CT * 2;         // token tile, rows of 384 B with the W1 image's swizzle
constexpr int MLP_CONST_FLOATS = 2 * 32 * MLP_HT + 32 * MLP_CT;   // c1 | c2b | b2
constexpr int MLP_LDS = MLP_WBUF + MLP_XT + MLP_CONST_FLOATS * 4;

// IN_LDS (edtr_swin_layer): the token tile is already in LDS (the attention half of the layer left it there) and stays there — no
// fetch, no store, no row statistics; the caller scatters it.
template <typename T, bool IN_LDS>
__device__ __forceinline__ void swin_mlp_body(const edtr_swin_mlp_params& p, char* smem) {
    int tid = threadIdx.x;
    if constexpr (IN_LDS) asm volatile("" : "+v"(tid));       // (re-derive the lane geometry here: nothing of it is carried through the attention half)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int t4 = wave & 3, hg = wave >> 2;
    char* xt = smem + MLP_WBUF;
    float* cst = reinterpret_cast<float*>(smem + MLP_WBUF + MLP_XT);
    const uint32_t lds0 = lds_addr_of(smem);
    constexpr int C = 32 * MLP_CT, HID = 32 * MLP_HT, NJ = MLP_HT / 2, KS = 2 * MLP_CT, PERIODS = 2 * NJ + 2;
    const int tok_base = blockIdx.x * MLP_TOKENS;

    for (int i = tid; i < HID; i += MLP_THREADS) { cst[i] = 0.5f * p.c1[i]; cst[HID + i] = 0.5f * p.c2b[i]; }     // halves: see gelu_erf_lockstep<true> (common.h)
    if (tid < C) cst[2 * HID + tid] = p.b2[tid];

    // ---- token tile: 48 DMA instructions of 1 KiB, six per wave (instruction Q = 6 wave + q covers rows 8 (Q / 3) .. + 7 in three
    // parts); LDS slot (row, c') holds chunk c' ^ key(row) of the row
    auto tile_slot = [&](int q, int ln, int& r, int& c) {       // row and SOURCE chunk of lane ln's 16 bytes of instruction 6 wave + q
        const int e = 64 * (q % 3) + ln, rl = e / 24;
        r = 8 * (2 * wave + q / 3) + rl;
        c = (e - 24 * rl) ^ ((r >> 1) & 7);
    };
    if constexpr (!IN_LDS) {
        const uint16_t* xg = static_cast<const uint16_t*>(p.x);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int r, c;
            tile_slot(q, lane, r, c);
            dma16(xg + (int64_t)min(tok_base + r, p.rows - 1) * p.ldx + c * 8, lds0 + MLP_WBUF + (wave * 6 + q) * 1024);
        }
    }
    // ---- weight units: unit(t) = half t & 1, tile j = t >> 1: [W2 slice j - 1 | W1 tile j] -> buffer j & 1 of the half; 24 KiB = three
    // instructions per wave (chunk e = wave, wave + 8, wave + 16 of 24; the absent parts of j = 0 / j = 6 fetch a neighbour)
    const char* w1g = static_cast<const char*>(p.w1) + lane * 16;
    const char* w2g = static_cast<const char*>(p.w2) + lane * 16;
    auto stage_unit = [&](int t) {
        const int g = t & 1, j = t >> 1;
        const int i2 = g * NJ + max(j - 1, 0), i1 = g * NJ + min(j, NJ - 1);
        const uint32_t dst = lds0 + (uint32_t)((g * 2 + (j & 1)) * 2 * MLP_IMG);
        dma16(w2g + (int64_t)i2 * MLP_IMG + wave * 1024, dst + wave * 1024);
        if (wave < 4) dma16(w2g + (int64_t)i2 * MLP_IMG + (wave + 8) * 1024, dst + (wave + 8) * 1024);
        else dma16(w1g + (int64_t)i1 * MLP_IMG + (wave - 4) * 1024, dst + MLP_IMG + (wave - 4) * 1024);
        dma16(w1g + (int64_t)i1 * MLP_IMG + (wave + 4) * 1024, dst + MLP_IMG + (wave + 4) * 1024);
    };
    stage_unit(0);
    stage_unit(1);

    const int tok = tok_base + t4 * 32 + l31;
    const bool live = IN_LDS || tok < p.rows;
    const int xrow = t4 * 32 + l31;
    const int x_off = xrow * (C * 2), xkey = (xrow >> 1) & 7;
    const int rs = (l31 & 16) | swap23(l31 & 15);              // the weight row this lane feeds to MFMA row l31
    const int w1_row = MLP_IMG + rs * (C * 2), key1 = (rs >> 1) & 7;
    const int key2 = (rs >> 2) & 3;
    const int w2_o0 = rs * 64 + ((lh ^ key2) << 4), w2_o1 = rs * 64 + (((2 + lh) ^ key2) << 4);

    U4 xf[KS];
    U4 hb0 = zero16(), hb1 = zero16();
    float k0 = 0.0f, k1 = 0.0f;
    f32x16 hacc;
    f32x16 oacc[MLP_CT];
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;
#pragma unroll
    for (int ct = 0; ct < MLP_CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ct][r] = 0.0f;

#pragma unroll 1
    for (int t = 0; t < PERIODS; ++t) {
        // unit(t) (and, at t = 0, the token tile) has landed: everything this wave issued except unit(t + 1)'s three
        if (t + 1 < PERIODS) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < PERIODS) stage_unit(t + 2);                 // into the buffer period t - 2 read
        if (t == 0) {
            // x^T fragments: lane (token, half) holds channels 16 ks + 8 half .. + 7; LayerNorm statistics over the stored values
            // (pad columns are zero): pre / 2 = k0 acc + (k1 c1[u] / 2 + c2b[u] / 2), k0 = rstd / 2, k1 = -rstd mean
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const U4*>(xt + x_off + (((2 * ks + lh) ^ xkey) << 4));
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float f[8];
                unpack8<T>(xf[ks], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s += f[j]; q = __builtin_fmaf(f[j], f[j], q); }
            }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            const float inv_c = 1.0f / (float)p.c_valid;
            const float mean = s * inv_c;
            const float var = fmaxf(q * inv_c - mean * mean, 0.0f);
            const float rstd = __builtin_amdgcn_rsqf(var + p.eps);
            k0 = 0.5f * rstd;
            k1 = -rstd * mean;
        }
        const int j = t >> 1;
        if ((t & 1) == hg) {
            // ---- multiply: second product of tile j - 1, first product of tile j
            const char* unit = smem + (hg * 2 + (j & 1)) * 2 * MLP_IMG;
            // each product's fragment reads run two groups of four ahead of the MFMAs that use them (nothing else hides the LDS
            // latency: the partner wave is busy in its VALU stream); the empty asms pin that order, which hipcc otherwise undoes
            // — two reads, two MFMAs — to save registers
            auto pin4 = [](u32x4 (&f)[4]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); };
            if (j >= 1) {
                // fragment e = 2 ct + s: output tile ct, k-step s
                auto rd = [&](int e) { return *reinterpret_cast<const u32x4*>(unit + (e >> 1) * 2048 + ((e & 1) ? w2_o1 : w2_o0)); };
                u32x4 f0[4], f1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) f0[e] = rd(e);
#pragma unroll
                for (int e = 0; e < 4; ++e) f1[e] = rd(4 + e);
                pin4(f0);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[e >> 1] = T::mfma(__builtin_bit_cast(U4, f0[e]), (e & 1) ? hb1 : hb0, oacc[e >> 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) f0[e] = rd(8 + e);
                pin4(f1);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[2 + (e >> 1)] = T::mfma(__builtin_bit_cast(U4, f1[e]), (e & 1) ? hb1 : hb0, oacc[2 + (e >> 1)]);
                pin4(f0);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[4 + (e >> 1)] = T::mfma(__builtin_bit_cast(U4, f0[e]), (e & 1) ? hb1 : hb0, oacc[4 + (e >> 1)]);
            }
            if (j < NJ) {
                auto rd = [&](int ks) { return *reinterpret_cast<const u32x4*>(unit + w1_row + (((2 * ks + lh) ^ key1) << 4)); };
                u32x4 f0[4], f1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) f0[e] = rd(e);
#pragma unroll
                for (int e = 0; e < 4; ++e) f1[e] = rd(4 + e);
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;
                pin4(f0);
#pragma unroll
                for (int e = 0; e < 4; ++e) hacc = T::mfma(__builtin_bit_cast(U4, f0[e]), xf[e], hacc);
#pragma unroll
                for (int e = 0; e < 4; ++e) f0[e] = rd(8 + e);
                pin4(f1);
#pragma unroll
                for (int e = 0; e < 4; ++e) hacc = T::mfma(__builtin_bit_cast(U4, f1[e]), xf[4 + e], hacc);
                pin4(f0);
#pragma unroll
                for (int e = 0; e < 4; ++e) hacc = T::mfma(__builtin_bit_cast(U4, f0[e]), xf[8 + e], hacc);
            }
        } else if (t >= 1 && ((t - 1) >> 1) < NJ) {
            // ---- LayerNorm fold + bias + GELU of the tile multiplied a period ago; registers 8s..8s+7 are hidden units u0 + 16 s + 0..7
            const int jv = (t - 1) >> 1;
            const float* cu = cst + (hg * NJ + jv) * 32 + 8 * lh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 ca = *reinterpret_cast<const f32x4*>(cu + 16 * s), cb = *reinterpret_cast<const f32x4*>(cu + 16 * s + 4);
                const f32x4 da = *reinterpret_cast<const f32x4*>(cu + HID + 16 * s), db = *reinterpret_cast<const f32x4*>(cu + HID + 16 * s + 4);
                const float c1v[8] = {ca[0], ca[1], ca[2], ca[3], cb[0], cb[1], cb[2], cb[3]};
                const float c2v[8] = {da[0], da[1], da[2], da[3], db[0], db[1], db[2], db[3]};
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = __builtin_fmaf(k0, hacc[8 * s + e], __builtin_fmaf(k1, c1v[e], c2v[e]));
                gelu_erf_lockstep<true>(g);
                if (s == 0) hb0 = pack8<T>(g); else hb1 = pack8<T>(g);
            }
        }
    }

    // ---- the two hidden halves meet: wave (t4, hg) keeps output tiles 3 hg .. 3 hg + 2 and hands over the other three
    __syncthreads();                                            // the weight buffers are dead: they become the exchange area
    constexpr int XCH = 3 * 16 * 64;                            // floats per wave
    float* mine = reinterpret_cast<float*>(smem) + wave * XCH;
    const float* theirs = reinterpret_cast<const float*>(smem) + (wave ^ 4) * XCH;
    auto hand_over = [&](auto HG) {
        constexpr int give0 = 3 * (1 - decltype(HG)::value);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& a = oacc[give0 + k];
                const f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                *reinterpret_cast<f32x4*>(mine + ((k * 4 + g) * 64 + lane) * 4) = v;
            }
    };
    if (hg == 0) hand_over(std::integral_constant<int, 0>{}); else hand_over(std::integral_constant<int, 1>{});
    __syncthreads();

    auto finish = [&](auto HG) {
        constexpr int keep0 = 3 * decltype(HG)::value;
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {                       // registers 8h .. 8h+7: channels 32 ct + 16 h + 8 lh + 0..7 = x fragment 2 ct + h
                const int ct = keep0 + k;
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(theirs + ((k * 4 + 2 * h) * 64 + lane) * 4);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(theirs + ((k * 4 + 2 * h + 1) * 64 + lane) * 4);
                const float* bp = cst + 2 * HID + 32 * ct + 16 * h + 8 * lh;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                float xr[8];
                unpack8<T>(xf[2 * ct + h], xr);
                const f32x16& a = oacc[ct];
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (a[8 * h + e] + o0[e]) + b0[e] + xr[e];
                    v[4 + e] = (a[8 * h + 4 + e] + o1[e]) + b1[e] + xr[4 + e];
                }
                const U4 w = pack8<T>(v);
                *reinterpret_cast<U4*>(xt + x_off + (((2 * (2 * ct + h) + lh) ^ xkey) << 4)) = w;       // in place: this lane's own chunk
                float f[8];
                unpack8<T>(w, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s += f[e]; q = __builtin_fmaf(f[e], f[e], q); }
            }
        if (!IN_LDS && p.row_stats) {           // this wave's 96 columns: slot 3 hg carries the sums, the two after it are zero
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (live && lh == 0) {
                f32x2* dst = reinterpret_cast<f32x2*>(p.row_stats) + (int64_t)tok * MLP_CT + keep0;
                const f32x2 sv = {s, q}, z = {0.0f, 0.0f};
                dst[0] = sv; dst[1] = z; dst[2] = z;
            }
        }
    };
    if (hg == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});
    __syncthreads();

    // ---- the finished tile leaves as whole rows: lane-linear LDS reads, 16-byte stores, 128-byte runs per 8 lanes
    if constexpr (!IN_LDS) {
        uint16_t* og = static_cast<uint16_t*>(p.out);
        int ln = lane;
        asm volatile("" : "+v"(ln));        // (recompute the slot arithmetic here instead of carrying the prologue's through the loop)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int r, c;
            tile_slot(q, ln, r, c);
            const U4 v = *reinterpret_cast<const U4*>(xt + (wave * 6 + q) * 1024 + ln * 16);
            if (tok_base + r < p.rows) stg16(og + (int64_t)(tok_base + r) * p.ldo + c * 8, v);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(MLP_THREADS) swin_mlp_kernel(const edtr_swin_mlp_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    swin_mlp_body<T, false>(p, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// edtr_swin_attn — x + proj(WindowAttention(LayerNorm(x))) of one Swin layer in ONE launch (reference model/swinir.py:254-279:
// norm1, cyclic shift, window partition, WindowAttention.forward :120-148 incl. the qkv and proj linears, window reverse,
// shift back, residual).  As three launches (qkv GEMM, edtr_window_attn, proj GEMM) the half layer is 26 + 29 + 15 us of
// launch-sized round trips of the token tensor and of the 3.1 x wider qkv tensor; here a workgroup owns TWO windows (128
// tokens, gathered into an LDS tile by LDS-DMA with the shift and window arithmetic applied to the row address) and nothing
// but the tile and the weight images ever moves.
//
// Wave (t4, g) of the eight owns token tile t4 (window t4 >> 1, its queries 32 (t4 & 1) .. + 31) and head group g (three
// heads).  Per head, in four barrier-separated parts that follow the weight stream (one 12 KiB image per group and part,
// LDS-DMA one part ahead):
//   q: q^T[d][t] = Wq x^T (A = weight rows, bits 2,3 of the row index swapped; B = x fragments)  -> folded LayerNorm -> B fragments
//   k: k^T likewise                                                    -> A fragments of S^T = K Q^T, published in LDS
//   v: v[t][d]  = x Wv^T (A = x fragments, B = weight rows)            -> A fragments of O^T = V^T P^T, published in LDS
//   o: S^T for both key tiles of the window (the partner tile's K / V come from LDS), + relative-position bias, + region mask,
//      softmax in registers (a query's 64 keys are 32 registers here and 32 in lane ^ 32), O^T, and the head's share of
//      proj: out^T[c][t] += Wp[c][h, d] O[t][d].
// The accumulator layout of one product is the operand layout of the next throughout (attention.hip's trick): keys sit on
// the MFMA rows of S^T in the same register pattern in which tokens sit on the rows of v, so P^T and V^T meet without a
// shuffle; the swapped weight rows make q, k, O and the output leave their accumulators in operand / storage order.
// The two head groups' partial outputs meet once, after the loop, as in edtr_swin_mlp; the result replaces x in the LDS tile
// and leaves as whole rows, scattered back through the same row map.
constexpr int SA_HEADS = 6, SA_CT = 6;                       // 6 heads x 32 (padded) columns; 6 x 32 token columns
constexpr int SA_IMG = 12288;                                // one weight image: 32 rows x 192 k (q / k / v of a head) or 192 rows x 32 k (proj)
constexpr int SA_THREADS = 512, SA_TOKENS = 128;
constexpr int SA_WBUF = 2 * 2 * SA_IMG;                      // [buffer][head group]
constexpr int SA_XT = SA_TOKENS * 384;
constexpr int SA_XCH = 48 * 1024;                            // loop: K / V fragments [window][group][key tile][k | v][step] x 1 KiB (32 KiB); after it, with WBUF: the partial outputs
constexpr int SA_CONST_FLOATS = 2 * 3 * SA_HEADS * 32 + 32 * SA_CT + 2 * SA_TOKENS;      // c1 | c2b | bproj | per-token (rstd, -rstd mean)
constexpr int SA_LDS = SA_WBUF + SA_XCH + SA_XT + SA_CONST_FLOATS * 4 + SA_TOKENS;       // + one region label per token

// KEEP (edtr_swin_layer): the finished tile stays in LDS for the MLP half; SCATTER_ONLY: nothing but the final scatter of the tile.
template <typename T, bool KEEP, bool SCATTER_ONLY = false>
__device__ __forceinline__ void swin_attn_body(const edtr_swin_attn_params& p, int total_windows, char* smem) {
    int tid = threadIdx.x;
    if constexpr (SCATTER_ONLY) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int t4 = wave & 3, g = wave >> 2, ww = t4 >> 1, tt = t4 & 1;
    char* xch = smem + SA_WBUF;
    char* xt = smem + SA_WBUF + SA_XCH;
    float* cst = reinterpret_cast<float*>(smem + SA_WBUF + SA_XCH + SA_XT);
    constexpr int NQKV = 3 * SA_HEADS * 32, C = 32 * SA_CT, KS = 2 * SA_CT;
    float* ts = cst + 2 * NQKV + C;
    uint8_t* lab = reinterpret_cast<uint8_t*>(ts + 2 * SA_TOKENS);
    const uint32_t lds0 = lds_addr_of(smem);
    const int nwx = p.W / WS, nwy = p.H / WS;

    // token n of window `win` -> row of the unshifted token matrix (the window is cut from the cyclically shifted image)
    auto token_row = [&](int win, int n) -> int64_t {
        const int wx = win % nwx, r = win / nwx, wy = r % nwy, b = r / nwy;
        int y = wy * WS + (n >> 3) + p.shift, x = wx * WS + (n & 7) + p.shift;
        if (y >= p.H) y -= p.H;
        if (x >= p.W) x -= p.W;
        return ((int64_t)b * p.H + y) * p.W + x;
    };
    const int win0 = blockIdx.x * 2;
    // LDS tile row r holds token (r & ~31) | swap(r & 31) of the tile (swap = bits 2,3 exchanged: an involution), so that a lane's
    // token row and the weight row it feeds carry the SAME swizzle key and share their twelve chunk offsets
    auto tile_token = [](int r) { return (r & ~15) | swap23(r & 15); };
    auto tile_slot = [&](int q, int ln, int& r, int& c) {       // tile row and SOURCE chunk of lane ln's 16 bytes of DMA instruction 6 wave + q
        const int e = 64 * (q % 3) + ln, rl = e / 24;
        r = 8 * (2 * wave + q / 3) + rl;
        c = (e - 24 * rl) ^ ((r >> 1) & 7);
    };

    if constexpr (!SCATTER_ONLY) {
    for (int i = tid; i < NQKV; i += SA_THREADS) { cst[i] = p.c1[i]; cst[NQKV + i] = p.c2b[i]; }
    if (tid < C) cst[2 * NQKV + tid] = p.bproj[tid];
    if (tid < SA_TOKENS) {
        uint8_t v = 0;
        if (p.labels) {
            const int win = min(win0 + (tid >> 6), total_windows - 1), n = tid & 63;
            const int wx = win % nwx, wy = (win / nwx) % nwy;
            v = p.labels[(int64_t)(wy * WS + (n >> 3)) * p.W + wx * WS + (n & 7)];
        }
        lab[tid] = v;
    }
    {
        const uint16_t* xg = static_cast<const uint16_t*>(p.x);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int r, c;
            tile_slot(q, lane, r, c);
            const int64_t row = token_row(min(win0 + (r >> 6), total_windows - 1), tile_token(r) & 63);
            dma16(xg + row * p.ldx + c * 8, lds0 + SA_WBUF + SA_XCH + (wave * 6 + q) * 1024);
        }
    }
    // weight unit n = 4 i + part (head 3 g' + i of each group g'; part 0 / 1 / 2 = the q / k / v image, 3 = the proj slice): 24 KiB,
    // three instructions per wave (chunk e = wave, wave + 8, wave + 16 of 24: e < 12 belongs to group 0)
    const int lane16 = lane * 16;
    auto stage_unit = [&](int n) {
        const int i = n >> 2, part = n & 3;
        const uint32_t dst = lds0 + (uint32_t)(n & 1) * (2 * SA_IMG);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = wave + 8 * k, gg = e >= 12 ? 1 : 0, off = (e - 12 * gg) * 1024;
            const int h = 3 * gg + i;
            const char* src = part < 3 ? static_cast<const char*>(p.wqkv) + (int64_t)(h * 3 + part) * SA_IMG : static_cast<const char*>(p.wproj) + (int64_t)h * SA_IMG;
            dma16(src + off + lane16, dst + e * 1024);
        }
    };
    stage_unit(0);

    const int rs = (l31 & 16) | swap23(l31 & 15);              // the weight row this lane feeds to MFMA row / column l31 ...
    const int xrow = t4 * 32 + l31;                             // this lane's token in the tile ...
    const int w1_row = rs * (C * 2), key1 = (rs >> 1) & 7;     // ... which sits in tile row t4 * 32 + rs: same key
    char* const xt4 = xt + t4 * 32 * (C * 2);                  // this wave's 32 rows of the tile (uniform)
    // chunk 2 ks + lh of a 384-byte row: the XOR key touches the low three chunk bits only and 2 ks + lh = 2 ks ^ lh, so ONE
    // offset serves all twelve k-steps: (ks >> 2) * 128 + (perm0 ^ 32 (ks & 3))
    const int perm0 = (lh ^ key1) << 4;
    auto chunk = [&](int ks) { return (ks >> 2) * 128 + (perm0 ^ (32 * (ks & 3))); };
    auto xfrag = [&](int ks) { return *reinterpret_cast<const U4*>(xt4 + w1_row + chunk(ks)); };
    auto wfrag = [&](const char* img, int ks) { return *reinterpret_cast<const U4*>(img + w1_row + chunk(ks)); };
    // a 32 x 32 x 192 product of a weight image with this wave's token tile: fragment reads run one group of four k-steps ahead of
    // the MFMAs (pinned: hipcc would otherwise issue all 24 reads up front — 96 registers at the kernel's pressure peak);
    // W_ROWS: the weight rows are the MFMA rows (q^T, k^T), else the tokens are (v)
    auto pin8f = [](u32x4 (&f)[8]) {
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    };
    auto project = [&](const char* img, f32x16& acc, auto W_ROWS) {
        u32x4 f0[8], f1[8];          // [0..3] weight, [4..7] token fragments of four k-steps
        auto load4 = [&](u32x4 (&f)[8], int ks0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = *reinterpret_cast<const u32x4*>(img + w1_row + chunk(ks0 + e));
                f[4 + e] = *reinterpret_cast<const u32x4*>(xt4 + w1_row + chunk(ks0 + e));
            }
        };
        auto mul4 = [&](u32x4 (&f)[8]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const U4 w = __builtin_bit_cast(U4, f[e]), x = __builtin_bit_cast(U4, f[4 + e]);
                if constexpr (decltype(W_ROWS)::value) acc = T::mfma(w, x, acc);
                else acc = T::mfma(x, w, acc);
            }
        };
        load4(f0, 0);
        load4(f1, 4);
        pin8f(f0);
        mul4(f0);
        load4(f0, 8);
        pin8f(f1);
        mul4(f1);
        pin8f(f0);
        mul4(f0);
    };
    const int key2 = (rs >> 2) & 3;
    const int w2_o0 = rs * 64 + ((lh ^ key2) << 4);           // k-step 1 is the chunk two further: offset ^ 32
    const int qn = 32 * tt + l31;                               // this lane's query within its window
    char* const my_x0 = xch + (((ww * 2 + g) * 2 + tt) * 4) * 1024;               // [k s0, k s1, v s0, v s1] (uniform; + lane16)
    const char* const win_x0 = xch + ((ww * 2 + g) * 2) * 4 * 1024;               // + key tile * 4096 + (k | v) * 2048 + step * 1024

    float k0 = 0.0f, k1 = 0.0f;
    U4 qf[2] = {zero16(), zero16()};
    f32x16 oacc[SA_CT];
#pragma unroll
    for (int ct = 0; ct < SA_CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ct][r] = 0.0f;

#pragma unroll 1
    for (int n = 0; n < 4 * 3; ++n) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of unit n (and, at n = 0, of the token tile) has landed
        __syncthreads();                                        // ... everyone's, and the other buffer is free
        if (n + 1 < 12) stage_unit(n + 1);
        if (n == 0) {
            // LayerNorm statistics of this lane's token over the stored values (pad columns are zero)
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float f[8];
                unpack8<T>(xfrag(ks), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s += f[j]; q = __builtin_fmaf(f[j], f[j], q); }
            }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            // (through an SGPR: as a hoisted VGPR value the reciprocal is one more register carried through the loop)
            const float inv_c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / (float)p.c_valid)));
            const float mean = s * inv_c;
            const float var = fmaxf(q * inv_c - mean * mean, 0.0f);
            k0 = __builtin_amdgcn_rsqf(var + p.eps);
            k1 = -k0 * mean;
            if (g == 0 && lh == 0) { ts[2 * xrow] = k0; ts[2 * xrow + 1] = k1; }      // (read in the v parts: two barriers from here)
        }
        const int i = n >> 2, part = n & 3, h = 3 * g + i;
        const char* img = smem + (n & 1) * (2 * SA_IMG) + g * SA_IMG;
        if (part < 2) {
            // ---- q^T / k^T [d][t]: folded LayerNorm per lane (token), constants per register (d = 16 s + 8 lh + j)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            project(img, acc, std::true_type{});
            const float* cu = cst + (part * SA_HEADS + h) * 32 + 8 * lh;
            U4 f2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 ca = *reinterpret_cast<const f32x4*>(cu + 16 * s), cb = *reinterpret_cast<const f32x4*>(cu + 16 * s + 4);
                const f32x4 da = *reinterpret_cast<const f32x4*>(cu + NQKV + 16 * s), db = *reinterpret_cast<const f32x4*>(cu + NQKV + 16 * s + 4);
                const float c1v[8] = {ca[0], ca[1], ca[2], ca[3], cb[0], cb[1], cb[2], cb[3]};
                const float c2v[8] = {da[0], da[1], da[2], da[3], db[0], db[1], db[2], db[3]};
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(k0, acc[8 * s + e], __builtin_fmaf(k1, c1v[e], c2v[e]));
                f2[s] = pack8<T>(v);
            }
            if (part == 0) { qf[0] = f2[0]; qf[1] = f2[1]; }
            else { *reinterpret_cast<U4*>(my_x0 + lane16) = f2[0]; *reinterpret_cast<U4*>(my_x0 + lane16 + 1024) = f2[1]; }
        } else if (part == 2) {
            // ---- v [t][d]: tokens on the rows (per register), this lane's column is head channel rs
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            project(img, acc, std::false_type{});
            const float c1v = cst[(2 * SA_HEADS + h) * 32 + rs], c2v = cst[NQKV + (2 * SA_HEADS + h) * 32 + rs];
            const float* tsw = ts + 2 * (t4 * 32 + 4 * lh);
            U4 f2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {               // registers 8 s + 4 gq + e: tokens 16 s + 8 gq + 4 lh + e
                    const f32x4 a = *reinterpret_cast<const f32x4*>(tsw + 2 * (16 * s + 8 * gq));
                    const f32x4 b = *reinterpret_cast<const f32x4*>(tsw + 2 * (16 * s + 8 * gq) + 4);
                    const float r0[4] = {a[0], a[2], b[0], b[2]}, r1[4] = {a[1], a[3], b[1], b[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * gq + e] = __builtin_fmaf(r0[e], acc[8 * s + 4 * gq + e], __builtin_fmaf(r1[e], c1v, c2v));
                }
                f2[s] = pack8<T>(v);
            }
            *reinterpret_cast<U4*>(my_x0 + lane16 + 2048) = f2[0];
            *reinterpret_cast<U4*>(my_x0 + lane16 + 3072) = f2[1];
        } else {
            // ---- scores of this tile's 32 queries against the window's 64 keys; register r of sc[kt] is key 32 kt + 8 (r >> 2) + 4 lh + (r & 3)
            f32x16 sc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[kt][r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 2; ++s) sc[kt] = T::mfma(*reinterpret_cast<const U4*>(win_x0 + lane16 + kt * 4096 + s * 1024), qf[s], sc[kt]);
            }
            int qo = qn;
            asm volatile("" : "+v"(qo));                        // (the per-lane bias address is formed here, not carried through the loop)
            const float* bias_h = p.bias + ((int64_t)h * 16 * 64 + qo) * 4;
            const uint32_t qlab = lab[ww * 64 + qo];
            float m = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int kg = 8 * kt + 2 * gq + lh;                          // keys 4 kg .. 4 kg + 3
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + kg * 64 * 4);
                    const uint32_t kl = *reinterpret_cast<const uint32_t*>(lab + ww * 64 + 4 * kg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = sc[kt][4 * gq + e] + b[e];
                        if (p.labels && ((kl >> (8 * e)) & 0xffu) != qlab) v -= 100.0f;
                        sc[kt][4 * gq + e] = v;
                        m = fmaxf(m, v);
                    }
                }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float mc = m * 1.4426950408889634f;
            float l = 0.0f;
            U4 pf[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                float pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][r], 1.4426950408889634f, -mc));
                    l += pr[r];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float v[8] = {pr[8 * s], pr[8 * s + 1], pr[8 * s + 2], pr[8 * s + 3], pr[8 * s + 4], pr[8 * s + 5], pr[8 * s + 6], pr[8 * s + 7]};
                    pf[kt][s] = pack8<T>(v);
                }
            }
            l += __shfl_xor(l, 32, 64);
            const float inv_l = 1.0f / l;
            // ---- O^T[d][q] = V^T P^T, then this head's share of proj
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) o = T::mfma(*reinterpret_cast<const U4*>(win_x0 + lane16 + kt * 4096 + 2048 + s * 1024), pf[kt][s], o);
            U4 of[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = o[8 * s + e] * inv_l;
                of[s] = pack8<T>(v);
            }
#pragma unroll
            for (int ct = 0; ct < SA_CT; ++ct) {
                oacc[ct] = T::mfma(*reinterpret_cast<const U4*>(img + ct * 2048 + w2_o0), of[0], oacc[ct]);
                oacc[ct] = T::mfma(*reinterpret_cast<const U4*>(img + ct * 2048 + (w2_o0 ^ 32)), of[1], oacc[ct]);
            }
        }
    }

    // ---- the two head groups meet: wave (t4, g) keeps output tiles 3 g .. 3 g + 2 and hands over the other three
    __syncthreads();                                            // weight buffers and K / V slots are dead: they become the exchange area
    constexpr int XCH = 3 * 16 * 64;                            // floats per wave
    int l4 = lane16 >> 2;
    asm volatile("" : "+v"(l4));
    float* mine = reinterpret_cast<float*>(smem) + wave * XCH + l4;
    const float* theirs = reinterpret_cast<const float*>(smem) + (wave ^ 4) * XCH + l4;
    auto hand_over = [&](auto G) {
        constexpr int give0 = 3 * (1 - decltype(G)::value);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x16& a = oacc[give0 + k];
                const f32x4 v = {a[4 * gq], a[4 * gq + 1], a[4 * gq + 2], a[4 * gq + 3]};
                *reinterpret_cast<f32x4*>(mine + (k * 4 + gq) * 256) = v;
            }
    };
    if (g == 0) hand_over(std::integral_constant<int, 0>{}); else hand_over(std::integral_constant<int, 1>{});
    __syncthreads();
    auto finish = [&](auto G) {
        constexpr int keep0 = 3 * decltype(G)::value;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                    // registers 8 hh .. 8 hh + 7: channels 32 ct + 16 hh + 8 lh + 0..7 = x fragment 2 ct + hh
                const int ct = keep0 + k;
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(theirs + (k * 4 + 2 * hh) * 256);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(theirs + (k * 4 + 2 * hh + 1) * 256);
                const float* bp = cst + 2 * NQKV + 32 * ct + 16 * hh + 8 * (l4 >> 7);      // (lane half from the re-derived lane offset)
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                char* slot = xt4 + w1_row + chunk(2 * ct + hh);
                float xr[8];
                unpack8<T>(*reinterpret_cast<const U4*>(slot), xr);
                const f32x16& a = oacc[ct];
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (a[8 * hh + e] + o0[e]) + b0[e] + xr[e];
                    v[4 + e] = (a[8 * hh + 4 + e] + o1[e]) + b1[e] + xr[4 + e];
                }
                *reinterpret_cast<U4*>(slot) = pack8<T>(v);    // in place: this lane's own chunk
            }
    };
    if (g == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});
    __syncthreads();
    }       // !SCATTER_ONLY

    // ---- the finished tile leaves as whole rows, scattered back through the row map
    if constexpr (!KEEP) {
        uint16_t* og = static_cast<uint16_t*>(p.out);
        int ln = lane;
        asm volatile("" : "+v"(ln));        // (recompute the slot arithmetic here instead of carrying the prologue's through the loop)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int r, c;
            tile_slot(q, ln, r, c);
            const U4 v = *reinterpret_cast<const U4*>(xt + (wave * 6 + q) * 1024 + ln * 16);
            const int win = win0 + (r >> 6);
            if (win < total_windows) stg16(og + token_row(win, tile_token(r) & 63) * p.ldo + c * 8, v);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(SA_THREADS) swin_attn_kernel(const edtr_swin_attn_params p, int total_windows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    swin_attn_body<T, false>(p, total_windows, smem);
}

// edtr_swin_layer — a whole Swin layer in one launch: the attention half leaves its finished token tile in LDS, the MLP half runs on
// it in place (same tile address, same row swizzle; it is a per-token operation, so the tile's window order does not matter), and
// the result is scattered back through the attention half's row map: one launch, one store and one fetch of the token tensor less
// per layer than edtr_swin_attn + edtr_swin_mlp.
static_assert(SA_WBUF + SA_XCH == MLP_WBUF && SA_XT == MLP_XT, "the two halves keep the token tile at the same LDS address");
constexpr int SL_LDS = SA_LDS > MLP_LDS ? SA_LDS : MLP_LDS;

template <typename T>
__global__ void __launch_bounds__(SA_THREADS) swin_layer_kernel(const edtr_swin_attn_params pa, const edtr_swin_mlp_params pm, int total_windows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    swin_attn_body<T, true>(pa, total_windows, smem);
    swin_mlp_body<T, true>(pm, smem);
    swin_attn_body<T, false, true>(pa, total_windows, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// edtr_conv64 — 3 x 3 / stride 1 / pad 1 convolution of a 64-channel image into <= 64 channels (SwinIR's reconstruction tail:
// conv_up1..3 behind a nearest-2x upsample, conv_hr, conv_last; reference model/swinir.py:776-787, :878-886).  On edtr_igemm these
// run on 128-column tiles that are half padding, one launch-sized workgroup per 128 pixels, at 310 - 330 TFLOP/s — 0.5 ms for
// the 512 x 512 level of a batch of 8, whose traffic is a 110 us job.  Here the workgroups are PERSISTENT (one per CU): the nine
// 64 x 64 tap matrices stay in LDS (72 KiB) for the kernel's life and the image streams through as 16 x 16-pixel patches with
// their 1-pixel halo (18 x 18 pixels x 128 B, two buffers, LDS-DMA one patch ahead; the nearest-2x upsample is the DMA's
// source address, zero padding a 16-byte block of zeros).  The product runs transposed, out^T[n][pixel] = W[n][k] x^T[k][pixel],
// so that a lane owns a pixel and eight consecutive channels per register group (weight rows with bits 2,3 swapped): results
// leave as 16-byte stores (or, for the network's last convolution, as fp32 NCHW planes).  Wave w owns patch rows 2 w, 2 w + 1
// (32 pixels) x 64 channels: per tap and 16-channel k-step one patch fragment feeds two MFMAs.
// Patch image: pixel (py, px) at (18 py + px) * 128 B, 16-byte chunk c in slot c ^ ((px >> 1) & 7): the 16 pixels of a
// ds_read_b128 lane group are 16 different columns (mod 16) for every tap — conflict-free; weights as tile_off (common.h).
constexpr int C64_THREADS = 512;
constexpr int C64_W_BYTES = 9 * 64 * 128;                    // 73728
constexpr int C64_PATCH_INSTRS = 41;                         // 18 * 18 * 128 B = 40.5 KiB: 41 DMA instructions, the last one half used
constexpr int C64_PATCH_BYTES = C64_PATCH_INSTRS * 1024;
constexpr int C64_LDS = C64_W_BYTES + 2 * C64_PATCH_BYTES + 64 * 4;
__device__ __attribute__((aligned(16))) uint32_t g_c64_zero[4];

template <typename T>
__global__ void __launch_bounds__(C64_THREADS) conv64_kernel(const edtr_conv64_params p, int patches) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char* wl = smem;
    char* pb = smem + C64_W_BYTES;
    float* bias_l = reinterpret_cast<float*>(smem + C64_W_BYTES + 2 * C64_PATCH_BYTES);
    const uint32_t lds0 = lds_addr_of(smem);
    const int tiles_x = p.W >> 4, tiles_y = p.H >> 4;
    const int SH = p.upsample2x ? p.H >> 1 : p.H, SW = p.upsample2x ? p.W >> 1 : p.W;        // source size

    if (tid < 64) bias_l[tid] = p.bias[tid];
    // weights: 72 instructions of 1 KiB, nine per wave (the images are already in LDS layout)
#pragma unroll
    for (int q = 0; q < 9; ++q) dma16(static_cast<const char*>(p.w) + (wave * 9 + q) * 1024 + lane * 16, lds0 + (wave * 9 + q) * 1024);

    // this lane's share of a patch fetch: instruction Q = wave + 8 q (< 41), 16 bytes at byte o = 1024 Q + 16 lane of the patch image
    int f_py[6], f_px[6], f_c[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int o = (wave + 8 * q) * 1024 + lane * 16, P = o >> 7, cs = (o >> 4) & 7;
        f_py[q] = P / 18;
        f_px[q] = P - 18 * f_py[q];
        f_c[q] = cs ^ ((f_px[q] >> 1) & 7);
    }
    const uint16_t* xg = static_cast<const uint16_t*>(p.x);
    auto fetch = [&](int patch, int buf) {
        const int tx = patch % tiles_x, r = patch / tiles_x, ty = r % tiles_y, b = r / tiles_y;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (wave + 8 * q < C64_PATCH_INSTRS) {           // (wave-uniform)
                const int Y = ty * 16 - 1 + f_py[q], X = tx * 16 - 1 + f_px[q];      // output-resolution coordinates of the patch pixel
                const bool in = f_py[q] < 18 && Y >= 0 && Y < p.H && X >= 0 && X < p.W;
                const int sy = p.upsample2x ? Y >> 1 : Y, sx = p.upsample2x ? X >> 1 : X;
                const void* src = in ? static_cast<const void*>(xg + (((int64_t)b * SH + sy) * SW + sx) * p.ldx + f_c[q] * 8)
                                     : static_cast<const void*>(g_c64_zero);
                dma16(src, lds0 + C64_W_BYTES + buf * C64_PATCH_BYTES + (wave + 8 * q) * 1024);
            }
        }
    };
    int patch = blockIdx.x;
    if (patch < patches) fetch(patch, 0);

    // fragment addressing: this lane's pixel (y, x) = (2 wave + (l31 >> 4), l31 & 15) of the patch; weight row rs of either 32-row tile
    const int px_x = l31 & 15;
    const int pix0 = (18 * (2 * wave + (l31 >> 4)) + px_x) * 128;
    int xperm[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xperm[kx] = (lh ^ (((px_x + kx) >> 1) & 7)) << 4;
    const int rs = (l31 & 16) | swap23(l31 & 15);
    const int w_off = rs * 128 + ((lh ^ ((rs >> 1) & 7)) << 4);
    uint16_t* og = static_cast<uint16_t*>(p.out);
    float* of32 = static_cast<float*>(p.out);

    int it = 0;
#pragma unroll 1
    for (; patch < patches; patch += gridDim.x, ++it) {
        // the fetch of this patch (and, the first time, the weights) has landed; the previous patch's stores may still be in flight
        // (counted: the fetch was issued BEFORE that patch's 4 — fp32 planes: n_valid >= 1 — store instructions)
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (p.out_nchw_f32) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        const int next = patch + gridDim.x;
        if (next < patches) fetch(next, (it + 1) & 1);
        const char* pbuf = pb + (it & 1) * C64_PATCH_BYTES + pix0;

        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const U4 xb = *reinterpret_cast<const U4*>(pbuf + (18 * ky + kx) * 128 + (xperm[kx] ^ (32 * ks)));
                const U4 w0 = *reinterpret_cast<const U4*>(wl + t * 8192 + (w_off ^ (32 * ks)));
                const U4 w1 = *reinterpret_cast<const U4*>(wl + t * 8192 + 4096 + (w_off ^ (32 * ks)));
                acc[0] = T::mfma(w0, xb, acc[0]);
                acc[1] = T::mfma(w1, xb, acc[1]);
            }
        }

        // ---- epilogue: registers 8 s .. 8 s + 7 of acc[nt] are channels 32 nt + 16 s + 8 lh + 0..7 of this lane's pixel
        const int tx = patch % tiles_x, r0 = patch / tiles_x, ty = r0 % tiles_y, b = r0 / tiles_y;
        const int Y = ty * 16 + 2 * wave + (l31 >> 4), X = tx * 16 + px_x;
        if (!p.out_nchw_f32) {
            uint16_t* orow = og + (((int64_t)b * p.H + Y) * p.W + X) * p.ldo + 8 * lh;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float* bp = bias_l + 32 * nt + 16 * s + 8 * lh;
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = __builtin_fmaf(p.alpha, acc[nt][8 * s + e], b0[e]);
                        v[4 + e] = __builtin_fmaf(p.alpha, acc[nt][8 * s + 4 + e], b1[e]);
                    }
                    if (p.act == EDTR_ACT_LRELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * p.act_slope);
                    }
                    stg16(orow + 32 * nt + 16 * s, pack8<T>(v));
                }
        } else {
            // fp32 planes [b][c][Y][X], c < n_valid <= 4: channels 0..3 are registers 0..3 of acc[0] in the lanes with lh == 0
            const int64_t plane = (int64_t)p.H * p.W;
            float* o0 = of32 + (int64_t)b * p.n_valid * plane + (int64_t)Y * p.W + X;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = __builtin_fmaf(p.alpha, acc[0][c], bias_l[c]);
                if (p.act == EDTR_ACT_LRELU) v = fmaxf(v, v * p.act_slope);
                if (lh == 0 && c < p.n_valid) o0[c * plane] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// edtr_conv128_out — the VAE decoder's last step: GroupNorm + SiLU, 3 x 3 convolution of the 128-channel image into <= 4 channels,
// fp32 NCHW result (reference model/vae.py:553-560: norm_out, nonlinearity, conv_out; the caller's NHWC -> NCHW).  As launches:
// edtr_gn_apply (213 us for a batch of 8 at 512 x 512: one more round trip of the 537 MB tensor), edtr_igemm's 256 x 32 tile (366 us
// at 105 TFLOP/s: 29 of its 32 columns are padding) and the layout kernel.  Here, edtr_conv64's structure with a 256-byte pixel:
// persistent workgroups, the nine tap matrices (32 rows x 128 k) resident in LDS, 16 x 16-pixel patches with halo streamed through
// ONE 81-KiB buffer (two do not fit), every lane normalising the 16 bytes it fetched — silu(x scale[c] + shift[c]) from the
// image's (scale, shift) table (edtr_gn_table), pixels outside the image stay zero — before the patch is multiplied.
// Pixel (py, px) at (18 py + px) * 256 B, chunk c in slot c ^ (px & 15); weight row n at n * 256 B, chunk c in slot c ^ (n & 15).
constexpr int CO_THREADS = 512;
constexpr int CO_W_BYTES = 9 * 32 * 256;                     // 73728
constexpr int CO_PATCH_INSTRS = 81;                          // 18 * 18 * 256 B = 81 KiB
constexpr int CO_PATCH_BYTES = CO_PATCH_INSTRS * 1024;
constexpr int CO_TBL = CO_W_BYTES + CO_PATCH_BYTES;          // (scale, shift) of the patch's image: 128 x 2 floats
constexpr int CO_LDS = CO_TBL + 1024 + 32 * 4;

template <typename T>
__global__ void __launch_bounds__(CO_THREADS) conv128_out_kernel(const edtr_conv128_out_params p, int patches) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char* wl = smem;
    char* pb = smem + CO_W_BYTES;
    float* bias_l = reinterpret_cast<float*>(smem + CO_TBL + 1024);
    const uint32_t lds0 = lds_addr_of(smem);
    const int tiles_x = p.W >> 4, tiles_y = p.H >> 4;

    if (tid < 32) bias_l[tid] = p.bias[tid];
#pragma unroll
    for (int q = 0; q < 9; ++q) dma16(static_cast<const char*>(p.w) + (wave * 9 + q) * 1024 + lane * 16, lds0 + (wave * 9 + q) * 1024);

    // this lane's share of a patch fetch: instruction Q = wave + 8 q (< 81), 16 bytes at byte o = 1024 Q + 16 lane of the patch image;
    // packed (py << 9 | px << 4 | source chunk)
    int f_geo[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) {
        const int o = (wave + 8 * q) * 1024 + lane * 16, P = o >> 8, cs = (o >> 4) & 15;
        const int py = P / 18, px = P - 18 * py;
        f_geo[q] = (py << 9) | (px << 4) | (cs ^ (px & 15));
    }
    const uint16_t* xg = static_cast<const uint16_t*>(p.x);

    const int px_x = l31 & 15;
    const int pix0 = (18 * (2 * wave + (l31 >> 4)) + px_x) * 256;
    int xperm[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xperm[kx] = (lh ^ ((px_x + kx) & 15)) << 4;
    const int rs = (l31 & 16) | swap23(l31 & 15);
    const int w_off = rs * 256 + ((lh ^ (rs & 15)) << 4);
    float* of32 = static_cast<float*>(p.out);

#pragma unroll 1
    for (int patch = blockIdx.x; patch < patches; patch += gridDim.x) {
        const int tx = patch % tiles_x, r0 = patch / tiles_x, ty = r0 % tiles_y, b = r0 / tiles_y;
        __syncthreads();                                        // everyone is done with the previous patch (and its table)
        if (p.gn_table) dma16(reinterpret_cast<const char*>(p.gn_table + (int64_t)b * 256) + lane * 16, lds0 + CO_TBL);   // (every wave: the same KiB)
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            if (wave + 8 * q < CO_PATCH_INSTRS) {               // (wave-uniform)
                const int py = f_geo[q] >> 9, px = (f_geo[q] >> 4) & 31, c = f_geo[q] & 15;
                const int Y = ty * 16 - 1 + py, X = tx * 16 - 1 + px;
                const bool in = Y >= 0 && Y < p.H && X >= 0 && X < p.W;
                const void* src = in ? static_cast<const void*>(xg + (((int64_t)b * p.H + Y) * p.W + X) * p.ldx + c * 8)
                                     : static_cast<const void*>(g_c64_zero);
                dma16(src, lds0 + CO_W_BYTES + (wave + 8 * q) * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (also the previous patch's stores, issued before these fetches)
        if (p.gn_table) {
            // GroupNorm apply + SiLU of this lane's own 16 bytes of every piece, in place (edtr_gn_apply's arithmetic and rounding)
            const float* tbl = reinterpret_cast<const float*>(smem + CO_TBL);
#pragma unroll
            for (int q = 0; q < 11; ++q) {
                if (wave + 8 * q < CO_PATCH_INSTRS) {
                    const int py = f_geo[q] >> 9, px = (f_geo[q] >> 4) & 31, c = f_geo[q] & 15;
                    const int Y = ty * 16 - 1 + py, X = tx * 16 - 1 + px;
                    if (Y >= 0 && Y < p.H && X >= 0 && X < p.W) {
                        char* q16 = smem + CO_W_BYTES + (wave + 8 * q) * 1024 + lane * 16;
                        const float* tb = tbl + c * 16;
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb), t1 = *reinterpret_cast<const f32x4*>(tb + 4);
                        const f32x4 t2 = *reinterpret_cast<const f32x4*>(tb + 8), t3 = *reinterpret_cast<const f32x4*>(tb + 12);
                        float f[8], e[8];
                        unpack8<T>(*reinterpret_cast<const U4*>(q16), f);
                        f[0] = f[0] * t0[0] + t0[1]; f[1] = f[1] * t0[2] + t0[3];
                        f[2] = f[2] * t1[0] + t1[1]; f[3] = f[3] * t1[2] + t1[3];
                        f[4] = f[4] * t2[0] + t2[1]; f[5] = f[5] * t2[2] + t2[3];
                        f[6] = f[6] * t3[0] + t3[1]; f[7] = f[7] * t3[2] + t3[3];
                        pin8(f);                                 // silu_f in lockstep (common.h: gelu_erf_lockstep)
#pragma unroll
                        for (int i = 0; i < 8; ++i) e[i] = f[i] * -1.4426950408889634f;
                        pin8(e);
#pragma unroll
                        for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
                        pin8(e);
#pragma unroll
                        for (int i = 0; i < 8; ++i) e[i] = 1.0f + e[i];
                        pin8(e);
#pragma unroll
                        for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
                        pin8(e);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = f[i] * e[i];
                        *reinterpret_cast<U4*>(q16) = pack8<T>(f);
                    }
                }
            }
        }
        __syncthreads();
        const char* pbuf = pb + pix0;

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const U4 xb = *reinterpret_cast<const U4*>(pbuf + (18 * ky + kx) * 256 + (xperm[kx] ^ (32 * ks)));
                const U4 w0 = *reinterpret_cast<const U4*>(wl + t * 8192 + (w_off ^ (32 * ks)));
                acc = T::mfma(w0, xb, acc);
            }
        }
        // fp32 planes [b][c][Y][X], c < n_valid <= 4: channels 0..3 are registers 0..3 in the lanes with lh == 0
        const int Y = ty * 16 + 2 * wave + (l31 >> 4), X = tx * 16 + px_x;
        const int64_t plane = (int64_t)p.H * p.W;
        float* o0 = of32 + (int64_t)b * p.n_valid * plane + (int64_t)Y * p.W + X;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v = __builtin_fmaf(p.alpha, acc[c], bias_l[c]);
            if (lh == 0 && c < p.n_valid) o0[c * plane] = v;
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

extern "C" int edtr_swin_mlp(const edtr_swin_mlp_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_swin_mlp_params& p = *pp;
    if (!p.x || !p.w1 || !p.w2 || !p.c1 || !p.c2b || !p.b2 || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.rows <= 0 || p.c_valid <= 0 || p.c_valid > p.C) return EDTR_E_SHAPE;
    if (p.C != 32 * MLP_CT || p.hidden != 32 * MLP_HT) return EDTR_E_UNSUPPORTED;      // the shipped SwinIR width (180 -> 192, 360 -> 384)
    if (p.ldx < p.C || p.ldo < p.C) return EDTR_E_SHAPE;
    if ((p.ldx & 7) || (p.ldo & 7)) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || !aligned16(p.w1) || !aligned16(p.w2) || !aligned16(p.out) || !aligned16(p.c1) || !aligned16(p.c2b) ||
        !aligned16(p.b2) || (p.row_stats && !aligned16(p.row_stats)))
        return EDTR_E_ALIGN;
    static EdtrLdsOnce attr_set[2];
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((p.rows + MLP_TOKENS - 1) / MLP_TOKENS));
    if (p.dtype == EDTR_BF16) {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&swin_mlp_kernel<BF16>), MLP_LDS, attr_set[0])) return rc_;
        hipLaunchKernelGGL(swin_mlp_kernel<BF16>, grid, dim3(MLP_THREADS), MLP_LDS, s, p);
    } else {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&swin_mlp_kernel<F16>), MLP_LDS, attr_set[1])) return rc_;
        hipLaunchKernelGGL(swin_mlp_kernel<F16>, grid, dim3(MLP_THREADS), MLP_LDS, s, p);
    }
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_swin_attn(const edtr_swin_attn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_swin_attn_params& p = *pp;
    if (!p.x || !p.wqkv || !p.wproj || !p.c1 || !p.c2b || !p.bproj || !p.bias || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || (p.H % WS) || (p.W % WS) || p.c_valid <= 0 || p.c_valid > p.C) return EDTR_E_SHAPE;
    if (p.C != 32 * SA_CT || p.heads != SA_HEADS || p.head_dim > HP || p.head_dim <= 0) return EDTR_E_UNSUPPORTED;   // the shipped SwinIR width
    if (p.shift < 0 || p.shift >= WS) return EDTR_E_SHAPE;
    if (p.shift > 0 && !p.labels) return EDTR_E_NULL;
    if (p.x == p.out) return EDTR_E_UNSUPPORTED;               // other workgroups gather rows this one scatters
    if (p.ldx < p.C || p.ldo < p.C) return EDTR_E_SHAPE;
    if ((p.ldx & 7) || (p.ldo & 7) || (p.W & 3)) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || !aligned16(p.wqkv) || !aligned16(p.wproj) || !aligned16(p.out) || !aligned16(p.c1) || !aligned16(p.c2b) ||
        !aligned16(p.bproj) || !aligned16(p.bias) || (p.labels && (reinterpret_cast<uintptr_t>(p.labels) & 3u)))
        return EDTR_E_ALIGN;
    const int64_t windows = (int64_t)p.B * (p.H / WS) * (p.W / WS);
    if (windows > 0x3fffffffLL) return EDTR_E_UNSUPPORTED;
    static EdtrLdsOnce attr_set[2];
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((windows + 1) / 2));
    if (p.dtype == EDTR_BF16) {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&swin_attn_kernel<BF16>), SA_LDS, attr_set[0])) return rc_;
        hipLaunchKernelGGL(swin_attn_kernel<BF16>, grid, dim3(SA_THREADS), SA_LDS, s, p, (int)windows);
    } else {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&swin_attn_kernel<F16>), SA_LDS, attr_set[1])) return rc_;
        hipLaunchKernelGGL(swin_attn_kernel<F16>, grid, dim3(SA_THREADS), SA_LDS, s, p, (int)windows);
    }
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_swin_layer(const edtr_swin_attn_params* pa, const edtr_swin_mlp_params* pm, edtr_stream_t stream) {
    if (!pa || !pm) return EDTR_E_NULL;
    const edtr_swin_attn_params& a = *pa;
    const edtr_swin_mlp_params& m = *pm;
    if (!a.x || !a.wqkv || !a.wproj || !a.c1 || !a.c2b || !a.bproj || !a.bias || !a.out) return EDTR_E_NULL;
    if (!m.w1 || !m.w2 || !m.c1 || !m.c2b || !m.b2) return EDTR_E_NULL;
    if ((a.dtype != EDTR_BF16 && a.dtype != EDTR_F16) || m.dtype != a.dtype) return EDTR_E_DTYPE;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0 || (a.H % WS) || (a.W % WS) || a.c_valid <= 0 || a.c_valid > a.C || m.c_valid != a.c_valid) return EDTR_E_SHAPE;
    if (a.C != 32 * SA_CT || a.heads != SA_HEADS || a.head_dim > HP || a.head_dim <= 0 || m.C != a.C || m.hidden != 32 * MLP_HT) return EDTR_E_UNSUPPORTED;
    if (a.shift < 0 || a.shift >= WS) return EDTR_E_SHAPE;
    if (a.shift > 0 && !a.labels) return EDTR_E_NULL;
    if (a.x == a.out) return EDTR_E_UNSUPPORTED;
    if (a.ldx < a.C || a.ldo < a.C) return EDTR_E_SHAPE;
    if ((a.ldx & 7) || (a.ldo & 7) || (a.W & 3)) return EDTR_E_ALIGN;
    if (!aligned16(a.x) || !aligned16(a.wqkv) || !aligned16(a.wproj) || !aligned16(a.out) || !aligned16(a.c1) || !aligned16(a.c2b) ||
        !aligned16(a.bproj) || !aligned16(a.bias) || (a.labels && (reinterpret_cast<uintptr_t>(a.labels) & 3u)) || !aligned16(m.w1) ||
        !aligned16(m.w2) || !aligned16(m.c1) || !aligned16(m.c2b) || !aligned16(m.b2))
        return EDTR_E_ALIGN;
    const int64_t windows = (int64_t)a.B * (a.H / WS) * (a.W / WS);
    if (windows > 0x3fffffffLL) return EDTR_E_UNSUPPORTED;
    static EdtrLdsOnce attr_set[2];
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((windows + 1) / 2));
    if (a.dtype == EDTR_BF16) {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&swin_layer_kernel<BF16>), SL_LDS, attr_set[0])) return rc_;
        hipLaunchKernelGGL(swin_layer_kernel<BF16>, grid, dim3(SA_THREADS), SL_LDS, s, a, m, (int)windows);
    } else {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&swin_layer_kernel<F16>), SL_LDS, attr_set[1])) return rc_;
        hipLaunchKernelGGL(swin_layer_kernel<F16>, grid, dim3(SA_THREADS), SL_LDS, s, a, m, (int)windows);
    }
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_conv64(const edtr_conv64_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_conv64_params& p = *pp;
    if (!p.x || !p.w || !p.bias || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || (p.H & 15) || (p.W & 15)) return EDTR_E_SHAPE;
    if (p.act != EDTR_ACT_NONE && p.act != EDTR_ACT_LRELU) return EDTR_E_UNSUPPORTED;
    if (p.ldx < 64 || (p.ldx & 7)) return EDTR_E_ALIGN;
    if (p.out_nchw_f32) {
        if (p.n_valid < 1 || p.n_valid > 4) return EDTR_E_UNSUPPORTED;
    } else if (p.ldo < 64 || (p.ldo & 7)) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.out) || !aligned16(p.bias)) return EDTR_E_ALIGN;
    const int64_t patches = (int64_t)p.B * (p.H >> 4) * (p.W >> 4);
    if (patches > 0x7fffffffLL || (int64_t)p.B * p.H * p.W > 0x7fffffffLL) return EDTR_E_UNSUPPORTED;
    static EdtrLdsOnce attr_set[2];
    const int cus = edtr_cu_count();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(patches < cus ? patches : cus));          // persistent: one 154-KiB workgroup per CU
    if (p.dtype == EDTR_BF16) {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&conv64_kernel<BF16>), C64_LDS, attr_set[0])) return rc_;
        hipLaunchKernelGGL(conv64_kernel<BF16>, grid, dim3(C64_THREADS), C64_LDS, s, p, (int)patches);
    } else {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&conv64_kernel<F16>), C64_LDS, attr_set[1])) return rc_;
        hipLaunchKernelGGL(conv64_kernel<F16>, grid, dim3(C64_THREADS), C64_LDS, s, p, (int)patches);
    }
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_conv128_out(const edtr_conv128_out_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_conv128_out_params& p = *pp;
    if (!p.x || !p.w || !p.bias || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || (p.H & 15) || (p.W & 15)) return EDTR_E_SHAPE;
    if (p.n_valid < 1 || p.n_valid > 4) return EDTR_E_UNSUPPORTED;
    if (p.ldx < 128 || (p.ldx & 7)) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || !aligned16(p.w) || !aligned16(p.out) || !aligned16(p.bias) || (p.gn_table && !aligned16(p.gn_table))) return EDTR_E_ALIGN;
    const int64_t patches = (int64_t)p.B * (p.H >> 4) * (p.W >> 4);
    if (patches > 0x7fffffffLL || (int64_t)p.B * p.H * p.W > 0x7fffffffLL) return EDTR_E_UNSUPPORTED;
    static EdtrLdsOnce attr_set[2];
    const int cus = edtr_cu_count();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(patches < cus ? patches : cus));
    if (p.dtype == EDTR_BF16) {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&conv128_out_kernel<BF16>), CO_LDS, attr_set[0])) return rc_;
        hipLaunchKernelGGL(conv128_out_kernel<BF16>, grid, dim3(CO_THREADS), CO_LDS, s, p, (int)patches);
    } else {
        if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&conv128_out_kernel<F16>), CO_LDS, attr_set[1])) return rc_;
        hipLaunchKernelGGL(conv128_out_kernel<F16>, grid, dim3(CO_THREADS), CO_LDS, s, p, (int)patches);
    }
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
