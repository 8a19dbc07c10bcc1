// Implicit-GEMM convolution / linear / batched GEMM for gfx950 (see include/edtr_hip.h).
//
// Block tile (64*MI) x (64*NI) x 64, 256 threads = 4 waves in a 2x2 arrangement, each wave owning
// a (32*MI) x (32*NI) sub-tile as MI x NI accumulators of v_mfma_f32_32x32x16_{bf16,f16}.
// Operands are staged global -> registers -> LDS (the A gather needs per-chunk predication for the
// conv halo / concat / upsample, so the staging is register-based, not LDS-DMA), LDS is double
// buffered with ONE barrier per K-tile: the loads of tile t+2 are issued right after the registers
// holding tile t+1 have been written to LDS, so their latency hides under the MFMAs of tile t+1.
// LDS tiles are [rows][64 x 16-bit] with the XOR swizzle of common.h (conflict-free ds_read_b128).
// The epilogue stages the fp32 accumulators through LDS so that bias / time-embedding row vector /
// residual / activation are applied on 8-wide row vectors and stored with 16-byte writes.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

// Diagnostic build only (tools/exp/igemm_stamps.py compiles with -DEDTR_STAMPS): thread 0 of every workgroup stores
// s_memtime at a few phase boundaries into p.workspace (otherwise unused when splitk <= 1).  The product library
// is built without the flag and contains no stamp code.
#ifdef EDTR_STAMPS
#define STAMP_VALUE 0
#define EDTR_STAMP(i)                                                                                                  \
    do {                                                                                                               \
        if (threadIdx.x == 0 && p.workspace && p.splitk <= 1) {                                                        \
            uint64_t* sb__ = static_cast<uint64_t*>(p.workspace) +                                                     \
                             (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16;             \
            if ((i) == 5) sb__[5] = (uint64_t)__builtin_amdgcn_s_getreg(63492) | ((uint64_t)__builtin_amdgcn_s_getreg(63508) << 32); \
            else if ((i) == 6 || (i) == 7) sb__[i] = __builtin_amdgcn_s_memrealtime();                             \
            else if ((i) >= 8) sb__[i] = (uint64_t)(STAMP_VALUE);                                             \
            else sb__[i] = __builtin_amdgcn_s_memtime();                                                               \
        }                                                                                                              \
    } while (0)
#define EDTR_STAMP_T(i)                                                                                                \
    do {                                                                                                               \
        if (threadIdx.x == 0 && p.workspace && p.splitk <= 1) {                                                        \
            uint64_t* sb__ = static_cast<uint64_t*>(p.workspace) +                                                     \
                             (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16;             \
            sb__[i] = __builtin_amdgcn_s_memtime();                                                                    \
        }                                                                                                              \
    } while (0)
#else
#define EDTR_STAMP(i)
#define EDTR_STAMP_T(i)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int BK = 64;


// Tile order inside an XCD's contiguous range of workgroups (every kernel first maps blockIdx -> bid so that an XCD owns a
// contiguous range).  Normally the column tile runs fastest: an XCD walks whole rows of tiles and its L2 keeps the A rows.
// When the weights are the larger operand by far (the 3x3 convolutions at the 16x16 / 8x8 latent levels: M = 512..2048 rows
// against 1280 x 11520 weights) the row tile runs fastest instead, so that the workgroups sharing a weight slice sit on ONE XCD:
// with the other order every XCD streamed every weight from memory (PMC: 226 MB fetched per launch for 30 MB of weights).
__device__ __forceinline__ void tile_coords(const edtr_igemm_params& p, int bid, int nbm, int nbn, int& tm, int& tn) {
    const int64_t a_bytes = (int64_t)(p.upsample2x ? p.M >> 2 : p.M) * (p.C1 + p.C2) * 2, w_bytes = (int64_t)p.N * p.K * 2;   // (upsample: a quarter of the rows exist)
    // Neither operand fits an XCD's 4 MiB L2 and both tile counts are multiples of 8 (the 32x32-latent GEGLU projection,
    // M = 8192 x N = 5120 x K = 640: PMC 393 MB fetched per launch for 17 MB of operands, 5 TB/s — the launch ran at the HBM
    // rate): walk 8 x 8 super-blocks of tiles (one resident round of an XCD) so that 8 A row-panels + 8 W column-panels
    // (<= 3 MiB) serve 64 tiles.  Only when the model says it halves the traffic of both linear orders.
    if (((nbm | nbn) & 7) == 0) {
        // (32-bit divisions: a 64-bit one costs a couple of hundred instructions in every workgroup's prologue)
        const int64_t a_row = (int64_t)((p.M + nbm - 1) / nbm) * (p.C1 + p.C2) * 2, w_col = (int64_t)((p.N + nbn - 1) / nbn) * p.K * 2, cap = 3 << 20;
        const int64_t cost_rows = a_bytes + w_bytes * (w_bytes <= cap ? 8 : nbm);      // column tile fastest
        const int64_t cost_cols = w_bytes + a_bytes * (a_bytes <= cap ? 8 : nbn);      // row tile fastest
        const int64_t cost_2d = (int64_t)nbm * nbn * (a_row + w_col) / 8;
        if (8 * (a_row + w_col) <= cap && 2 * cost_2d <= (cost_rows < cost_cols ? cost_rows : cost_cols)) {
            const int blk = bid >> 6, r = bid & 63, nbn8 = nbn >> 3;
            const int bm = blk / nbn8, bn = blk - bm * nbn8;
            tm = bm * 8 + (r >> 3);
            tn = bn * 8 + (r & 7);
            return;
        }
    }
    const bool weight_heavy = w_bytes > 4 * a_bytes;
    if (weight_heavy) {
        tn = bid / nbm;
        tm = bid - tn * nbm;
    } else {
        tm = bid / nbn;
        tn = bid - tm * nbn;
    }
}

// Row-vector epilogue shared by the main kernel and the split-K reducer: 8 consecutive output columns of row m.
template <typename T>
__device__ __forceinline__ void finish_vector(const edtr_igemm_params& p, float (&f)[8], int m, int n, bool scale_bias,
                                              int64_t o_zoff) {
    if (scale_bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= p.alpha;
        if (p.bias_n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += p.bias_n[n + j];
        }
    }
    if (p.bias_m) {
        const float bm = p.bias_m[m];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += bm;
    }
    if (p.rowvec) {
        const float* rv = p.rowvec + (int64_t)(m / p.rows_per_image) * p.rowvec_ld + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += rv[j];
    }
    if (p.act == EDTR_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
    } else if (p.act == EDTR_ACT_GELU) {
        gelu_erf_lockstep<false>(f);
    } else if (p.act == EDTR_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], f[j] * p.act_slope);   // slope in [0, 1]
    }
    if (p.residual && p.residual_f32) {
        const float* rp = static_cast<const float*>(p.residual) + (int64_t)m * p.ldr + n;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] += r0[j]; f[j + 4] += r1[j]; }
    } else if (p.residual) {
        const U4 rv = ldg16(static_cast<const uint16_t*>(p.residual) + (int64_t)m * p.ldr + n);
        float rf[8];
        unpack8<T>(rv, rf);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += rf[j];
    }
    const int64_t oidx = o_zoff + (int64_t)m * p.ldc + n;
    if (p.out_f32) {
        float* o = static_cast<float*>(p.out) + oidx;
        f32x4 o0, o1;
        o0[0] = f[0]; o0[1] = f[1]; o0[2] = f[2]; o0[3] = f[3];
        o1[0] = f[4]; o1[1] = f[5]; o1[2] = f[6]; o1[3] = f[7];
        *reinterpret_cast<f32x4*>(o) = o0;
        *reinterpret_cast<f32x4*>(o + 4) = o1;
        if (p.out16) stg16(static_cast<uint16_t*>(p.out16) + (int64_t)m * p.ld16 + n, pack8<T>(f));      // 16-bit mirror of the fp32 stream
    } else {
        stg16(static_cast<uint16_t*>(p.out) + oidx, pack8<T>(f));
    }
}

// Split-K second stage: out = epilogue(sum over splits of the fp32 partial slabs [S][M][N]).
// The slabs are summed in split order (the result does not depend on how the loop is written), four splits per step with all eight
// 16-byte loads of a step in flight.  6.6 us per launch on average in the round-3 trace (3.6 % of the kernel time of a pass);
// against the rolled loop this form measured the same (hw_ab_tiles.py smallm: 31.8 vs 31.1 us for conv + reduce at 10 splits) — the
// launch is bound by the 13 - 31 MB of fp32 partials it reads, not by the loop.
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const edtr_igemm_params p) {
    const int nv = p.N >> 3;
    const int64_t total = (int64_t)p.M * nv;
    const float* ws = static_cast<const float*>(p.workspace);
    const int64_t slab = (int64_t)p.M * p.N;
    const bool small = total < (1LL << 31);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int m, n;
        if (small) { m = (int)((uint32_t)i / (uint32_t)nv); n = ((int)i - m * nv) * 8; }      // (a 64-bit division is a few hundred instructions)
        else { m = (int)(i / nv); n = (int)(i - (int64_t)m * nv) * 8; }
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.0f;
        const float* src = ws + (int64_t)m * p.N + n;
        int sidx = 0;
        for (; sidx + 4 <= p.splitk; sidx += 4) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* q = src + (sidx + k) * slab;
                a[k] = *reinterpret_cast<const f32x4*>(q);
                b[k] = *reinterpret_cast<const f32x4*>(q + 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[0] += a[k][0]; f[1] += a[k][1]; f[2] += a[k][2]; f[3] += a[k][3];
                f[4] += b[k][0]; f[5] += b[k][1]; f[6] += b[k][2]; f[7] += b[k][3];
            }
        }
        {
            f32x4 a[3], b[3];
            const int rem = p.splitk - sidx;       // 0..3: the loads of the tail are issued together as well
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < rem) {
                    const float* q = src + (sidx + k) * slab;
                    a[k] = *reinterpret_cast<const f32x4*>(q);
                    b[k] = *reinterpret_cast<const f32x4*>(q + 4);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < rem) {
                    f[0] += a[k][0]; f[1] += a[k][1]; f[2] += a[k][2]; f[3] += a[k][3];
                    f[4] += b[k][0]; f[5] += b[k][1]; f[6] += b[k][2]; f[7] += b[k][3];
                }
            }
        }
        finish_vector<T>(p, f, m, n, true, 0);
    }
}

// Split-K second stage WITH the fused GroupNorm statistics of the output (round 6): the split convolutions / linears of the 16 x 16 and
// 8 x 8 latent levels could not hand their consumer's GroupNorm its statistics (the main loops only write slabs), so an edtr_gn_stats
// launch re-read every such tensor — 100 of a batch-8 pass's 204 statistics launches, 2000 of 3250 on the 50-step workload, each one more
// dependent launch between a convolution and the normalisation behind it.  Here a block owns ONE statistics slot (SR = 64 or 128 rows,
// never across an image) x 32 columns: thread (row lane, column group) reduces the slabs of its 8-column vectors in split order like the
// plain reducer, finishes them, and the per-column sums of the FINISHED fp32 values (what the main-loop epilogues sum) meet in LDS.
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_stats_kernel(const edtr_igemm_params p, int SR) {
    __shared__ float red[64][4][16];                           // [row lane][column group][8 sums | 8 sums of squares]
    const int tid = threadIdx.x, g = tid & 3, rl = tid >> 2;    // 4 column groups of 8 columns x 64 row lanes
    const int n = blockIdx.x * 32 + g * 8, m_base = blockIdx.y * SR;
    const float* ws = static_cast<const float*>(p.workspace);
    const int64_t slab = (int64_t)p.M * p.N;
    float cs[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] = 0.0f; cq[j] = 0.0f; }
    if (n < p.N) {
        for (int m = m_base + rl; m < m_base + SR; m += 64) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.0f;
            const float* src = ws + (int64_t)m * p.N + n;
            for (int sidx = 0; sidx < p.splitk; ++sidx) {      // split order: the result does not depend on the launch geometry
                const f32x4 a = *reinterpret_cast<const f32x4*>(src + sidx * slab), b = *reinterpret_cast<const f32x4*>(src + sidx * slab + 4);
                f[0] += a[0]; f[1] += a[1]; f[2] += a[2]; f[3] += a[3];
                f[4] += b[0]; f[5] += b[1]; f[6] += b[2]; f[7] += b[3];
            }
            finish_vector<T>(p, f, m, n, true, 0);              // (f holds the finished values afterwards)
#pragma unroll
            for (int j = 0; j < 8; ++j) { cs[j] += f[j]; cq[j] += f[j] * f[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[rl][g][j] = cs[j]; red[rl][g][8 + j] = cq[j]; }
    __syncthreads();
    if (tid < 32 && blockIdx.x * 32 + tid < p.N) {              // one thread per column: the 64 row lanes in lane order
        const int gg = tid >> 3, j = tid & 7;
        float a = 0.0f, q = 0.0f;
        for (int r = 0; r < 64; ++r) { a += red[r][gg][j]; q += red[r][gg][8 + j]; }
        float* dst = p.gn_partial + ((int64_t)blockIdx.y * gn_ld_of(p) + blockIdx.x * 32 + tid) * 2;
        dst[0] = a;
        dst[1] = q;
    }
}

// the second launch of a split-K edtr_igemm: the plain reducer, or the one that also writes the output's GroupNorm statistics
template <typename T>
static int launch_splitk_reducer(const edtr_igemm_params& p, hipStream_t stream) {
    if (p.gn_partial) {
        const int SR = p.gn_slot_rows > 0 ? p.gn_slot_rows : 128;
        hipLaunchKernelGGL(splitk_reduce_stats_kernel<T>, dim3((unsigned)((p.N + 31) / 32), (unsigned)(p.M / SR)), dim3(256), 0, stream, p, SR);
    } else {
        const int64_t nvec = (int64_t)p.M * (p.N >> 3);
        int64_t blocks = (nvec + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    }
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

// Second half of the tile epilogue: the staged fp32 tile [BM][BNO] -> 8-column row vectors -> bias / time-embedding
// row / SiLU / residual -> 16-byte stores.  A thread owns ONE 8-column group for all of its ITER rows (256 threads,
// BNO/8 groups), so the per-column operands (bias, and the time-embedding row when the tile lies inside one image) are
// loaded once, and the ITER residual vectors are requested up front — before the barrier that publishes the staged
// tile — so that their L2/HBM latency overlaps the staging instead of serialising ITER dependent round trips
// (measured with tools/exp/igemm_stamps.py: 9-15k cycles per tile before).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// Two workgroups share a CU in the 2-per-CU kernels, and the hardware starts blocks i and i + 256 on the same CU within the same
// microsecond (tools/exp/igemm_stamps.py, EDTR_STAMP_PAIRS): they then run in LOCKSTEP — both in their K loop (VALU and the store
// path idle), both in their epilogue (MFMA idle; the 512 simultaneous store bursts are bound by HBM write bandwidth: 6.5k of a
// short-K tile's 20k cycles).  Holding the second block of each CU back by half a workgroup life at the START of the launch
// interleaves the two for the rest of it.  stagger = cycles to wait (0 = off); only blocks 256..511 of the dispatch order wait.
__device__ __forceinline__ void stagger_second_slot(const edtr_igemm_params& p) {
    if (p.stagger <= 0) return;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin < 256u || lin >= 512u) return;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < (int64_t)p.stagger) __builtin_amdgcn_s_sleep(32);
}

// ---- LayerNorm folded into the GEMMs around it (edtr_hip.h: row_stats / ln_stats) ------------------------------------------
// One LDS array of per-row (mean, rstd) for the rows of the tile being finished (all igemm kernels have <= 256 rows per pass).
__device__ __forceinline__ float2* ln_rows_lds() {
    __shared__ float2 rows[256];
    return rows;
}
// consumer side: fold the row's ln_slots (sum, sum of squares) slots into (mean, rstd); published by the caller's next barrier
template <int BM, int THREADS>
__device__ __forceinline__ void ln_rows_fill(const edtr_igemm_params& p, int m0) {
    float2* dst = ln_rows_lds();
    const float inv = 1.0f / (float)p.ln_C;
    for (int r = threadIdx.x; r < BM; r += THREADS) {
        const int m = m0 + r;
        float s = 0.0f, q = 0.0f;
        if (m < p.M) {
            const f32x2* src = reinterpret_cast<const f32x2*>(p.ln_stats) + (int64_t)m * p.ln_slots;
            for (int k = 0; k < p.ln_slots; ++k) { const f32x2 v = src[k]; s += v[0]; q += v[1]; }
        }
        const float mean = s * inv;
        const float var = fmaxf(q * inv - mean * mean, 0.0f);
        dst[r] = make_float2(mean, __builtin_amdgcn_rsqf(var + p.ln_eps));
    }
}

// Transposed store of a staged tile whose columns are the V part of a fused [Q; K; V] projection: V^T[image][c][token], 8
// consecutive tokens (rows of the tile) of one column per 16-byte store.  Lane (slot, cl): 16 adjacent columns per slot — the
// LDS reads of a slot walk 16 adjacent banks (slots collide 4-way: 8 reads per lane, negligible), the 16-byte stores of the
// lanes that share a column are adjacent in memory.
template <typename T, int BM, int BNO, int THREADS, int PITCH, int FOLD = 3>
__device__ __forceinline__ void vt_store(const edtr_igemm_params& p, const float* stage, int m0, int no0) {
    static_assert(BM % 8 == 0 && BNO % 16 == 0 && THREADS % 16 == 0, "vt_store tiling");
    constexpr int TG = BM / 8, NCB = BNO / 16, SLOTS = THREADS / 16, PAIRS = TG * NCB;
    const int tid = threadIdx.x, cl = tid & 15, slot = tid >> 4;
    const int vt_rows = p.N - p.vt_col0;
    uint16_t* vt = static_cast<uint16_t*>(p.vt_out);
    for (int u = slot; u < PAIRS; u += SLOTS) {
        const int tg = u % TG, cb = u / TG;
        const int cloc = cb * 16 + cl, n = no0 + cloc, m = m0 + tg * 8;
        if (m >= p.M || n >= p.N) continue;
        float b = p.bias_n ? p.bias_n[n] : 0.0f;
        float f[8];
        if ((FOLD & 1) && p.ln_stats) {        // folded LayerNorm of the tokens (rows): rstd (alpha acc - mean alpha c1) + alpha c2 + bias
            const float c1a = p.vt_alpha * p.ln_c1[n];
            b += p.vt_alpha * p.ln_c2[n];
            const float2* lr = ln_rows_lds() + tg * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = lr[i].y * (stage[(tg * 8 + i) * PITCH + cloc] * p.vt_alpha - lr[i].x * c1a) + b;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(stage[(tg * 8 + i) * PITCH + cloc], p.vt_alpha, b);
        }
        const int img = m / p.rows_per_image, tok = m - img * p.rows_per_image;
        stg16(vt + ((int64_t)img * vt_rows + (n - p.vt_col0)) * p.vt_ld + tok, pack8<T>(f));
    }
}

// FOLD: bit 0 = the folded LayerNorm's consumer side (row scalars in 2 KiB of static LDS), bit 1 = its producer side (row statistics
// of the output; no extra LDS).  FOLD = 0 compiles the folded-LayerNorm paths out (the halo kernels: their register budget is full and no LayerNorm sits next
// to a 3x3 convolution)
// PF: how the staged vectors are read — 2 = all of this thread's vectors up front (needs 8 ITER free registers: the kernels whose
// accumulators are all dead by now), 1 = one row iteration ahead (16 registers), 0 = inside the iteration (the 512-thread kernels,
// whose register file is full)
template <typename T, int BM, int BNO, bool GEGLU, int THREADS = 256, int PATCH16 = 0, int PITCH = BNO, int FOLD = 3, int PF = 0,
          typename Hook = NoHook>
__device__ __forceinline__ void rows_phase(const edtr_igemm_params& p, const float* stage, int m0, int no0, int n_out,
                                           int64_t o_zoff, bool gn_acc, float (&gs)[8], float (&gq)[8], Hook before_publish = Hook()) {
    // VPR column groups; RPI rows per iteration (threads beyond RPI * VPR idle when VPR does not divide the block)
    constexpr int VPR = BNO / 8, RPI = THREADS / VPR, ITER = (BM + RPI - 1) / RPI;
    constexpr bool EXACT = (THREADS % VPR == 0) && (BM % RPI == 0);
    const int tid = (int)threadIdx.x;
    // staged row -> output row.  PATCH16 = 1: the rows are the pixels of a 16-pixel-wide patch (row ml = 16 y + x), m0 = its first pixel
    // 3: one output PHASE of the sub-pixel upsample convolution — staged row ml = 16 y + x is SOURCE pixel (y, x) of a 16 x 16 source
    // block, its output pixel lies at (2 y, 2 x) from m0 (the phase's first output pixel)
    auto row_m = [&](int ml) {
        if constexpr (PATCH16 == 0) return m0 + ml;
        else if constexpr (PATCH16 == 3) return m0 + 2 * ((ml >> 4) * p.OW + (ml & 15));
        else return m0 + (ml >> 4) * p.OW + (ml & 15);
    };
    const int n8 = tid % VPR, r0 = tid / VPR;
    const int n = no0 + n8 * 8;
    const bool n_ok = n < n_out && (EXACT || r0 < RPI);

    if (p.splitk > 1) {                       // fp32 partial slab, finished by splitk_reduce_kernel
        before_publish();
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int ml = r0 + RPI * it, m = row_m(ml);
            if (m < p.M && n_ok && (EXACT || ml < BM)) {
                f32x4 s0, s1;
                s0 = *reinterpret_cast<const f32x4*>(stage + ml * PITCH + n8 * 8);
                s1 = *reinterpret_cast<const f32x4*>(stage + ml * PITCH + n8 * 8 + 4);
                float* o = static_cast<float*>(p.workspace) + ((int64_t)blockIdx.y * p.M + m) * p.N + n;
                *reinterpret_cast<f32x4*>(o) = s0;
                *reinterpret_cast<f32x4*>(o + 4) = s1;
            }
        }
        return;
    }

    const bool ln = (FOLD & 1) && !GEGLU && !PATCH16 && p.ln_stats != nullptr;      // (GEGLU: the caller applied it before the gate product)
    if constexpr (!GEGLU && !PATCH16) {
        if (ln) ln_rows_fill<BM, THREADS>(p, m0);        // visible after the barrier that publishes the staged tile
        if (p.vt_out != nullptr && no0 >= p.vt_col0) {    // a V tile of the fused [Q; K; V] projection: transposed store
            before_publish();
            __syncthreads();
            vt_store<T, BM, BNO, THREADS, PITCH, FOLD>(p, stage, m0, no0);
            return;
        }
    }

    U4 res[ITER];
    const bool res16 = p.residual && !p.residual_f32;     // an fp32 residual (fp32 activation stream) is read inside the row loop
    if (res16) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int mlr = r0 + RPI * it;
            const int m = row_m(mlr);
            res[it] = (m < p.M && n_ok && (EXACT || r0 + RPI * it < BM)) ? ldg16(static_cast<const uint16_t*>(p.residual) + (int64_t)m * p.ldr + n) : zero16();
        }
    }
    // per-column addends: bias, and the time-embedding row when it is uniform over the tile.  The two are added SEPARATELY, in the
    // order the per-row form uses (fma(acc, alpha, bias) + row): whether a tile lies inside one image depends on the batch size,
    // and an image's result must not (EDTR_AMD_BATCH_INVARIANT; folding the row into the bias rounds differently).
    float cb[8], rvu[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cb[j] = 0.0f; rvu[j] = 0.0f; }
    if (!GEGLU && p.bias_n && n_ok) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias_n + n), b1 = *reinterpret_cast<const f32x4*>(p.bias_n + n + 4);
        cb[0] = b0[0]; cb[1] = b0[1]; cb[2] = b0[2]; cb[3] = b0[3]; cb[4] = b1[0]; cb[5] = b1[1]; cb[6] = b1[2]; cb[7] = b1[3];
    }
    bool rv_rows = false;                     // time-embedding row differs between this tile's rows
    if (p.rowvec) {
        const int m_last = PATCH16 == 3 ? m0 + 30 * p.OW + 30 : PATCH16 ? m0 + 15 * p.OW + 15 : min(m0 + BM, p.M) - 1;      // (PATCH16: the patch's last pixel or a later one of the same image)
        const int img0 = m0 / p.rows_per_image;
        rv_rows = (m_last / p.rows_per_image) != img0;
        if (!rv_rows && n_ok) {
            const float* rv = p.rowvec + (int64_t)img0 * p.rowvec_ld + n;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(rv), b1 = *reinterpret_cast<const f32x4*>(rv + 4);
            if constexpr (PATCH16 != 0) {      // the halo kernels' tiles never leave an image (and their register file is full): folded
                cb[0] += b0[0]; cb[1] += b0[1]; cb[2] += b0[2]; cb[3] += b0[3]; cb[4] += b1[0]; cb[5] += b1[1]; cb[6] += b1[2]; cb[7] += b1[3];
            } else {
                rvu[0] = b0[0]; rvu[1] = b0[1]; rvu[2] = b0[2]; rvu[3] = b0[3]; rvu[4] = b1[0]; rvu[5] = b1[1]; rvu[6] = b1[2]; rvu[7] = b1[3];
            }
        }
    }
    const float alpha = GEGLU ? 1.0f : p.alpha;
    float c1a[8];                             // folded LayerNorm: alpha c1[n] (and alpha c2[n] joins the per-column addend)
    if (ln && n_ok) {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(p.ln_c1 + n), u1 = *reinterpret_cast<const f32x4*>(p.ln_c1 + n + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.ln_c2 + n), v1 = *reinterpret_cast<const f32x4*>(p.ln_c2 + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c1a[j] = alpha * u0[j]; c1a[j + 4] = alpha * u1[j];
            cb[j] += alpha * v0[j]; cb[j + 4] += alpha * v1[j];
        }
    }
    const bool stats_out = (FOLD & 2) && !GEGLU && !PATCH16 && p.row_stats != nullptr;
    float rsum[ITER], rsq[ITER];              // producer side of the fold: this thread's 8-column share of each row's sum / sum of squares
#pragma unroll
    for (int it = 0; it < ITER; ++it) { rsum[it] = 0.0f; rsq[it] = 0.0f; }
    const bool silu = p.act == EDTR_ACT_SILU, gelu = p.act == EDTR_ACT_GELU, lrelu = p.act == EDTR_ACT_LRELU;
    before_publish();                         // the caller's last staging step runs under the flight of the loads above
    __syncthreads();                          // staged tile (and the folded LayerNorm's row scalars) visible
    EDTR_STAMP_T(13);

    // All of this thread's staged vectors are requested up front, outside the row loop's control flow: inside it the compiler
    // cannot move an iteration's ds_reads above the previous iteration's (predicated) stores, and the loop then pays one LDS
    // round trip per iteration — 6.5k cycles of a 128 x 128 tile's 8.6k-cycle epilogue in the in-kernel stamps (round 3).  The
    // accumulators are dead by now, so the 8 ITER registers are free.  (Rows beyond BM in the non-exact split are clamped: the read
    // must stay inside the staged tile, its value is never used.)
    constexpr int NPF = PF == 2 ? ITER : (PF == 1 ? 2 : 1);
    f32x4 sv0[NPF], sv1[NPF];
    auto stage_read = [&](int it, int slot) {
        const int mlr = r0 + RPI * it, mlc = EXACT ? mlr : (mlr < BM ? mlr : BM - 1);
        const int n8c = EXACT ? n8 : (n8 < VPR ? n8 : 0);
        sv0[slot] = *reinterpret_cast<const f32x4*>(stage + mlc * PITCH + n8c * 8);
        sv1[slot] = *reinterpret_cast<const f32x4*>(stage + mlc * PITCH + n8c * 8 + 4);
    };
    if constexpr (PF == 2) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) stage_read(it, it);
    } else if constexpr (PF == 1) {
        stage_read(0, 0);
    }

    bool did_fast = false;
    // The common case — 16-bit output, bias (+ a uniform time-embedding row) in cb, optional 16-bit residual, optional GroupNorm
    // partials, nothing else — gets its own loop: the general loop below tests ~10 launch-uniform options INSIDE every iteration
    // (the compiler does not unswitch them), and with one wave per SIMD each of those scalar branches and the 64-bit address
    // arithmetic they fence costs its full latency: 800 cycles per iteration, 6.4k of a 128 x 128 tile's 8.4k-cycle epilogue
    // (tools/exp/igemm_stamps.py, round 3).  Here the only control flow is the store's predicate.
    {
        const bool fast = !p.bias_m && !rv_rows && !silu && !gelu && !lrelu && !(p.debug_flags & 1);
        // OUT32: fp32 output (the fp32 activation stream of the mixed / high modes); RES32: fp32 residual, requested up front
        // like the staged vectors where the registers allow (PF == 2), inside the iteration otherwise
        // LNF / STF: the two sides of the folded LayerNorm (consumer: row scalars from LDS; producer: per-row sums of the stored
        // values, folded after the loop) — 16-bit output, no fp32 residual: the SwinIR layers and the UNet blocks under EDTR_LN_FOLD
        // MIR: the fp32 stream's 16-bit MIRROR (p.out16: the one-part operand its 16-bit consumers read without a cast launch)
        auto fast_loop = [&](auto out32_c, auto res32_c, auto lnf_c, auto stf_c, auto mir_c) {
            constexpr bool OUT32 = decltype(out32_c)::value, RES32 = decltype(res32_c)::value;
            constexpr bool LNF = decltype(lnf_c)::value, STF = decltype(stf_c)::value, MIR = decltype(mir_c)::value && OUT32;
            constexpr int NRF = (RES32 && PF == 2) ? ITER : 1;
            f32x4 rf0[NRF], rf1[NRF];
            auto res_read = [&](int it, int slot) {
                const int ml = r0 + RPI * it, m = row_m(ml);
                if (n_ok && m < p.M && (EXACT || ml < BM)) {
                    const float* rp = static_cast<const float*>(p.residual) + (int64_t)m * p.ldr + n;
                    rf0[slot] = *reinterpret_cast<const f32x4*>(rp);
                    rf1[slot] = *reinterpret_cast<const f32x4*>(rp + 4);
                } else {
                    rf0[slot] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    rf1[slot] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            };
            if constexpr (RES32 && PF == 2) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) res_read(it, it);
            }
            char* const outp = static_cast<char*>(p.out) + (o_zoff + n) * (OUT32 ? 4 : 2);
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int ml = r0 + RPI * it, m = row_m(ml);
                if constexpr (PF == 1) {
                    if (it + 1 < ITER) stage_read(it + 1, (it + 1) & 1);
                }
                if constexpr (RES32 && PF != 2) res_read(it, 0);
                const bool ok = n_ok && m < p.M && (EXACT || ml < BM);
                f32x4 s0, s1;
                if constexpr (PF == 0) {
                    const int mlc = EXACT ? ml : (ml < BM ? ml : BM - 1), n8c = EXACT ? n8 : (n8 < VPR ? n8 : 0);
                    s0 = *reinterpret_cast<const f32x4*>(stage + mlc * PITCH + n8c * 8);
                    s1 = *reinterpret_cast<const f32x4*>(stage + mlc * PITCH + n8c * 8 + 4);
                } else {
                    s0 = sv0[PF == 2 ? it : (it & 1)];
                    s1 = sv1[PF == 2 ? it : (it & 1)];
                }
                float f[8];
                f[0] = s0[0]; f[1] = s0[1]; f[2] = s0[2]; f[3] = s0[3]; f[4] = s1[0]; f[5] = s1[1]; f[6] = s1[2]; f[7] = s1[3];
                if constexpr (LNF) {
                    const float2 mr = ln_rows_lds()[EXACT ? ml : (ml < BM ? ml : BM - 1)];
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = mr.y * (f[j] * alpha - mr.x * c1a[j]) + cb[j] + rvu[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = PATCH16 != 0 ? __builtin_fmaf(f[j], alpha, cb[j]) : __builtin_fmaf(f[j], alpha, cb[j]) + rvu[j];
                }
                if constexpr (RES32) {
                    const f32x4 q0 = rf0[PF == 2 ? it : 0], q1 = rf1[PF == 2 ? it : 0];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { f[j] += q0[j]; f[j + 4] += q1[j]; }
                } else if (res16) {
                    float rf[8];
                    unpack8<T>(res[it], rf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] += rf[j];
                }
                if (ok) {
                    if constexpr (OUT32) {
                        float* o = reinterpret_cast<float*>(outp) + (int64_t)m * p.ldc;
                        f32x4 o0, o1;
                        o0[0] = f[0]; o0[1] = f[1]; o0[2] = f[2]; o0[3] = f[3];
                        o1[0] = f[4]; o1[1] = f[5]; o1[2] = f[6]; o1[3] = f[7];
                        *reinterpret_cast<f32x4*>(o) = o0;
                        *reinterpret_cast<f32x4*>(o + 4) = o1;
                        if constexpr (MIR) stg16(static_cast<uint16_t*>(p.out16) + (int64_t)m * p.ld16 + n, pack8<T>(f));
                    } else {
                        stg16(reinterpret_cast<uint16_t*>(outp) + (int64_t)m * p.ldc, pack8<T>(f));
                    }
                    if (gn_acc) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { gs[j] += f[j]; gq[j] += f[j] * f[j]; }
                    }
                    if constexpr (STF) {
                        // statistics of the STORED 16-bit values (what the consuming GEMM multiplies), not of the fp32 ones
                        // (the mismatch is of the order of the rounding itself on rows whose mean dwarfs their spread)
                        float fr[8];
                        unpack8<T>(pack8<T>(f), fr);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { rsum[it] += fr[j]; rsq[it] += fr[j] * fr[j]; }
                    }
                }
            }
        };
        if (fast) {
            using std::true_type;
            using std::false_type;
            const bool res32 = p.residual && p.residual_f32;
            if (!ln && !stats_out) {
                const bool mir = p.out16 != nullptr;
                if (!p.out_f32 && !res32) fast_loop(false_type{}, false_type{}, false_type{}, false_type{}, false_type{});
                else if (p.out_f32 && res32 && mir) fast_loop(true_type{}, true_type{}, false_type{}, false_type{}, true_type{});
                else if (p.out_f32 && res32) fast_loop(true_type{}, true_type{}, false_type{}, false_type{}, false_type{});
                else if (p.out_f32 && mir) fast_loop(true_type{}, false_type{}, false_type{}, false_type{}, true_type{});
                else if (p.out_f32) fast_loop(true_type{}, false_type{}, false_type{}, false_type{}, false_type{});
                else fast_loop(false_type{}, true_type{}, false_type{}, false_type{}, false_type{});
                did_fast = true;
            } else if (!p.out_f32 && !res32) {
                if constexpr (FOLD != 0 && !GEGLU && PATCH16 == 0) {
                    if (ln && stats_out) {
                        if constexpr (FOLD == 3) { fast_loop(false_type{}, false_type{}, true_type{}, true_type{}, false_type{}); did_fast = true; }
                    } else if (ln) {
                        if constexpr ((FOLD & 1) != 0) { fast_loop(false_type{}, false_type{}, true_type{}, false_type{}, false_type{}); did_fast = true; }
                    } else {
                        if constexpr ((FOLD & 2) != 0) { fast_loop(false_type{}, false_type{}, false_type{}, true_type{}, false_type{}); did_fast = true; }
                    }
                }
            }
            if (did_fast && !stats_out) {
                EDTR_STAMP_T(14);
                return;
            }
        }
    }

    if (!did_fast) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        // PATCH16: the tile's rows are the pixels of a 16 x 16 output patch (row ml = 16 * y + x), m0 = its first pixel
        const int ml = r0 + RPI * it, m = row_m(ml);
        if constexpr (PF == 1) {
            if (it + 1 < ITER) stage_read(it + 1, (it + 1) & 1);       // the next iteration's vectors fly under this one's arithmetic
        }
        if (m < p.M && n_ok && (EXACT || ml < BM)) {
            f32x4 s0, s1;
            if constexpr (PF == 0) {
                s0 = *reinterpret_cast<const f32x4*>(stage + ml * PITCH + n8 * 8);
                s1 = *reinterpret_cast<const f32x4*>(stage + ml * PITCH + n8 * 8 + 4);
            } else {
                s0 = sv0[PF == 2 ? it : (it & 1)];
                s1 = sv1[PF == 2 ? it : (it & 1)];
            }
            float f[8];
            f[0] = s0[0]; f[1] = s0[1]; f[2] = s0[2]; f[3] = s0[3]; f[4] = s1[0]; f[5] = s1[1]; f[6] = s1[2]; f[7] = s1[3];
            if (ln) {
                const float2 mr = ln_rows_lds()[ml];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = mr.y * (f[j] * alpha - mr.x * c1a[j]) + cb[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = __builtin_fmaf(f[j], alpha, cb[j]);
            }
            if (p.bias_m) {
                const float bm = p.bias_m[m];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += bm;
            }
            if (rv_rows) {
                const float* rv = p.rowvec + (int64_t)(m / p.rows_per_image) * p.rowvec_ld + n;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += rv[j];
            } else if constexpr (PATCH16 == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += rvu[j];
            }
            if (silu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
            } else if (gelu) {
                gelu_erf_lockstep<false>(f);
            } else if (lrelu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], f[j] * p.act_slope);
            }
            if (res16) {
                float rf[8];
                unpack8<T>(res[it], rf);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += rf[j];
            } else if (p.residual) {
                const float* rp = static_cast<const float*>(p.residual) + (int64_t)m * p.ldr + n;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] += r0[j]; f[j + 4] += r1[j]; }
            }
            const int64_t oidx = o_zoff + (int64_t)m * p.ldc + n;
            if (p.out_f32) {
                float* o = static_cast<float*>(p.out) + oidx;
                f32x4 o0, o1;
                o0[0] = f[0]; o0[1] = f[1]; o0[2] = f[2]; o0[3] = f[3];
                o1[0] = f[4]; o1[1] = f[5]; o1[2] = f[6]; o1[3] = f[7];
                *reinterpret_cast<f32x4*>(o) = o0;
                *reinterpret_cast<f32x4*>(o + 4) = o1;
                if (p.out16) stg16(static_cast<uint16_t*>(p.out16) + (int64_t)m * p.ld16 + n, pack8<T>(f));
            } else {
                stg16(static_cast<uint16_t*>(p.out) + oidx, pack8<T>(f));
            }
            if (gn_acc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[j] += f[j]; gq[j] += f[j] * f[j]; }
            }
            if (stats_out) {
                float fr[8];
                if (p.out_f32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) fr[j] = f[j];
                } else {
                    unpack8<T>(pack8<T>(f), fr);       // the stored 16-bit values
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { rsum[it] += fr[j]; rsq[it] += fr[j] * fr[j]; }
            }
        }
    }
    }
    EDTR_STAMP_T(14);
    if (stats_out) {
        // per-row statistics of the tensor just written (the input of a LayerNorm that the NEXT GEMM folds in): the VPR threads
        // that share a row meet in LDS (the staged tile is dead), one thread per row writes the tile's slot
        __syncthreads();
        float2* red = reinterpret_cast<float2*>(const_cast<float*>(stage));
        if (EXACT || r0 < RPI) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int ml = r0 + RPI * it;
                if (EXACT || ml < BM) red[ml * VPR + n8] = make_float2(rsum[it], rsq[it]);
            }
        }
        __syncthreads();
        const int nslots = p.N >> 5, slot0 = no0 >> 5;
        for (int r = tid; r < BM; r += THREADS) {
            const int m = m0 + r;
            if (m < p.M) {
                float a = 0.0f, q = 0.0f;
#pragma unroll
                for (int k = 0; k < VPR; ++k) { const float2 v = red[r * VPR + k]; a += v.x; q += v.y; }
                float* dst = p.row_stats + ((int64_t)m * nslots + slot0) * 2;
                dst[0] = a;
                dst[1] = q;
                for (int k = 1; k < BNO / 32 && slot0 + k < nslots; ++k) { dst[2 * k] = 0.0f; dst[2 * k + 1] = 0.0f; }
            }
        }
    }
}

// Tile epilogue shared by both main-loop variants: accumulators -> LDS (fp32) -> row vectors of 8 columns.
template <typename T, int MI, int NI>
__device__ __forceinline__ void tile_epilogue(const edtr_igemm_params& p, f32x16 (&acc)[MI][NI], char* smem, int m0, int n0,
                                              int64_t o_zoff) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool geglu = (p.act == EDTR_ACT_GEGLU);
    const int BNO = geglu ? BN / 2 : BN;      // output columns of this block
    float* stage = reinterpret_cast<float*>(smem);
    if (geglu) {
        if constexpr (NI == 2) {
            const int nv = n0 + wn * 64 + l31;  // packed column of the value half; gate = +32
            float bv = p.bias_n ? p.bias_n[nv] : 0.0f;
            float bg = p.bias_n ? p.bias_n[nv + 32] : 0.0f;
            const bool ln = p.ln_stats != nullptr;
            float c1v = 0.0f, c1g = 0.0f;
            if (ln) {                            // folded LayerNorm: row scalars through LDS, the two per-column vectors in registers
                ln_rows_fill<BM, kThreads>(p, m0);
                c1v = p.alpha * p.ln_c1[nv]; c1g = p.alpha * p.ln_c1[nv + 32];
                bv += p.alpha * p.ln_c2[nv]; bg += p.alpha * p.ln_c2[nv + 32];
                __syncthreads();
            }
            // two straight-line loops, not one loop with the launch-uniform `ln` test inside: with the branch in every iteration each
            // GELU (a dependent chain of ~16 instructions, two of them transcendental) sits in its own basic block and runs at its
            // latency — 250 cycles per element, 8.2k of the tile's 20k cycles in the in-kernel stamps (round 3)
            float* const st = stage + (wm * 32 * MI + 4 * lh) * BNO + wn * 32 + l31;
            const float alpha = p.alpha;
            // (round 4) ... and eight gates at a time through gelu_erf_lockstep: even in straight-line code hipcc runs each value's
            // chain to its end before the next (EDTR_IGEMM_GEGLU_SERIAL=1 — debug_flags bit 2 — is the A/B)
            if (p.debug_flags & 4) {
                if (ln) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int mo = mi * 32 + (r & 3) + 8 * (r >> 2);
                            const float2 mr = ln_rows_lds()[wm * 32 * MI + 4 * lh + mo];
                            const float val = mr.y * (acc[mi][0][r] * alpha - mr.x * c1v) + bv;
                            const float gate = mr.y * (acc[mi][1][r] * alpha - mr.x * c1g) + bg;
                            st[mo * BNO] = val * gelu_erf_f(gate);
                        }
                } else {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int mo = mi * 32 + (r & 3) + 8 * (r >> 2);
                            const float val = acc[mi][0][r] * alpha + bv;
                            const float gate = acc[mi][1][r] * alpha + bg;
                            st[mo * BNO] = val * gelu_erf_f(gate);
                        }
                }
            } else if (ln) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg) {
                        float val[8], gate[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * rg + e, mo = mi * 32 + (r & 3) + 8 * (r >> 2);
                            const float2 mr = ln_rows_lds()[wm * 32 * MI + 4 * lh + mo];
                            val[e] = mr.y * (acc[mi][0][r] * alpha - mr.x * c1v) + bv;
                            gate[e] = mr.y * (acc[mi][1][r] * alpha - mr.x * c1g) + bg;
                        }
                        gelu_erf_lockstep<false>(gate);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * rg + e, mo = mi * 32 + (r & 3) + 8 * (r >> 2);
                            st[mo * BNO] = val[e] * gate[e];
                        }
                    }
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg) {
                        float val[8], gate[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * rg + e;
                            val[e] = acc[mi][0][r] * alpha + bv;
                            gate[e] = acc[mi][1][r] * alpha + bg;
                        }
                        gelu_erf_lockstep<false>(gate);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * rg + e, mo = mi * 32 + (r & 3) + 8 * (r >> 2);
                            st[mo * BNO] = val[e] * gate[e];
                        }
                    }
            }
        }
    } else {
        // (staging these inside rows_phase, as its pre-publish hook behind the operand loads — as the 128 x 160 tile does — was
        //  tried here: the 128 x 128 kernels then spill 24 - 28 registers and the row loop triples)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    stage[ml * BNO + wn * 32 * NI + ni * 32 + l31] = acc[mi][ni][r];
                }
    }
    EDTR_STAMP_T(12);
    const int n_out = geglu ? p.N / 2 : p.N;
    const int no0 = geglu ? n0 / 2 : n0;
    // fused GroupNorm statistics of the tensor being written (per-column sum / sum of squares over this tile's rows):
    // each thread keeps the same 8 columns for all of its row vectors (16 vectors per 128-column row, 256 threads)
    const bool gn_acc = p.gn_partial != nullptr && !geglu && p.splitk <= 1 && NI == 2 && MI == 2;
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.0f; gq[j] = 0.0f; }
    if (geglu) rows_phase<T, BM, BN / 2, true, 256, false, BN / 2, 3, 2>(p, stage, m0, no0, n_out, o_zoff, false, gs, gq);
    else rows_phase<T, BM, BN, false, 256, false, BN, 3, 2>(p, stage, m0, no0, n_out, o_zoff, gn_acc, gs, gq);
    if (gn_acc) {
        // lanes l, l+16, l+32, l+48 of a wave own the same columns: fold them, then fold the 4 waves through LDS
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gs[j] += __shfl_xor(gs[j], 16, 64); gs[j] += __shfl_xor(gs[j], 32, 64);
            gq[j] += __shfl_xor(gq[j], 16, 64); gq[j] += __shfl_xor(gq[j], 32, 64);
        }
        __syncthreads();                       // every thread is done reading the staged tile
        if (lane < 16) {
            float* dst = stage + (wave * 128 + lane * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[j]; dst[2 * j + 1] = gq[j]; }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float a = 0.0f, q = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += stage[(w * 128 + tid) * 2]; q += stage[(w * 128 + tid) * 2 + 1]; }
            float* dst = p.gn_partial + ((int64_t)(m0 / BM) * gn_ld_of(p) + n0 + tid) * 2;
            dst[0] = a;
            dst[1] = q;
        }
    }
}

template <typename T, int MI, int NI, bool SPATIAL>
__global__ void __launch_bounds__(kThreads, 2) igemm_kernel(const edtr_igemm_params p) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    constexpr int RA = BM / 32, RW = BN / 32;  // 16-byte chunks per thread per K-tile
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- block -> tile, XCD-aware (blocks b and b+8 share an XCD/L2: give each XCD a contiguous run)
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nblk = nbm * nbn, q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn;
    tile_coords(p, bid, nbm, nbn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- batch (grid.z) offsets
    const int z = blockIdx.z, zo = z / p.zdiv, zi = z - zo * p.zdiv;
    const int64_t a_zoff = zo * p.a_zs_outer + zi * p.a_zs_inner;
    const int64_t w_zoff = zo * p.w_zs_outer + zi * p.w_zs_inner;
    const int64_t o_zoff = zo * p.o_zs_outer + zi * p.o_zs_inner;

    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1) + a_zoff;
    const uint16_t* a2 = p.a2 ? static_cast<const uint16_t*>(p.a2) + a_zoff : nullptr;
    const uint16_t* wp = static_cast<const uint16_t*>(p.w) + w_zoff;

    // ---- per-thread load geometry: chunk column kc, rows (tid>>3) + 32*i
    const int kc = tid & 7, r0 = tid >> 3;
    const int Cin = p.C1 + p.C2;
    const int LH = p.upsample2x ? p.IH * 2 : p.IH, LW = p.upsample2x ? p.IW * 2 : p.IW;

    int a_iy0[RA], a_ix0[RA], a_pix[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < p.M;
        if (SPATIAL) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
            a_pix[i] = b * p.IH * p.IW;
        } else {
            a_iy0[i] = 0; a_ix0[i] = 0; a_pix[i] = m;
        }
    }
    const int nvalid = p.n_valid > 0 ? p.n_valid : p.N;
    bool w_ok[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) w_ok[i] = (n0 + r0 + 32 * i) < nvalid;

    U4 ra[RA], rw[RW];

    auto load_tile = [&](int kt) {
        const int kg = kt * BK + kc * 8;
        const bool kvalid = kg < p.K;
        int c = (p.a_wrap > 0 && kg >= p.a_wrap) ? kg - p.a_wrap : kg, ky = 0, kx = 0;      // a_wrap: the A columns are read twice (K = 2 a_wrap)
        if (SPATIAL && p.taps == 9) {
            const int tap = kg / Cin;
            c = kg - tap * Cin;
            ky = tap / 3;
            kx = tap - 3 * ky;
        }
        const bool second = c >= p.C1;
        const uint16_t* base = second ? a2 : a1;
        const int ld = second ? p.ld2 : p.ld1;
        const int cc = second ? c - p.C1 : c;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bool ok = a_ok[i] && kvalid;
            int64_t pix;
            if (SPATIAL) {
                int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                ok = ok && iy >= 0 && iy < LH && ix >= 0 && ix < LW;
                if (p.upsample2x) { iy >>= 1; ix >>= 1; }
                pix = (int64_t)a_pix[i] + iy * p.IW + ix;
            } else {
                pix = a_pix[i];
            }
            ra[i] = zero16();
            if (ok) ra[i] = ldg16(base + pix * ld + cc);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            rw[i] = zero16();
            if (w_ok[i] && kvalid) rw[i] = ldg16(wp + (int64_t)(n0 + r0 + 32 * i) * p.ldw + kg);
        }
    };
    auto store_tile = [&](int buf) {
        char* sa = smem + buf * STAGE;
        char* sw = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<U4*>(sa + tile_off(r0 + 32 * i, kc)) = ra[i];
#pragma unroll
        for (int i = 0; i < RW; ++i) *reinterpret_cast<U4*>(sw + tile_off(r0 + 32 * i, kc)) = rw[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int nkt_all = (p.K + BK - 1) / BK;
    int kt0 = 0, nkt = nkt_all;
    if (p.splitk > 1) {   // blockIdx.y = K split: a contiguous run of K-tiles, partial sums go to the workspace
        const int per = (nkt_all + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        nkt = min(per, nkt_all - kt0);
        if (nkt < 0) nkt = 0;
    }
    if (nkt > 0) {
        load_tile(kt0);
        store_tile(0);
        if (nkt > 1) load_tile(kt0 + 1);
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const char* sa = smem + cur * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int c = 2 * ks + lh;
            U4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const U4*>(sa + tile_off(wm * 32 * MI + mi * 32 + l31, c));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bf[ni] = *reinterpret_cast<const U4*>(sw + tile_off(wn * 32 * NI + ni * 32 + l31, c));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T::mfma(af[mi], bf[ni], acc[mi][ni]);
        }
        if (kt + 1 < nkt) {
            store_tile(cur ^ 1);               // registers hold tile kt+1 (loaded one iteration ago)
            if (kt + 2 < nkt) load_tile(kt0 + kt + 2);  // in flight during the next iteration's MFMAs
        }
        __syncthreads();
    }

    tile_epilogue<T, MI, NI>(p, acc, smem, m0, n0, o_zoff);
}


// 16 bytes of zeros in device memory: the LDS-DMA source of every out-of-range chunk (conv halo, M / N tails).
__device__ __attribute__((aligned(16))) uint32_t g_zero_chunk[4];

// ------------------------------------------------------------------------------------------------------
// LDS-DMA main loop (128x128x64 tile): operands go global -> LDS directly (global_load_lds_dwordx4), no VGPR
// staging and no ds_write traffic.  One wave instruction fills 1 KiB = 8 tile rows; the XOR swizzle is applied
// on the SOURCE side (lane (row, slot) fetches logical chunk slot ^ f(row)) because the DMA destination is
// lane-linear.  Out-of-range chunks fetch g_zero_chunk.  Two LDS buffers, tile t+1 in flight while tile t is
// multiplied: counted s_waitcnt vmcnt(8) + raw s_barrier (a __syncthreads() would drain the prefetch).
// Requires (C1 % 64 == 0, C2 == 0): every 64-wide K-tile lies inside one filter tap.
// ------------------------------------------------------------------------------------------------------
template <typename T, bool SPATIAL, bool FAST>
__global__ void __launch_bounds__(kThreads, 2) igemm_dma_kernel(const edtr_igemm_params p) {
    EDTR_STAMP(0); EDTR_STAMP(6); EDTR_STAMP(5);
    stagger_second_slot(p);
    constexpr int MI = 2, NI = 2, BM = 128, BN = 128;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nblk = nbm * nbn, q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn;
    tile_coords(p, bid, nbm, nbn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    EDTR_STAMP_T(15);

    const int z = blockIdx.z, zo = z / p.zdiv, zi = z - zo * p.zdiv;
    const int64_t a_zoff = zo * p.a_zs_outer + zi * p.a_zs_inner;
    const int64_t w_zoff = zo * p.w_zs_outer + zi * p.w_zs_inner;
    const int64_t o_zoff = zo * p.o_zs_outer + zi * p.o_zs_inner;
    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1) + a_zoff;
    const uint16_t* wp = static_cast<const uint16_t*>(p.w) + w_zoff;
    const void* zsrc = g_zero_chunk;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    // this lane's 4 A rows / 4 W rows: tile row = wave*32 + 8*j + (lane>>3); LDS slot = lane&7
    const int rsub = lane >> 3, slot = lane & 7;
    const int Cin = p.C1;
    const int LH = p.upsample2x ? p.IH * 2 : p.IH, LW = p.upsample2x ? p.IW * 2 : p.IW;
    int a_iy0[4], a_ix0[4], a_pix[4], coff[4];
    bool a_ok[4], w_ok[4];
    int64_t w_row[4];
    const int nvalid = p.n_valid > 0 ? p.n_valid : p.N;
    // (image, oy, ox) of the tile's first row by ONE exact division pair on wave-uniform values; the 128 rows of the
    // tile are then reached with two small float-reciprocal divmods each (operands < 2^20, one fix-up step either
    // way makes them exact) instead of two 32-bit integer divisions per row (~40 VALU instructions each).
    int b0 = 0, oy0 = 0, ox0 = 0;
    float rcp_ow = 0.0f, rcp_oh = 0.0f;
    if (SPATIAL) {
        const int hw = p.OH * p.OW;
        b0 = m0 / hw;
        const int rem0 = m0 - b0 * hw;
        oy0 = rem0 / p.OW;
        ox0 = rem0 - oy0 * p.OW;
        rcp_ow = 1.0f / (float)p.OW;
        rcp_oh = 1.0f / (float)p.OH;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + 8 * j + rsub;
        coff[j] = (slot ^ ((row >> 1) & 7)) * 8;        // logical 8-element chunk held by this LDS slot
        const int m = m0 + row;
        a_ok[j] = m < p.M;
        if (SPATIAL) {
            const int x = ox0 + row;
            int qx = (int)((float)x * rcp_ow), ox = x - qx * p.OW;
            if (ox >= p.OW) { ++qx; ox -= p.OW; }
            if (ox < 0) { --qx; ox += p.OW; }
            const int y = oy0 + qx;
            int qy = (int)((float)y * rcp_oh), oy = y - qy * p.OH;
            if (oy >= p.OH) { ++qy; oy -= p.OH; }
            if (oy < 0) { --qy; oy += p.OH; }
            a_iy0[j] = oy * p.stride - p.pad_t;
            a_ix0[j] = ox * p.stride - p.pad_l;
            a_pix[j] = (b0 + qy) * p.IH * p.IW;
        } else {
            a_iy0[j] = 0; a_ix0[j] = 0; a_pix[j] = m;
        }
        const int n = n0 + row;
        w_ok[j] = n < nvalid;
        w_row[j] = (int64_t)n * p.ldw;
    }
    // FAST path (no upsample, tensors < 4 GiB): buffer-addressed LDS-DMA.  Each lane keeps ONE constant 32-bit byte
    // offset per row (its pixel's centre tap / its weight row) and a 9-bit "tap in range" mask; a K-tile only moves the
    // wave-uniform soffset (SALU).  Halo / tail lanes present an out-of-range offset and the buffer range check makes
    // the DMA write zeros (verified on gfx950: tools/exp/buffer_lds_probe.hip) — no zero page, no 64-bit VALU address math.
    uint32_t voff_a[4], voff_w[4], a_mask[4], a_par[4];
    u32x4 srd_a, srd_w;
    if constexpr (FAST) {
        const int64_t bias = SPATIAL ? ((int64_t)p.pad_t * p.IW + p.pad_l) * p.ld1 * 2 : 0;
        srd_a = make_srd(reinterpret_cast<const char*>(a1) - bias);
        srd_w = make_srd(wp);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t mask = 0;
            a_par[j] = 0;
            if (SPATIAL) {
                // output pixel in the (possibly 2x upsampled) logical input grid; source pixel of the centre tap
                const int ly = a_iy0[j] + p.pad_t, lx = a_ix0[j] + p.pad_l;
                const int sy = p.upsample2x ? ly >> 1 : ly, sx = p.upsample2x ? lx >> 1 : lx;
                a_par[j] = p.upsample2x ? (uint32_t)((ly & 1) | ((lx & 1) << 1)) : 0u;
                const int64_t pc = (int64_t)a_pix[j] + (int64_t)sy * p.IW + sx;
                voff_a[j] = (uint32_t)((pc * p.ld1 + coff[j]) * 2);
                // tap t = 3 ky + kx is in range iff row ky and column kx are: mask = rowbits (x) colbits
                if (p.taps == 9) {
                    uint32_t rb = 0, cbits = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int iy = a_iy0[j] + k, ix = a_ix0[j] + k;
                        rb |= (iy >= 0 && iy < LH) ? (1u << (3 * k)) : 0u;
                        cbits |= (ix >= 0 && ix < LW) ? (1u << k) : 0u;
                    }
                    mask = a_ok[j] ? rb * cbits : 0u;
                } else {
                    const int iy = a_iy0[j] + p.pad_t, ix = a_ix0[j] + p.pad_l;
                    mask = (a_ok[j] && iy >= 0 && iy < LH && ix >= 0 && ix < LW) ? 1u : 0u;
                }
            } else {
                voff_a[j] = a_ok[j] ? (uint32_t)(((int64_t)a_pix[j] * p.ld1 + coff[j]) * 2) : kOobOffset;
                mask = 1u;
            }
            a_mask[j] = mask;
            voff_w[j] = w_ok[j] ? (uint32_t)((w_row[j] + coff[j]) * 2) : kOobOffset;
        }
    }
    int run_tap = 0, run_c0 = 0;   // (tap, channel offset) of the NEXT tile to issue; tiles are issued in order
    // (fast path) convs walk K channel-chunk major / tap minor, see igemm_256_kernel: the nine taps of a chunk hit L2

    // Per-lane A offsets of the CURRENT filter tap (halo / tail lanes -> out of range), recomputed only when the tap
    // changes (every Cin/64 K-tiles); within a tap a K-tile moves just the wave-uniform soffset, so the issue of a
    // K-tile is 8 x (M0 write + buffer_load ... lds) and a handful of SALU instructions.
    uint32_t vsel[4] = {0u, 0u, 0u, 0u};
    uint32_t soff_tap = 0;
    int sel_tap = -1;
    auto select_tap = [&](int tap) {
        if constexpr (FAST && SPATIAL) {
            int ky = p.pad_t, kx = p.pad_l;     // 1x1 conv in spatial mode: the centre tap
            uint32_t tapbit = 1u;
            if (p.taps == 9) { ky = (tap * 11) >> 5; kx = tap - 3 * ky; tapbit = 1u << tap; }
            if (p.upsample2x) {
                // source row of logical row ly + ky - pad is (ly >> 1) + ((parity + ky - pad) >> 1); bias included, >= 0
                const int rowb = p.IW * p.ld1 * 2, colb = p.ld1 * 2;
                const uint32_t dy0 = (uint32_t)((((0 + ky - p.pad_t) >> 1) + p.pad_t) * rowb);
                const uint32_t dy1 = (uint32_t)((((1 + ky - p.pad_t) >> 1) + p.pad_t) * rowb);
                const uint32_t dx0 = (uint32_t)((((0 + kx - p.pad_l) >> 1) + p.pad_l) * colb);
                const uint32_t dx1 = (uint32_t)((((1 + kx - p.pad_l) >> 1) + p.pad_l) * colb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t vo = voff_a[j] + ((a_par[j] & 1u) ? dy1 : dy0) + ((a_par[j] & 2u) ? dx1 : dx0);
                    vsel[j] = (a_mask[j] & tapbit) ? vo : kOobOffset;
                }
                soff_tap = 0;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) vsel[j] = (a_mask[j] & tapbit) ? voff_a[j] : kOobOffset;
                soff_tap = (uint32_t)(((ky * p.IW + kx) * p.ld1) * 2);
            }
        }
    };

    auto issue_tile = [&](int kt, int buf) {
        if constexpr (FAST) {
            uint32_t soff_a;
            uint32_t soff_w = (uint32_t)kt * (BK * 2);
            if constexpr (SPATIAL) {
                if (run_tap != sel_tap) { select_tap(run_tap); sel_tap = run_tap; }
                soff_a = soff_tap + (uint32_t)(run_c0 * 2);
                soff_w = (uint32_t)((run_tap * Cin + run_c0) * 2);   // the weight tile of (tap, chunk) in [Cout][ky][kx][Cin]
                if (++run_tap == p.taps) { run_tap = 0; run_c0 += BK; }   // next tap of the same 64-channel chunk
            } else {
                soff_a = (uint32_t)((p.a_wrap > 0 && kt * BK >= p.a_wrap) ? kt * BK - p.a_wrap : kt * BK) * 2;      // a_wrap: A read twice
            }
            const uint32_t sa = smem_base + buf * STAGE + wave * (32 * 128);
            const uint32_t sw = sa + A_BYTES;
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16_buf(SPATIAL ? vsel[j] : voff_a[j], srd_a, soff_a, sa + j * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16_buf(voff_w[j], srd_w, soff_w, sw + j * 1024);
            return;
        }
        const int k0 = kt * BK;
        int c0 = (p.a_wrap > 0 && k0 >= p.a_wrap) ? k0 - p.a_wrap : k0, ky = 0, kx = 0;
        if (SPATIAL && p.taps == 9) {
            const int tap = k0 / Cin;
            c0 = k0 - tap * Cin;
            ky = tap / 3;
            kx = tap - 3 * ky;
        }
        const uint32_t sa = smem_base + buf * STAGE + wave * (32 * 128);
        const uint32_t sw = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = a_ok[j];
            int64_t pix;
            if (SPATIAL) {
                int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
                ok = ok && iy >= 0 && iy < LH && ix >= 0 && ix < LW;
                if (p.upsample2x) { iy >>= 1; ix >>= 1; }
                pix = (int64_t)a_pix[j] + iy * p.IW + ix;
            } else {
                pix = a_pix[j];
            }
            const void* src = ok ? static_cast<const void*>(a1 + pix * p.ld1 + c0 + coff[j]) : zsrc;
            dma16(src, sa + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const void* src = w_ok[j] ? static_cast<const void*>(wp + w_row[j] + k0 + coff[j]) : zsrc;
            dma16(src, sw + j * 1024);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int nkt_all = p.K / BK;
    int kt0 = 0, nkt = nkt_all;
    if (p.splitk > 1) {
        const int per = (nkt_all + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        nkt = min(per, nkt_all - kt0);
        if (nkt < 0) nkt = 0;
    }
    if (SPATIAL) {
        run_c0 = (kt0 / p.taps) * BK;
        run_tap = kt0 - (kt0 / p.taps) * p.taps;
    }
    EDTR_STAMP(1);
    if (nkt > 0) issue_tile(kt0, 0);

#ifdef EDTR_STAMPS
    uint64_t stamp_acc[4] = {0, 0, 0, 0};
#endif
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
#ifdef EDTR_STAMPS
        const uint64_t ts0 = __builtin_amdgcn_s_memtime();
#endif
        if (kt + 1 < nkt) {
            issue_tile(kt0 + kt + 1, cur ^ 1);
#ifdef EDTR_STAMPS
            stamp_acc[0] += __builtin_amdgcn_s_memtime() - ts0;
#endif
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's 8 DMAs of tile kt have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef EDTR_STAMPS
        const uint64_t ts1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();                           // ... and every other wave's
        asm volatile("" ::: "memory");
#ifdef EDTR_STAMPS
        const uint64_t ts2 = __builtin_amdgcn_s_memtime();
        stamp_acc[1] += ts1 - ts0;      // issue + own vmcnt wait
        stamp_acc[2] += ts2 - ts1;      // barrier wait
        if (kt == 0) EDTR_STAMP(2);
#endif
        const char* sa = smem + cur * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int c = 2 * ks + lh;
            U4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const U4*>(sa + tile_off(wm * 64 + mi * 32 + l31, c));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bf[ni] = *reinterpret_cast<const U4*>(sw + tile_off(wn * 64 + ni * 32 + l31, c));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T::mfma(af[mi], bf[ni], acc[mi][ni]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef EDTR_STAMPS
        stamp_acc[3] += __builtin_amdgcn_s_memtime() - ts2;     // ds_read + MFMA issue section
#endif
        __builtin_amdgcn_s_barrier();                           // buffer `cur` may be refilled next iteration
        asm volatile("" ::: "memory");
    }
    if (nkt == 0) __syncthreads();
    EDTR_STAMP(3);
#ifdef EDTR_STAMPS
#undef STAMP_VALUE
#define STAMP_VALUE stamp_acc[0]
    EDTR_STAMP(8);
#undef STAMP_VALUE
#define STAMP_VALUE stamp_acc[1]
    EDTR_STAMP(9);
#undef STAMP_VALUE
#define STAMP_VALUE stamp_acc[2]
    EDTR_STAMP(10);
#undef STAMP_VALUE
#define STAMP_VALUE stamp_acc[3]
    EDTR_STAMP(11);
#undef STAMP_VALUE
#define STAMP_VALUE 0
#endif
    tile_epilogue<T, MI, NI>(p, acc, smem, m0, n0, o_zoff);
    EDTR_STAMP(4); EDTR_STAMP(7);
}

template <typename T, bool SPATIAL, bool FAST>
int launch_dma(const edtr_igemm_params& p, hipStream_t stream) {
    constexpr int lds = 2 * (128 + 128) * BK * 2;
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_dma_kernel<T, SPATIAL, FAST>), lds, attr_set)) return rc_;
    const int nbm = (p.M + 127) / 128, nbn = (p.N + 127) / 128;
    dim3 grid(nbm * nbn, p.splitk > 1 ? p.splitk : 1, p.Z);
    hipLaunchKernelGGL((igemm_dma_kernel<T, SPATIAL, FAST>), grid, dim3(kThreads), lds, stream, p);
    EDTR_LAUNCH_CHECK();
    if (p.splitk > 1) return launch_splitk_reducer<T>(p, stream);
    return EDTR_OK;
}


// ------------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves, ping-pong schedule (tile = 6).  The structure the 128x128 two-barrier loop cannot
// reach (its L2->LDS traffic is 64 B/clk/CU at MFMA peak and its DMA issue, LDS reads and MFMAs run back to back
// in each wave): a workgroup of 8 waves per CU, waves w and w+4 share a SIMD and run HALF A PHASE APART (waves
// 4-7 take one extra barrier before the loop), so on every SIMD one wave issues its 16 MFMAs while its partner
// issues LDS reads and the LDS-DMA of a future half-tile.
//   * wave (wr = w>>2, wc = w&3) owns rows {wr*64 .. +63} of BOTH 128-row A halves and columns {wc*32 .. +31} of
//     BOTH 128-column B halves: 4 quadrants of 64x32, one quadrant x K=64 per phase (16 x v_mfma_f32_16x16x32).
//   * a K-tile = 4 half-tiles of 16 KiB (A_lo, A_hi, B_lo, B_hi; [128][64] 16-bit, XOR-swizzled on the source
//     side); two K-tile buffers = 128 KiB LDS.  An iteration = 2 K-tiles = 8 phases; every phase stages ONE
//     half-tile (2 DMAs per wave):  even tile t+2: B_lo@3 A_lo@4 B_hi@5 A_hi@6, odd tile t+3: B_lo@7 A_lo@8,
//     B_hi@1 A_hi@2 (next iteration).  Reads: A_lo,B_lo @1/5, B_hi @2/6, A_hi @3/7 (B_lo stays in registers for
//     phase 4/8).  Every restage is >= 2 phases after the last read of its slot (WAR), every phase waits
//     vmcnt(6) = "all but the 3 newest half-tiles landed" before its first barrier, and a slot is read >= 1
//     phase after the wait that retires it (RAW) — three half-tiles stay in flight across the barriers.
// Buffer addressing only (operands < 4 GiB), Cin % 64 == 0, no GEGLU / split-K.  Epilogue: two passes of
// 128 rows through the 128 KiB of LDS, then the shared row-vector phase (and the fused GroupNorm partials).
// ------------------------------------------------------------------------------------------------------
struct KState { int t, tap, c0; };

template <typename T, bool SPATIAL>
__global__ void __launch_bounds__(512, 1) igemm_256_kernel(const edtr_igemm_params p) {
    constexpr int HALF = 128 * BK * 2;       // bytes of one half-tile
    constexpr int BUF = 4 * HALF;            // A_lo, A_hi, B_lo, B_hi
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;

    const int nbm = (p.M + 255) / 256, nbn = (p.N + 255) / 256;
    int bid = blockIdx.x;
    {
        const int nblk = nbm * nbn, q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn;
    tile_coords(p, bid, nbm, nbn, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    const int z = blockIdx.z, zo = z / p.zdiv, zi = z - zo * p.zdiv;
    const int64_t a_zoff = zo * p.a_zs_outer + zi * p.a_zs_inner;
    const int64_t w_zoff = zo * p.w_zs_outer + zi * p.w_zs_inner;
    const int64_t o_zoff = zo * p.o_zs_outer + zi * p.o_zs_inner;
    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1) + a_zoff;
    const uint16_t* wp = static_cast<const uint16_t*>(p.w) + w_zoff;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    // ---- staging geometry: this wave fills rows wave*16 + 8*j + (lane>>3), j = 0,1, of every half-tile
    const int rsub = lane >> 3, slot = lane & 7;
    const int Cin = p.C1;
    const int LH = p.upsample2x ? p.IH * 2 : p.IH, LW = p.upsample2x ? p.IW * 2 : p.IW;
    const int nvalid = p.n_valid > 0 ? p.n_valid : p.N;
    uint32_t voff_a[4], voff_w[4], a_mask[4], a_par[4];   // index = half * 2 + j
    int b0 = 0, oy0 = 0, ox0 = 0;
    float rcp_ow = 0.0f, rcp_oh = 0.0f;
    if (SPATIAL) {
        const int hw = p.OH * p.OW;
        b0 = m0 / hw;
        const int rem0 = m0 - b0 * hw;
        oy0 = rem0 / p.OW;
        ox0 = rem0 - oy0 * p.OW;
        rcp_ow = 1.0f / (float)p.OW;
        rcp_oh = 1.0f / (float)p.OH;
    }
    const int64_t a_bias = SPATIAL ? ((int64_t)p.pad_t * p.IW + p.pad_l) * p.ld1 * 2 : 0;
    const u32x4 srd_a = make_srd(reinterpret_cast<const char*>(a1) - a_bias);
    const u32x4 srd_w = make_srd(wp);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rih = wave * 16 + 8 * (q & 1) + rsub;               // row inside the half-tile
        const int row = (q >> 1) * 128 + rih;                         // row inside the 256-row tile
        const int coff = (slot ^ ((rih >> 1) & 7)) * 8;               // logical 8-element chunk held by this LDS slot
        const int m = m0 + row;
        const bool ok = m < p.M;
        uint32_t mask = 0;
        a_par[q] = 0;
        if (SPATIAL) {
            const int x = ox0 + row;
            int qx = (int)((float)x * rcp_ow), ox = x - qx * p.OW;
            if (ox >= p.OW) { ++qx; ox -= p.OW; }
            if (ox < 0) { --qx; ox += p.OW; }
            const int y = oy0 + qx;
            int qy = (int)((float)y * rcp_oh), oy = y - qy * p.OH;
            if (oy >= p.OH) { ++qy; oy -= p.OH; }
            if (oy < 0) { --qy; oy += p.OH; }
            const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
            const int ly = iy0 + p.pad_t, lx = ix0 + p.pad_l;
            const int sy = p.upsample2x ? ly >> 1 : ly, sx = p.upsample2x ? lx >> 1 : lx;
            a_par[q] = p.upsample2x ? (uint32_t)((ly & 1) | ((lx & 1) << 1)) : 0u;
            const int64_t pc = (int64_t)(b0 + qy) * p.IH * p.IW + (int64_t)sy * p.IW + sx;
            voff_a[q] = (uint32_t)((pc * p.ld1 + coff) * 2);
            if (p.taps == 9) {
                uint32_t rb = 0, cbits = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int iy = iy0 + k, ix = ix0 + k;
                    rb |= (iy >= 0 && iy < LH) ? (1u << (3 * k)) : 0u;
                    cbits |= (ix >= 0 && ix < LW) ? (1u << k) : 0u;
                }
                mask = ok ? rb * cbits : 0u;
            } else {
                mask = (ok && ly >= 0 && ly < LH && lx >= 0 && lx < LW) ? 1u : 0u;
            }
        } else {
            voff_a[q] = ok ? (uint32_t)(((int64_t)m * p.ld1 + coff) * 2) : kOobOffset;
            mask = 1u;
        }
        a_mask[q] = mask;
        const int n = n0 + row;
        voff_w[q] = n < nvalid ? (uint32_t)(((int64_t)n * p.ldw + coff) * 2) : kOobOffset;
    }

    const int nkt = p.K / BK;
    // K order of a 3x3 conv: channel-chunk major, tap minor (K-tile t = chunk t/9, tap t%9).  The nine taps of one
    // 64-channel chunk re-read (shifted) the same input lines back to back, so they hit L2 while a workgroup walks its
    // K loop; tap-major order returned to a line only after Cin/64 tiles, by which time the ~4 MiB of input that the 32
    // resident workgroups of an XCD cover had evicted it (PMC: 2.6-5x the algorithmic fetch bytes).  The weight tile of
    // (tap, chunk) is simply at K offset tap*Cin + chunk*64 of the unchanged [Cout][ky][kx][Cin] packing.
    auto advance2 = [&](KState& s) {
        s.t += 2;
        if (SPATIAL) {
            s.tap += 2;
            while (s.tap >= p.taps) { s.tap -= p.taps; s.c0 += BK; }
        }
    };
    // stage half-tile `H` (0 = A_lo, 1 = A_hi, 2 = B_lo, 3 = B_hi) of K-tile s.t into buffer `buf`
    auto stage_half = [&](const KState& s, int buf, auto Hc) {
        constexpr int H = decltype(Hc)::value;
        const bool live = s.t < nkt;
        const uint32_t dst = smem_base + buf * BUF + H * HALF + wave * 2048;
        if constexpr (H < 2) {
            uint32_t soff = (uint32_t)s.t * (BK * 2), tapbit = 1u;
            uint32_t dy0 = 0, dy1 = 0, dx0 = 0, dx1 = 0;
            if (SPATIAL) {
                int ky = p.pad_t, kx = p.pad_l;
                if (p.taps == 9) { ky = (s.tap * 11) >> 5; kx = s.tap - 3 * ky; tapbit = 1u << s.tap; }
                if (p.upsample2x) {
                    const int rowb = p.IW * p.ld1 * 2, colb = p.ld1 * 2;
                    dy0 = (uint32_t)((((0 + ky - p.pad_t) >> 1) + p.pad_t) * rowb);
                    dy1 = (uint32_t)((((1 + ky - p.pad_t) >> 1) + p.pad_t) * rowb);
                    dx0 = (uint32_t)((((0 + kx - p.pad_l) >> 1) + p.pad_l) * colb);
                    dx1 = (uint32_t)((((1 + kx - p.pad_l) >> 1) + p.pad_l) * colb);
                    soff = (uint32_t)(s.c0 * 2);
                } else {
                    soff = (uint32_t)(((ky * p.IW + kx) * p.ld1 + s.c0) * 2);
                }
            }
            if (!live) tapbit = 0u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                constexpr int q0 = H * 2;
                uint32_t vo = voff_a[q0 + j];
                if (SPATIAL && p.upsample2x) vo += ((a_par[q0 + j] & 1u) ? dy1 : dy0) + ((a_par[q0 + j] & 2u) ? dx1 : dx0);
                vo = (a_mask[q0 + j] & tapbit) ? vo : kOobOffset;
                dma16_buf(vo, srd_a, soff, dst + j * 1024);
            }
        } else {
            const uint32_t soff = SPATIAL ? (uint32_t)((s.tap * Cin + s.c0) * 2) : (uint32_t)s.t * (BK * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                constexpr int q0 = (H - 2) * 2;
                const uint32_t vo = live ? voff_w[q0 + j] : kOobOffset;
                dma16_buf(vo, srd_w, soff, dst + j * 1024);
            }
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using H2 = std::integral_constant<int, 2>;
    using H3 = std::integral_constant<int, 3>;

    // ---- fragment read geometry (conflict-free ds_read_b128 of the swizzled image, see common.h tile_off)
    const int a_rd = tile_off(wr * 64 + l15, lq);      // + blk * 2048, ^ 64 for the second k-step
    const int b_rd = tile_off(wc * 32 + l15, lq);

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[i][j][a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    U4 afr[4][2], bfr[2][2][2];

    auto read_a = [&](int buf, int ah) {
        const char* base = smem + buf * BUF + ah * HALF;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            afr[mb][0] = *reinterpret_cast<const U4*>(base + a_rd + mb * 2048);
            afr[mb][1] = *reinterpret_cast<const U4*>(base + (a_rd ^ 64) + mb * 2048);
        }
    };
    auto read_b = [&](int buf, auto BHc) {
        constexpr int BH = decltype(BHc)::value;
        const char* base = smem + buf * BUF + (2 + BH) * HALF;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            bfr[BH][nb][0] = *reinterpret_cast<const U4*>(base + b_rd + nb * 2048);
            bfr[BH][nb][1] = *reinterpret_cast<const U4*>(base + (b_rd ^ 64) + nb * 2048);
        }
    };
    auto mma = [&](auto AHc, auto BHc) {
        constexpr int AH = decltype(AHc)::value, BH = decltype(BHc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[AH][BH][mb][nb] = T::mfma16(afr[mb][ks], bfr[BH][nb][ks], acc[AH][BH][mb][nb]);
    };
    // second half of every phase: wait for the half-tiles that must have landed, rendezvous, multiply, rendezvous
    auto compute = [&](auto AHc, auto BHc) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma(AHc, BHc);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: K-tile 0 complete, K-tile 1's B_lo / A_lo (its B_hi / A_hi are staged by phases 1 and 2)
    KState sE{0, 0, 0}, sO{1, 1, 0};
    if (!SPATIAL || p.taps == 1) { sO.tap = 0; sO.c0 = BK; }
    stage_half(sE, 0, H0{}); stage_half(sE, 0, H1{}); stage_half(sE, 0, H2{}); stage_half(sE, 0, H3{});
    stage_half(sO, 1, H2{}); stage_half(sO, 1, H0{});
    advance2(sE);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run half a phase behind their SIMD partners
    asm volatile("" ::: "memory");

    const int niter = (nkt + 1) >> 1;
    for (int it = 0; it < niter; ++it) {
        // even K-tile (buffer 0)
        read_b(0, H0{}); read_a(0, 0); stage_half(sO, 1, H3{}); compute(H0{}, H0{});                 // phase 1
        read_b(0, H1{});               stage_half(sO, 1, H1{}); compute(H0{}, H1{}); advance2(sO);   // phase 2
        read_a(0, 1);                  stage_half(sE, 0, H2{}); compute(H1{}, H1{});                 // phase 3
                                       stage_half(sE, 0, H0{}); compute(H1{}, H0{});                 // phase 4
        // odd K-tile (buffer 1)
        read_b(1, H0{}); read_a(1, 0); stage_half(sE, 0, H3{}); compute(H0{}, H0{});                 // phase 5
        read_b(1, H1{});               stage_half(sE, 0, H1{}); compute(H0{}, H1{}); advance2(sE);   // phase 6
        read_a(1, 1);                  stage_half(sO, 1, H2{}); compute(H1{}, H1{});                 // phase 7
                                       stage_half(sO, 1, H0{}); compute(H1{}, H0{});                 // phase 8
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two wave groups
    __syncthreads();

    // ---- epilogue: rows of A half `h` (128 x 256 fp32 = the whole 128 KiB) per pass
    float* stage = reinterpret_cast<float*>(smem);
    const bool gn_acc = p.gn_partial != nullptr && p.splitk <= 1;      // (split-K: the reducer writes the statistics)
    // (two passes written as two calls of one generic lambda with a compile-time pass index: as a `#pragma unroll` loop the body —
    //  rows_phase and its specialised row loops — outgrew the unroller in round 4, the loop stayed rolled, `acc[h]` became a
    //  dynamically indexed array and the 128 accumulators went through 528 bytes of scratch per lane)
    auto epilogue_pass = [&](auto HPc) {
        constexpr int h = decltype(HPc)::value;
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(wr * 64 + mb * 16 + 4 * lq + r) * 256 + bh * 128 + wc * 32 + nb * 16 + l15] = acc[h][bh][mb][nb][r];
        float gs[8], gq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gs[j] = 0.0f; gq[j] = 0.0f; }
        rows_phase<T, 128, 256, false, 512, 0, 256, false>(p, stage, m0 + h * 128, n0, p.N, o_zoff, gn_acc, gs, gq);      // (FOLD off: only tiles 1 / 3 carry the folded-LayerNorm paths)
        __syncthreads();                       // every thread is done reading the staged rows
        if (gn_acc) {
            // thread (row group tid/32, column group tid%32): lanes l and l^32 share a column group; fold, then the 8 waves
#pragma unroll
            for (int j = 0; j < 8; ++j) { gs[j] += __shfl_xor(gs[j], 32, 64); gq[j] += __shfl_xor(gq[j], 32, 64); }
            if (lane < 32) {
                float* dst = stage + (wave * 256 + lane * 8) * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[j]; dst[2 * j + 1] = gq[j]; }
            }
            __syncthreads();
            if (tid < 256 && n0 + tid < p.N && m0 + h * 128 < p.M) {
                float a = 0.0f, q = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) { a += stage[(w * 256 + tid) * 2]; q += stage[(w * 256 + tid) * 2 + 1]; }
                float* dst = p.gn_partial + ((int64_t)((m0 >> 7) + h) * gn_ld_of(p) + n0 + tid) * 2;
                dst[0] = a;
                dst[1] = q;
            }
            __syncthreads();
        }
    };
    epilogue_pass(H0{});
    epilogue_pass(H1{});
}

template <typename T, bool SPATIAL>
int launch_256(const edtr_igemm_params& p, hipStream_t stream) {
    constexpr int lds = 2 * 4 * 128 * BK * 2;     // 128 KiB
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_256_kernel<T, SPATIAL>), lds, attr_set)) return rc_;
    const int nbm = (p.M + 255) / 256, nbn = (p.N + 255) / 256;
    dim3 grid(nbm * nbn, 1, p.Z);
    hipLaunchKernelGGL((igemm_256_kernel<T, SPATIAL>), grid, dim3(512), lds, stream, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

// ------------------------------------------------------------------------------------------------------
// 128x160x64 tile (tile = 8): the SD-2.1 channel counts are 320 / 640 / 1280 = 2 / 4 / 8 x 160, so a 160-column
// tile wastes nothing where the 128-column grid pads N = 320 to 384 (one third-empty tile in three), and at the
// 64x64-latent level (M = 32768, N = 320) it makes 256 x 2 = 512 workgroups — exactly one resident round at two
// workgroups per CU instead of 768 (one and a half).  Same two-stage buffer-addressed LDS-DMA loop as tile 3; 4 waves
// as 2 x 2, a wave owns 64 rows x 80 columns = 4 x 5 blocks of v_mfma_f32_16x16x32 (80 accumulator registers).
// Per K-tile a wave issues 4 A + 5 W pieces (vmcnt(9)).  LDS: 2 x (16 + 20) KiB; the epilogue stages 64 rows x 160
// columns of fp32 (40 KiB) per pass, two passes.  Buffer addressing only, no GEGLU, no split-K.
// ------------------------------------------------------------------------------------------------------
template <typename T, bool SPATIAL, int MB, int NB>
__global__ void __launch_bounds__(kThreads, 2) igemm_n160_kernel(const edtr_igemm_params p) {
    // wave tile = (16 MB) x (16 NB) as MB x NB blocks of 16x16; workgroup tile = 2 x 2 waves
    constexpr int BM = 32 * MB, BN = 32 * NB;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EDTR_STAMP(0); EDTR_STAMP(6); EDTR_STAMP(5);
    stagger_second_slot(p);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nblk = nbm * nbn, q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn;
    tile_coords(p, bid, nbm, nbn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.z, zo = z / p.zdiv, zi = z - zo * p.zdiv;
    const int64_t a_zoff = zo * p.a_zs_outer + zi * p.a_zs_inner;
    const int64_t w_zoff = zo * p.w_zs_outer + zi * p.w_zs_inner;
    const int64_t o_zoff = zo * p.o_zs_outer + zi * p.o_zs_inner;
    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1) + a_zoff;
    const uint16_t* wp = static_cast<const uint16_t*>(p.w) + w_zoff;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    // ---- staging geometry: A rows wave*8MB + 8j + (lane>>3), j < MB; W rows wave*8NB + 8j + (lane>>3), j < NB
    const int rsub = lane >> 3, slot = lane & 7;
    const int Cin = p.C1;
    const int LH = p.upsample2x ? p.IH * 2 : p.IH, LW = p.upsample2x ? p.IW * 2 : p.IW;
    const int nvalid = p.n_valid > 0 ? p.n_valid : p.N;
    uint32_t voff_a[MB], voff_w[NB], a_mask[MB], a_par[MB];
    int b0 = 0, oy0 = 0, ox0 = 0;
    float rcp_ow = 0.0f, rcp_oh = 0.0f;
    if (SPATIAL) {
        const int hw = p.OH * p.OW;
        b0 = m0 / hw;
        const int rem0 = m0 - b0 * hw;
        oy0 = rem0 / p.OW;
        ox0 = rem0 - oy0 * p.OW;
        rcp_ow = 1.0f / (float)p.OW;
        rcp_oh = 1.0f / (float)p.OH;
    }
    const int64_t a_bias = SPATIAL ? ((int64_t)p.pad_t * p.IW + p.pad_l) * p.ld1 * 2 : 0;
    const u32x4 srd_a = make_srd(reinterpret_cast<const char*>(a1) - a_bias);
    const u32x4 srd_w = make_srd(wp);
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int row = wave * (8 * MB) + 8 * j + rsub;
        const int coff = (slot ^ ((row >> 1) & 7)) * 8;
        const int m = m0 + row;
        const bool ok = m < p.M;
        uint32_t mask = 0;
        a_par[j] = 0;
        if (SPATIAL) {
            const int x = ox0 + row;
            int qx = (int)((float)x * rcp_ow), ox = x - qx * p.OW;
            if (ox >= p.OW) { ++qx; ox -= p.OW; }
            if (ox < 0) { --qx; ox += p.OW; }
            const int y = oy0 + qx;
            int qy = (int)((float)y * rcp_oh), oy = y - qy * p.OH;
            if (oy >= p.OH) { ++qy; oy -= p.OH; }
            if (oy < 0) { --qy; oy += p.OH; }
            const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
            const int ly = iy0 + p.pad_t, lx = ix0 + p.pad_l;
            const int sy = p.upsample2x ? ly >> 1 : ly, sx = p.upsample2x ? lx >> 1 : lx;
            a_par[j] = p.upsample2x ? (uint32_t)((ly & 1) | ((lx & 1) << 1)) : 0u;
            const int64_t pc = (int64_t)(b0 + qy) * p.IH * p.IW + (int64_t)sy * p.IW + sx;
            voff_a[j] = (uint32_t)((pc * p.ld1 + coff) * 2);
            if (p.taps == 9) {
                uint32_t rb = 0, cbits = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int iy = iy0 + k, ix = ix0 + k;
                    rb |= (iy >= 0 && iy < LH) ? (1u << (3 * k)) : 0u;
                    cbits |= (ix >= 0 && ix < LW) ? (1u << k) : 0u;
                }
                mask = ok ? rb * cbits : 0u;
            } else {
                mask = (ok && ly >= 0 && ly < LH && lx >= 0 && lx < LW) ? 1u : 0u;
            }
        } else {
            voff_a[j] = ok ? (uint32_t)(((int64_t)m * p.ld1 + coff) * 2) : kOobOffset;
            mask = 1u;
        }
        a_mask[j] = mask;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = wave * (8 * NB) + 8 * j + rsub;
        const int coff = (slot ^ ((row >> 1) & 7)) * 8;
        const int n = n0 + row;
        voff_w[j] = n < nvalid ? (uint32_t)(((int64_t)n * p.ldw + coff) * 2) : kOobOffset;
    }

    int run_tap = 0, run_c0 = 0, sel_tap = -1;
    uint32_t vsel[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) vsel[j] = 0u;
    uint32_t soff_tap = 0;
    auto select_tap = [&](int tap) {
        if constexpr (SPATIAL) {
            int ky = p.pad_t, kx = p.pad_l;
            uint32_t tapbit = 1u;
            if (p.taps == 9) { ky = (tap * 11) >> 5; kx = tap - 3 * ky; tapbit = 1u << tap; }
            if (p.upsample2x) {
                const int rowb = p.IW * p.ld1 * 2, colb = p.ld1 * 2;
                const uint32_t dy0 = (uint32_t)((((0 + ky - p.pad_t) >> 1) + p.pad_t) * rowb);
                const uint32_t dy1 = (uint32_t)((((1 + ky - p.pad_t) >> 1) + p.pad_t) * rowb);
                const uint32_t dx0 = (uint32_t)((((0 + kx - p.pad_l) >> 1) + p.pad_l) * colb);
                const uint32_t dx1 = (uint32_t)((((1 + kx - p.pad_l) >> 1) + p.pad_l) * colb);
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const uint32_t vo = voff_a[j] + ((a_par[j] & 1u) ? dy1 : dy0) + ((a_par[j] & 2u) ? dx1 : dx0);
                    vsel[j] = (a_mask[j] & tapbit) ? vo : kOobOffset;
                }
                soff_tap = 0;
            } else {
#pragma unroll
                for (int j = 0; j < MB; ++j) vsel[j] = (a_mask[j] & tapbit) ? voff_a[j] : kOobOffset;
                soff_tap = (uint32_t)(((ky * p.IW + kx) * p.ld1) * 2);
            }
        }
    };
    auto issue_tile = [&](int kt, int buf) {
        uint32_t soff_a;
        uint32_t soff_w = (uint32_t)kt * (BK * 2);
        if constexpr (SPATIAL) {
            if (run_tap != sel_tap) { select_tap(run_tap); sel_tap = run_tap; }
            soff_a = soff_tap + (uint32_t)(run_c0 * 2);
            soff_w = (uint32_t)((run_tap * Cin + run_c0) * 2);
            if (++run_tap == p.taps) { run_tap = 0; run_c0 += BK; }      // K order: see igemm_256_kernel
        } else {
            soff_a = (uint32_t)((p.a_wrap > 0 && kt * BK >= p.a_wrap) ? kt * BK - p.a_wrap : kt * BK) * 2;          // a_wrap: A read twice
        }
        const uint32_t sa = smem_base + buf * STAGE + wave * (8 * MB * 128);
        const uint32_t sw = smem_base + buf * STAGE + A_BYTES + wave * (8 * NB * 128);
#pragma unroll
        for (int j = 0; j < MB; ++j) dma16_buf(SPATIAL ? vsel[j] : voff_a[j], srd_a, soff_a, sa + j * 1024);
#pragma unroll
        for (int j = 0; j < NB; ++j) dma16_buf(voff_w[j], srd_w, soff_w, sw + j * 1024);
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int a_rd = tile_off(wm * (16 * MB) + l15, lq);      // + mb * 2048; ^ 64 for the second k-step
    const int b_rd = tile_off(wn * (16 * NB) + l15, lq);      // + nb * 2048
    const int nkt_all = p.K / BK;
    int kt0 = 0, nkt = nkt_all;
    if (p.splitk > 1) {
        const int per = (nkt_all + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        nkt = min(per, nkt_all - kt0);
        if (nkt < 0) nkt = 0;
    }
    if (SPATIAL) {
        run_c0 = (kt0 / p.taps) * BK;
        run_tap = kt0 - (kt0 / p.taps) * p.taps;
    }
    EDTR_STAMP(1);
    if (nkt > 0) issue_tile(kt0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            issue_tile(kt0 + kt + 1, cur ^ 1);
            // this wave's MB + NB DMAs of tile kt have landed
            static_assert(MB + NB == 9 || MB + NB == 8 || MB + NB == 6, "add the literal wait count for this geometry");
            if constexpr (MB + NB == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if constexpr (MB + NB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt == 0) EDTR_STAMP(2);
        const char* sa = smem + cur * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            U4 af[MB], bf[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) af[mb] = *reinterpret_cast<const U4*>(sa + (a_rd ^ (ks * 64)) + mb * 2048);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const U4*>(sw + (b_rd ^ (ks * 64)) + nb * 2048);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = T::mfma16(af[mb], bf[nb], acc[mb][nb]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (nkt == 0) __syncthreads();
    EDTR_STAMP(3);

    // ---- epilogue: the whole BM x BN fp32 tile through LDS in ONE pass (round 3; 80 KiB for 128 x 160 — two workgroups are exactly
    // the CU's 160 KiB).  Two passes of 64 rows in the main loop's 72 KiB cost a second set of barriers, a second round of
    // bias / residual load latency and wrote with half of the waves at a time: 10.7k cycles of a K = 320 workgroup's 25k in the
    // stamps, against 6.5k for the 128 x 128 tile.
    float* stage = reinterpret_cast<float*>(smem);
    const bool gn_acc = p.gn_partial != nullptr && p.splitk <= 1;
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.0f; gq[j] = 0.0f; }
    if (p.debug_flags & 2) {       // EDTR_IGEMM_N160_TWO_PASS=1: the former two-pass epilogue (rows of the waves wm == h), A/B on one device
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (wm == h) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stage[(mb * 16 + 4 * lq + r) * BN + wn * (16 * NB) + nb * 16 + l15] = acc[mb][nb][r];
            }
            rows_phase<T, 16 * MB, BN, false, kThreads, false, BN, false, 1>(p, stage, m0 + h * (16 * MB), n0, p.N, o_zoff, gn_acc, gs, gq);
            __syncthreads();
        }
    } else {
    auto stage_acc = [&]() {               // rows_phase's pre-publish hook: its operand loads are in flight under this staging
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(wm * (16 * MB) + mb * 16 + 4 * lq + r) * BN + wn * (16 * NB) + nb * 16 + l15] = acc[mb][nb][r];
        EDTR_STAMP_T(12);
    };
    // (FOLD = 2: the producer side of the LayerNorm fold only — the consumer side keeps 2 KiB of static LDS for the row scalars, which
    //  would push two 80 KiB workgroups over the CU's 160 KiB; launches with ln_stats take the 128 x 128 tiles)
    rows_phase<T, BM, BN, false, kThreads, false, BN, 2, 1>(p, stage, m0, n0, p.N, o_zoff, gn_acc, gs, gq, stage_acc);
    __syncthreads();                       // every thread is done reading the staged rows
    EDTR_STAMP_T(15);
    }
    if (gn_acc) {
        // thread (row group r0 = tid / VPR < RG, column group tid % VPR) -> LDS [r0][BN][2], then BN threads fold the row groups
        constexpr int VPR = BN / 8, RG = kThreads / VPR;
        const int n8 = tid % VPR, r0 = tid / VPR;
        if (r0 < RG) {
            float* dst = stage + (r0 * BN + n8 * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[j]; dst[2 * j + 1] = gq[j]; }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N && m0 < p.M) {
            float a = 0.0f, q = 0.0f;
#pragma unroll
            for (int g = 0; g < RG; ++g) { a += stage[(g * BN + tid) * 2]; q += stage[(g * BN + tid) * 2 + 1]; }
            float* dst = p.gn_partial + ((int64_t)(m0 >> 7) * gn_ld_of(p) + n0 + tid) * 2;
            dst[0] = a;
            dst[1] = q;
        }
    }
    EDTR_STAMP(4); EDTR_STAMP(7);
}

template <typename T, bool SPATIAL, int MB, int NB>
int launch_n160(const edtr_igemm_params& p, hipStream_t stream) {
    constexpr int BM = 32 * MB, BN = 32 * NB;
    constexpr int loop_lds = 2 * (BM + BN) * BK * 2, stage_lds = BM * BN * 4;     // 72 KiB main loop / 80 KiB fp32 staging for 128 x 160
    constexpr int lds = loop_lds > stage_lds ? loop_lds : stage_lds;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_n160_kernel<T, SPATIAL, MB, NB>), lds, attr_set)) return rc_;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, p.splitk > 1 ? p.splitk : 1, p.Z);
    hipLaunchKernelGGL((igemm_n160_kernel<T, SPATIAL, MB, NB>), grid, dim3(kThreads), lds, stream, p);
    EDTR_LAUNCH_CHECK();
    if (p.splitk > 1) return launch_splitk_reducer<T>(p, stream);
    return EDTR_OK;
}

template <typename T, int MI, int NI, bool SPATIAL>
int launch(const edtr_igemm_params& p, hipStream_t stream) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    static_assert(lds >= BM * BN * 4, "epilogue staging must fit the main-loop LDS");
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_kernel<T, MI, NI, SPATIAL>), lds, attr_set)) return rc_;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, p.splitk > 1 ? p.splitk : 1, p.Z);
    hipLaunchKernelGGL((igemm_kernel<T, MI, NI, SPATIAL>), grid, dim3(kThreads), lds, stream, p);
    EDTR_LAUNCH_CHECK();
    if (p.splitk > 1) return launch_splitk_reducer<T>(p, stream);
    return EDTR_OK;
}

// buffer-addressed fast path: 32-bit byte offsets must cover the A and W operands (per z slice)
// ------------------------------------------------------------------------------------------------------
// Halo tile (tile = 16): 3x3 / stride 1 / pad 1 convolutions whose 128x128 loop is bound by the L1 -> LDS path (64 B/clk/CU
// at MFMA peak: every input element crosses it nine times, once per tap).  A workgroup owns a 16 x 16 patch of output
// pixels x 128 output channels and keeps the 18 x 18 INPUT patch of the current 64-channel chunk in LDS: it is staged
// once per chunk (40.5 KiB) and the nine taps read it at shifted addresses, so a chunk moves 40.5 + 9 x 16 KiB (weights)
// instead of 9 x 48 KiB -> 20 B/clk/CU at MFMA peak.
//   * LDS: two patch buffers (chunk c+1 lands while chunk c is multiplied) + a ring of three 16 KiB weight slices
//     (tap t+2 lands while tap t is multiplied; 9 % 3 == 0 keeps the ring position a compile-time function of the tap).
//     Patch image: pixel (py, px) at (py * 18 + px) * 128 B, 16-byte chunk c at slot c ^ (px & 7): the key depends on
//     the COLUMN only and a patch row is 2304 B = 9 bank rows, so a tap's dy and the MFMA block's y are immediate offsets
//     and only the three dx need their own address.  px & 7 keeps the ds_read_b128 of 16 consecutive px conflict-free for
//     every dx (the tile rows' (r >> 1) & 7 is 2-way conflicted when the run starts at an odd / offset column).
//   * 8 waves: wave (g, wr, wc) owns pixels y in [8 wr, 8 wr + 8) x all x (128 rows = 8 blocks of one pixel row), output
//     channels [64 wc, +64) and the K HALF g of every 64-channel chunk (32 channels = one v_mfma_f32_16x16x32 step):
//     128 x 64 per wave keeps the LDS reads at 12 fragments per 32 MFMAs; the two K halves are summed in the epilogue's
//     fp32 staging.  Waves w and w+4 (g = 0 / 1) share a SIMD and run half a phase apart (ping-pong as in tile 6):
//     a phase = 16 MFMAs (4 pixel rows x 4 column blocks); while one wave multiplies, its partner reads fragments and
//     issues the DMA of a future slice.
//   * per phase and wave: one weight piece (tap t+2), in phases 2..7 of a chunk also one piece of the next patch;
//     odd phases wait vmcnt(issued in this and the previous phase) = "everything up to two phases ago has landed".
// Output rows are patch pixels (rows_phase<PATCH16>); fused GroupNorm partials go to slot 2 * patch (slot 2 * patch + 1 = 0).
// Split-K (blockIdx.y) cuts the chunk range; the partial slabs go through the shared reducer.  UP2 = the nearest-2x upsample fused
// into the gather (10 x 10 source patch, see the constants below).
// Requires taps == 9, stride 1, pad 1, no concat, C1 % 64 == 0, OH % 16 == OW % 16 == 0, buffer addressing.
// ------------------------------------------------------------------------------------------------------
template <int V> using IC = std::integral_constant<int, V>;

// GEO: 0 = 16 x 16 output patch of one image, 1 = the same with the nearest-2x upsample fused into the gather (UP2),
//      2 = FOUR WHOLE 8 x 8 IMAGES per workgroup (IMG8; round 3): the 3x3 convolutions of the 8x8 latent level (M = 64 B rows
//          against 1280 x 11520 .. 23040 weights).  The implicit-GEMM tiles run them at 380 - 470 TFLOP/s: each of 240 workgroups
//          re-stages its activations nine times and crawls through 30 - 60 K-tiles alone on its CU.  Here the unit is 256 rows =
//          4 images with their zero halo, a [4][10][10]-pixel patch per 64-channel chunk (50 KiB, staged once per chunk), the
//          same nine-tap / ping-pong loop, split-K over chunks.  Patch pixel (i, py, px) at (100 i + 10 py + px) * 128 B, chunk
//          slot c ^ ((px ^ (py & 1)) & 7): an MFMA row block is TWO image rows of 8 pixels, and the row parity in the key keeps
//          the 16 lanes of a ds_read_b128 group on 16 distinct 16-byte bank slots (10 pixels per patch row is even, so the
//          column key alone would put (y, x) and (y + 1, x) on the same slot).
template <typename T, int GEO>
__global__ void __launch_bounds__(512, 1) igemm_halo_kernel(const edtr_igemm_params p) {
    constexpr bool UP2 = GEO == 1, IMG8 = GEO == 2, SUBPIX = GEO == 3;
    // SUBPIX (GEO 3, round 4): the nearest-2x upsample convolution in its SUB-PIXEL form.  The 2 x 2 blocks of the upsampled image are
    // constant, so output pixel (2 s + py, 2 r + px) is a 2 x 2 convolution of the SOURCE image around (s, r) whose four weights are
    // sums of the 3 x 3 kernel's (row py = 0: {w0 | w1 + w2} at source rows s - 1, s; py = 1: {w0 + w1 | w2} at rows s, s + 1; columns
    // likewise): 4 taps instead of 9 per output pixel, 2.25 x fewer MACs than the UP2 gather, which multiplies the same source
    // element by up to four taps separately.  The host packs the four phase matrices [2 py + px][N][2][2][C1] (sums in fp32, then the
    // 16-bit / multi-part rounding) at p.w + phase * p.w_phase_stride.  A workgroup owns ONE phase of a 16 x 16 SOURCE block (256 output
    // pixels of a 32 x 32 output block, stride 2) x 128 channels: the patch is the plain 18 x 18 geometry at source origin
    // (16 ty + py - 1, 16 tx + px - 1), tap (dy, dx) reads patch pixel (y + dy, x + dx); the four phases of a block are consecutive
    // units (their patches overlap in L2).  4 taps do not divide the ring of three weight slices, so the ring position is carried
    // across chunks (rb) instead of being a compile-time function of the tap.
    constexpr int NT = SUBPIX ? 4 : 9, TW = SUBPIX ? 2 : 3;         // taps per chunk, taps per kernel row
    EDTR_STAMP(0); EDTR_STAMP(6); EDTR_STAMP(5);
    // UP2 (nearest-2x upsample fused into the gather, `Upsample` of the UNet / VAE decoder): output pixel (y, x), tap (ky, kx) reads
    // SOURCE pixel ((oy0 + y + ky - 1) >> 1, (ox0 + x + kx - 1) >> 1): the patch is 10 x 10 source pixels (12.5 KiB per chunk),
    // patch row of block row y and tap ky = (y + ky + 1) >> 1, patch column of lane x and tap kx = (x + kx + 1) >> 1.
    constexpr int PW = (UP2 || IMG8) ? 10 : 18, PROW = PW * 128;
    // patch pieces (1 KiB = 8 pixels) per wave and chunk.  SUBPIX reads patch rows 0..16 only (306 pixels): 5 pieces per wave (320
    // pixels) cover them, and they are issued in phases 1..5 of the chunk's 8 — the last one must be >= 2 phases before the chunk
    // ends, because the counted waits only retire what was issued up to two phases ago and the next chunk reads the patch at once
    // (with 9 taps the pieces of phases 2..7 have ten more phases to land)
    constexpr int NPP = UP2 ? 2 : (IMG8 ? 7 : (SUBPIX ? 5 : 6));
    constexpr int PP0 = SUBPIX ? 1 : 2;                     // first phase of a chunk that stages a piece of the next patch
    // 324 (100; IMG8: 400) pixels x 128 B, filled by 48 (16; 50) one-KiB pieces, the tail lands in padding (IMG8: pieces 50..55 land
    // in a scratch KiB behind the weight ring, so that every wave issues the same number of DMAs and the vmcnt counts hold)
    constexpr int PATCHB = UP2 ? 16 * 1024 : (IMG8 ? 50 * 1024 : 48 * 1024);
    constexpr int BTAP = 128 * BK * 2;         // 16 KiB
    constexpr int B_BASE = 2 * PATCHB;
    constexpr int SCRATCH = B_BASE + 3 * BTAP; // IMG8 only
    // GEO 0, a_gn != NULL (round 4): GroupNorm apply + SiLU of the INPUT fused into the patch staging.  The producing launch leaves
    // the raw tensor in memory; edtr_gn_table turns its statistics into (scale, shift) per image and channel; here every wave
    // normalises the patch pieces IT staged, in place in LDS, once they have landed and before the chunk that multiplies them
    // starts: y = silu(x * scale[c] + shift[c]) exactly as edtr_gn_apply computes it, pixels outside the image keep the zeros the
    // padding needs.  One piece (64 lanes x 8 elements) per phase in phases 6..11 of the previous chunk, behind that phase's MFMA
    // issue (the fragment registers are dead there and the matrix pipe runs under the arithmetic); the 64 x 2 table entries of
    // the next chunk arrive by one extra DMA per wave in phase 0 (every wave fetches the same 512 bytes: equal DMA counts keep
    // the counted waits simple) into one of two 1-KiB slots behind the weight ring.  The normalised tensor never exists in
    // memory: one read + one write of the activation per GroupNorm less.  Every 128-column tile of a patch repeats the arithmetic,
    // so the callers use it where N == 128 (ops.gn_in_conv_ok: measured +1.4 % on the headline there, a loss at N >= 256).
    constexpr int TBL = B_BASE + 3 * BTAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    // patches per row / per image (SUBPIX: 16 x 16 source blocks per row, units = blocks x 4 phases per image)
    const int tw = IMG8 ? 1 : (SUBPIX ? p.IW >> 4 : p.OW >> 4), tpi = IMG8 ? 1 : (SUBPIX ? tw * (p.IH >> 4) * 4 : tw * (p.OH >> 4));
    const int nbm = IMG8 ? p.M >> 8 : (p.M / (p.OH * p.OW)) * tpi, nbn = (p.N + 127) / 128;
    int bid = blockIdx.x;
    {
        const int nblk = nbm * nbn, q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn;
    tile_coords(p, bid, nbm, nbn, tm, tn);
    const int img = IMG8 ? tm * 4 : tm / tpi, trp = IMG8 ? 0 : tm - img * tpi;     // IMG8: first of the unit's 4 images
    const int phase_id = SUBPIX ? trp & 3 : 0, phy = phase_id >> 1, phx = phase_id & 1, tr = SUBPIX ? trp >> 2 : trp;
    const int ty = tr / tw, tx = tr - ty * tw;
    const int oy0 = ty * 16, ox0 = tx * 16, n0 = tn * 128;          // (SUBPIX: source block origin)
    // source pixel of patch position (0, 0)
    const int sy0 = UP2 ? (oy0 >> 1) - 1 : oy0 - 1 + phy, sx0 = UP2 ? (ox0 >> 1) - 1 : ox0 - 1 + phx;
    // first pixel of the patch (SUBPIX: the phase's first output pixel, the others lie at even offsets from it)
    const int m0 = IMG8 ? tm * 256 : SUBPIX ? (img * p.OH + 2 * oy0 + phy) * p.OW + 2 * ox0 + phx : (img * p.OH + oy0) * p.OW + ox0;

    const uint16_t* a1 = static_cast<const uint16_t*>(p.a1);
    const uint16_t* wp = static_cast<const uint16_t*>(p.w) + (SUBPIX ? (int64_t)phase_id * p.w_phase_stride : 0);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const u32x4 srd_a = make_srd(a1);
    const u32x4 srd_w = make_srd(wp);
    const int Cin = p.C1;
    const int nvalid = p.n_valid > 0 ? p.n_valid : p.N;
    // split-K (blockIdx.y): this workgroup owns the 64-channel chunks [cbeg, cend) and writes an fp32 partial slab
    const int nsplit = p.splitk > 1 ? p.splitk : 1, nchunk_all = Cin / BK;
    const int cbeg = (int)blockIdx.y * nchunk_all / nsplit, cend = ((int)blockIdx.y + 1) * nchunk_all / nsplit;

    // ---- staging geometry.  Patch piece q = wave + 8 j: LDS bytes [q KiB, +1 KiB) = pixels 8 q .. 8 q + 7, lane -> (pixel, slot)
    uint32_t voff_p[NPP], voff_w[2];
    uint32_t gn_bits = 0;                      // per patch piece j: bits 4 j .. 4 j + 2 = the 8-channel group of this lane's 16 bytes, bit 4 j + 3 = inside the image
    uint32_t lds_p[NPP];                       // LDS byte offset of patch piece j inside a patch buffer (IMG8: or the scratch KiB)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 8 + (lane >> 3), slot = lane & 7, n = n0 + r;
        const int c = slot ^ ((r >> 1) & 7);
        voff_w[j] = n < nvalid ? (uint32_t)(((int64_t)n * p.ldw + c * 8) * 2) : kOobOffset;
    }
    // the weight slices of taps 0 and 1 start their flight before the patch addresses are worked out
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dma16_buf(voff_w[j], srd_w, (uint32_t)((t * Cin + cbeg * BK) * 2), smem_base + B_BASE + t * BTAP + (wave + 8 * j) * 1024);
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int u = (wave + 8 * j) * 64 + lane, pp = u >> 3, slot = u & 7;
        if constexpr (IMG8) {
            const int i = pp / 100, rem = pp - i * 100, py = rem / 10, px = rem - py * 10;
            const int iy = py - 1, ix = px - 1;
            const bool ok = pp < 400 && iy >= 0 && iy < 8 && ix >= 0 && ix < 8;
            const int c = slot ^ ((px ^ (py & 1)) & 7);
            voff_p[j] = ok ? (uint32_t)(((((int64_t)(img + i) * 8 + iy) * 8 + ix) * p.ld1 + c * 8) * 2) : kOobOffset;
            lds_p[j] = wave + 8 * j < 50 ? (uint32_t)((wave + 8 * j) * 1024) : 0xFFFFFFFFu;
        } else {
            const int py = pp / PW, px = pp - py * PW;
            const int iy = sy0 + py, ix = sx0 + px;
            const bool ok = pp < PW * PW && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const int c = slot ^ (px & 7);
            voff_p[j] = ok ? (uint32_t)(((((int64_t)img * p.IH + iy) * p.IW + ix) * p.ld1 + c * 8) * 2) : kOobOffset;
            lds_p[j] = (uint32_t)((wave + 8 * j) * 1024);
            if (ok) gn_bits |= (uint32_t)(8 | c) << (4 * j);
        }
    }
    auto stage_w = [&](int chunk, int tap, int j, int buf) {
        const uint32_t vo = chunk < cend ? voff_w[j] : kOobOffset;
        dma16_buf(vo, srd_w, (uint32_t)((tap * Cin + chunk * BK) * 2), smem_base + B_BASE + buf * BTAP + (wave + 8 * j) * 1024);
    };
    auto stage_p = [&](int chunk, int j, int par) {
        const uint32_t vo = chunk < cend ? voff_p[j] : kOobOffset;
        const uint32_t dst = (IMG8 && lds_p[j] == 0xFFFFFFFFu) ? (uint32_t)SCRATCH : (uint32_t)(par * PATCHB) + lds_p[j];
        dma16_buf(vo, srd_a, (uint32_t)(chunk * BK * 2), smem_base + dst);
    };
    const bool gnf = GEO == 0 && p.a_gn != nullptr;
    const u32x4 srd_t = make_srd(gnf ? p.a_gn + (int64_t)img * Cin * 2 : reinterpret_cast<const float*>(a1));
    auto stage_t = [&](int chunk, int par) {                 // (scale, shift) of the chunk's 64 channels: 512 bytes, lanes 0..31
        const uint32_t vo = (chunk < cend && lane < 32) ? (uint32_t)(lane * 16) : kOobOffset;
        dma16_buf(vo, srd_t, (uint32_t)(chunk * BK * 8), smem_base + TBL + par * 1024);
    };
    auto gn_piece = [&](int j, int par) {
        const uint32_t bits = gn_bits >> (4 * j);
        if (bits & 8) {
            char* q = smem + par * PATCHB + (wave + 8 * j) * 1024 + lane * 16;
            const float* tb = reinterpret_cast<const float*>(smem + TBL + par * 1024) + (bits & 7) * 16;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb), t1 = *reinterpret_cast<const f32x4*>(tb + 4);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(tb + 8), t3 = *reinterpret_cast<const f32x4*>(tb + 12);
            float f[8];
            unpack8<T>(*reinterpret_cast<const U4*>(q), f);
            f[0] = f[0] * t0[0] + t0[1]; f[1] = f[1] * t0[2] + t0[3];
            f[2] = f[2] * t1[0] + t1[1]; f[3] = f[3] * t1[2] + t1[3];
            f[4] = f[4] * t2[0] + t2[1]; f[5] = f[5] * t2[2] + t2[3];
            f[6] = f[6] * t3[0] + t3[1]; f[7] = f[7] * t3[2] + t3[3];
            if (p.a_gn_silu) {
                // silu_f for the eight values in lockstep (common.h, gelu_erf_lockstep: left alone, hipcc runs each value's exp -> add ->
                // rcp -> mul chain to its end before the next, at the VALU latency)
                float e[8];
                pin8(f);
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
            }
            *reinterpret_cast<U4*>(q) = pack8<T>(f);
        }
    };

    // ---- fragment read geometry: A block mb of tap (ky, kx) = patch row 8 wr + mb + ky, pixels l15 + kx, K chunk 4 g + lq
    // IMG8: block mb = image 2 wr + mb / 4, image rows 2 (mb % 4) + (l15 >> 3), pixel l15 & 7; the key needs the row parity, so
    // there is one address per (kx, ky & 1): a_rd[kx + 3 (ky & 1)]
    int a_rd[IMG8 ? 6 : 3];
#pragma unroll
    for (int kx = 0; kx < TW; ++kx) {
        if constexpr (IMG8) {
            const int yy = l15 >> 3, px = (l15 & 7) + kx;
#pragma unroll
            for (int kyp = 0; kyp < 2; ++kyp)
                a_rd[kx + 3 * kyp] = (wr * 200 + yy * 10 + px) * 128 + (((g * 4 + lq) ^ ((px ^ ((yy + kyp) & 1)) & 7)) << 4);
        } else {
            const int px = UP2 ? (l15 + kx + 1) >> 1 : l15 + kx;
            a_rd[kx] = (wr * (UP2 ? 4 : 8) * PW + px) * 128 + (((g * 4 + lq) ^ (px & 7)) << 4);
        }
    }
    const int b_rd = tile_off(wc * 64 + l15, g * 4 + lq);            // + nb * 2048

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    U4 afr[4], bfr[4];

    // ---- prologue: patch of chunk 0, weight slices of taps 0 and 1
    EDTR_STAMP(1);
    if constexpr (GEO == 0) {
        if (gnf) stage_t(cbeg, 0);
    }
#pragma unroll
    for (int j = 0; j < NPP; ++j) stage_p(cbeg, j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (GEO == 0) {
        if (gnf) {           // the first chunk's patch: every wave normalises its own pieces (its own table copy has landed too)
#pragma unroll
            for (int j = 0; j < NPP; ++j) gn_piece(j, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run half a phase behind their SIMD partners
    asm volatile("" ::: "memory");
    EDTR_STAMP(2);

    int rb = 0;                                        // SUBPIX: ring slot of this chunk's tap 0 (4 taps per chunk over a ring of 3)
    for (int c = cbeg; c < cend; ++c) {
        const int par = (c - cbeg) & 1;
        const char* pa = smem + par * PATCHB;
        const int rbuf[3] = {rb, rb == 2 ? 0 : rb + 1, rb == 0 ? 2 : rb - 1};        // (rb + k) % 3
        auto phase = [&](auto TAPc, auto SUBc) {
            constexpr int TAP = decltype(TAPc)::value, SUB = decltype(SUBc)::value, KY = TAP / TW, KX = TAP % TW;
            constexpr int TAP2 = (TAP + 2) % NT, PH = 2 * TAP + SUB;      // PH: phase inside the chunk, 0..2 NT - 1
            // ring slots of tap TAP and of the slice staged now (tap TAP + 2): compile-time where 9 % 3 == 0 allows it
            const int BUF = SUBPIX ? rbuf[TAP % 3] : TAP % 3, BUF2 = SUBPIX ? rbuf[(TAP + 2) % 3] : TAP2 % 3;
            const int c2 = TAP + 2 >= NT ? c + 1 : c;
            if constexpr (SUB == 0) {
                const char* pb = smem + B_BASE + BUF * BTAP + b_rd;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const U4*>(pb + nb * 2048);
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                if constexpr (IMG8)       // block SUB * 4 + mb: image SUB of this wave's two, image rows 2 mb + {0, 1}
                    afr[mb] = *reinterpret_cast<const U4*>(pa + a_rd[KX + 3 * (KY & 1)] + (SUB * 100 + (2 * mb + KY) * 10) * 128);
                else
                    afr[mb] = *reinterpret_cast<const U4*>(pa + a_rd[KX] + (UP2 ? (SUB * 4 + mb + KY + 1) >> 1 : SUB * 4 + mb + KY) * PROW);
            }
            stage_w(c2, TAP2, SUB, BUF2);
            if constexpr (PH >= PP0 && PH < PP0 + NPP) stage_p(c + 1, PH - PP0, par ^ 1);
            if constexpr (GEO == 0 && PH == 0) {
                if (gnf) stage_t(c + 1, par ^ 1);
            }
            if constexpr (SUB == 1) {
                // issued in this and the previous phase: 1 weight piece each, + 1 patch piece each in phases PP0 .. PP0 + NPP - 1
                // (+ the table piece of phase 0 when the GroupNorm is fused)
                constexpr int INFLIGHT = 2 + (PH >= PP0 && PH < PP0 + NPP ? 1 : 0) + (PH - 1 >= PP0 && PH - 1 < PP0 + NPP ? 1 : 0);
                if (GEO == 0 && PH == 1 && gnf) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");       // (GEO 0: INFLIGHT is 2 at phase 1)
                else if constexpr (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (INFLIGHT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[SUB * 4 + mb][nb] = T::mfma16(bfr[nb], afr[mb], acc[SUB * 4 + mb][nb]);    // transposed: D[channel][pixel]
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (GEO == 0 && PH >= 6 && PH < 6 + NPP) {
                // the piece this wave staged in phase PH - 4 landed before the wait of phase PH - 1 at the latest
                if (gnf && c + 1 < cend) gn_piece(PH - 6, par ^ 1);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto tap_body = [&](auto TAPc) { phase(TAPc, IC<0>{}); phase(TAPc, IC<1>{}); };
        tap_body(IC<0>{}); tap_body(IC<1>{}); tap_body(IC<2>{}); tap_body(IC<3>{});
        if constexpr (!SUBPIX) {
            tap_body(IC<4>{}); tap_body(IC<5>{}); tap_body(IC<6>{}); tap_body(IC<7>{}); tap_body(IC<8>{});
        } else {
            rb = rbuf[1];                                // 4 % 3 == 1
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave groups
    __syncthreads();
    EDTR_STAMP(3);

    // ---- epilogue: the two K halves meet in the fp32 staging tile [256 pixels][128 channels], 528-byte rows (129 KiB).  The product
    // is transposed (weights are the MFMA's row operand), so a lane holds 4 CONSECUTIVE channels of pixel l15 per block: 16-byte
    // LDS accesses (32 per lane and K half instead of 128 dword ones; the 528-byte pitch keeps the 8-lane write groups conflict-free)
    constexpr int SPITCH = 132;                          // floats per staged row
    float* stage = reinterpret_cast<float*>(smem);
    auto sptr = [&](int mb, int nb) { return stage + (wr * 128 + mb * 16 + l15) * SPITCH + wc * 64 + nb * 16 + 4 * lq; };
    if (g == 0) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) *reinterpret_cast<f32x4*>(sptr(mb, nb)) = acc[mb][nb];
    }
    __syncthreads();
#ifdef EDTR_STAMPS
#undef STAMP_VALUE
#define STAMP_VALUE __builtin_amdgcn_s_memtime()
    EDTR_STAMP(8);
#endif
    // the second K half is added inside rows_phase, between its operand prefetch (bias, time-embedding row, residual vectors)
    // and the barrier that publishes the tile: the loads fly under the 2.6k cycles of this pass
    auto add_second_half = [&]() {
        if (g == 1) {
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {         // (ds_add_f32 here: 10x slower, measured)
                    f32x4* q = reinterpret_cast<f32x4*>(sptr(mb, nb));
                    *q = *q + acc[mb][nb];
                }
        }
    };
    const bool gn_acc = p.gn_partial != nullptr && p.splitk <= 1;      // (split-K: the reducer writes the statistics)
    float gs[8], gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.0f; gq[j] = 0.0f; }
    // (PF = 0: the staged vectors are read inside the row iteration.  Round 4 measured PF = 1 — one iteration ahead, no spill at 246 - 250
    //  VGPRs — in a same-device A/B of two builds: 108.69 / 108.42 vs 108.62 / 108.24 images/s, the K = 1152 convolutions 701 / 706 vs
    //  699 / 695 us: no change, this epilogue is not bound by its LDS round trips; PF = 2 spills 107 - 128 registers)
    rows_phase<T, 256, 128, false, 512, SUBPIX ? 3 : !IMG8, SPITCH, false, 0>(p, stage, m0, n0, p.N, 0, gn_acc, gs, gq, add_second_half);   // IMG8: rows are consecutive
    if (gn_acc) {
        // thread (row group tid / 16, column group tid % 16): lanes l, l+16, l+32, l+48 share a column group; fold, then the 8 waves
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gs[j] += __shfl_xor(gs[j], 16, 64); gs[j] += __shfl_xor(gs[j], 32, 64);
            gq[j] += __shfl_xor(gq[j], 16, 64); gq[j] += __shfl_xor(gq[j], 32, 64);
        }
        __syncthreads();                       // every thread is done reading the staged tile
        if (lane < 16) {
            float* dst = stage + (wave * 128 + lane * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dst[2 * j] = gs[j]; dst[2 * j + 1] = gq[j]; }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float a = 0.0f, q = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { a += stage[(w * 128 + tid) * 2]; q += stage[(w * 128 + tid) * 2 + 1]; }
            float* dst = p.gn_partial + ((int64_t)(2 * tm) * gn_ld_of(p) + n0 + tid) * 2;     // two 128-row slots per 256-pixel patch
            dst[0] = a;
            dst[1] = q;
            dst[2 * gn_ld_of(p)] = 0.0f;
            dst[2 * gn_ld_of(p) + 1] = 0.0f;
        }
    }
    EDTR_STAMP(4); EDTR_STAMP(7);
}

template <typename T, int GEO>
int launch_halo(const edtr_igemm_params& p, hipStream_t stream) {
    constexpr bool UP2 = GEO == 1;
    // 144 KiB (also SUBPIX); UP2: main loop 80 KiB, the epilogue's staging tile 132 KiB; IMG8: 2 x 50 KiB patches + 48 KiB weights + 1 KiB scratch
    constexpr int lds = UP2 ? 256 * 132 * 4 : (GEO == 2 ? 2 * 50 * 1024 + 3 * 128 * BK * 2 + 1024 : 2 * 48 * 1024 + 3 * 128 * BK * 2 + (GEO == 0 ? 2048 : 0));
    static EdtrLdsOnce attr_set;
    if (int rc_ = edtr_lds_attr(reinterpret_cast<const void*>(&igemm_halo_kernel<T, GEO>), lds, attr_set)) return rc_;
    const int nbm = GEO == 2 ? p.M >> 8 : (p.M / (p.OH * p.OW)) * (p.OH >> 4) * (p.OW >> 4), nbn = (p.N + 127) / 128;
    hipLaunchKernelGGL((igemm_halo_kernel<T, GEO>), dim3(nbm * nbn, p.splitk > 1 ? p.splitk : 1, 1), dim3(512), lds, stream, p);
    EDTR_LAUNCH_CHECK();
    if (p.splitk > 1) return launch_splitk_reducer<T>(p, stream);
    return EDTR_OK;
}

// the halo tile's shape requirements (the caller checks buffer addressability)
// four whole 8 x 8 images per workgroup (GEO 2 of the halo kernel)
static bool igemm_halo_img8(const edtr_igemm_params& p) {
    return !p.upsample2x && p.OH == 8 && p.OW == 8 && p.IH == 8 && p.IW == 8 && (p.M & 255) == 0;
}

static bool igemm_halo_ok(const edtr_igemm_params& p, bool spatial) {
    const int up = p.upsample2x ? 2 : 1;
    if (p.upsample2x == 2 && ((p.IH & 15) || (p.IW & 15) || p.w_phase_stride < (int64_t)p.N * p.ldw || p.ldw < 4 * p.C1)) return false;   // sub-pixel form: 16 x 16 source blocks
    return spatial && p.taps == 9 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.C2 == 0 && (p.C1 & 63) == 0 &&
           p.OH == p.IH * up && p.OW == p.IW * up && (((p.OH & 15) == 0 && (p.OW & 15) == 0) || igemm_halo_img8(p)) && p.Z == 1 &&
           p.splitk <= p.C1 / 64 && p.act != EDTR_ACT_GEGLU && p.M == (p.M / (p.OH * p.OW)) * p.OH * p.OW;
}

static bool igemm_fast_addressable(const edtr_igemm_params& p, bool spatial) {
    const int64_t a_rows = spatial ? (int64_t)(p.M / (p.OH * p.OW)) * p.IH * p.IW : p.M;
    const int64_t a_bytes = (a_rows + (spatial ? 3 * (int64_t)p.IW + 3 : 0)) * p.ld1 * 2 + (int64_t)p.K * 2;
    const int64_t w_bytes = (int64_t)p.N * p.ldw * 2 + (int64_t)p.K * 2;
    return a_bytes < 0xF0000000LL && w_bytes < 0xF0000000LL;
}

// A/B switches for measurements on one device (edtr_hip.h: debug_flags)
static int dbg_flags() {
    static int dbg = -1;
    if (dbg < 0) {
        const char* e8 = getenv("EDTR_IGEMM_GENERAL_EPILOGUE");
        const char* e10 = getenv("EDTR_IGEMM_N160_TWO_PASS");
        const char* e12 = getenv("EDTR_IGEMM_GEGLU_SERIAL");
        dbg = ((e8 && e8[0] == '1') ? 1 : 0) | ((e10 && e10[0] == '1') ? 2 : 0) | ((e12 && e12[0] == '1') ? 4 : 0);
    }
    return dbg;
}

// dry: edtr_igemm_plan — every check of the chosen kernel, no launch; the answer is the tile number
template <typename T>
int dispatch(const edtr_igemm_params& p, int tile, bool spatial, hipStream_t s, bool dry) {
    if (tile == 20) {      // halo tile, 16 x 16 pixels x 160 channels (halo512.hip)
        if (!spatial || !edtr_halo160_ok(p) || !igemm_fast_addressable(p, spatial)) return EDTR_E_UNSUPPORTED;
        if (dry) return tile;
        return edtr_launch_halo160(p, s);
    }
    if (tile == 21) {      // tile 17 as a persistent kernel with deferred stores (halo512.hip)
        if (!spatial || !edtr_halo512p_ok(p) || !igemm_fast_addressable(p, spatial)) return EDTR_E_UNSUPPORTED;
        if (dry) return tile;
        return edtr_launch_halo512p(p, s);
    }
    if (tile == 17) {      // halo tile, 512-pixel units (halo512.hip)
        if (!spatial || !edtr_halo512_ok(p) || !igemm_fast_addressable(p, spatial)) return EDTR_E_UNSUPPORTED;
        if (dry) return tile;
        return edtr_launch_halo512(p, s);
    }
    if (tile == 16) {      // halo tile for 3x3 / stride 1 convolutions
        if (!igemm_halo_ok(p, spatial) || !igemm_fast_addressable(p, spatial)) return EDTR_E_UNSUPPORTED;
        if (dry) return tile;
        if (igemm_halo_img8(p)) return launch_halo<T, 2>(p, s);
        if (p.upsample2x == 2) return launch_halo<T, 3>(p, s);
        return p.upsample2x ? launch_halo<T, 1>(p, s) : launch_halo<T, 0>(p, s);
    }
    if (tile >= 3 && tile <= 14) {
        // buffer-addressed fast path: 32-bit byte offsets must cover the A and W operands (per z slice)
        const bool fast = (!p.upsample2x || (p.stride == 1 && (tile == 3 || tile >= 6))) && igemm_fast_addressable(p, spatial);
        if (tile == 14) {  // 256 x 32 tile for skinny-N convolutions (N <= 32: the decoder's 3-channel output conv wastes 94 % of a
                           // 128-wide tile); an instantiation of the 16x16x32 template, experiment, same status
            if (!fast || p.act == EDTR_ACT_GEGLU || p.gn_partial) return EDTR_E_UNSUPPORTED;
            if (dry) return tile;
            return spatial ? launch_n160<T, true, 8, 1>(p, s) : launch_n160<T, false, 8, 1>(p, s);
        }
        if (tile == 8) {
            if (!fast || p.act == EDTR_ACT_GEGLU) return EDTR_E_UNSUPPORTED;
            if (dry) return tile;
            return spatial ? launch_n160<T, true, 4, 5>(p, s) : launch_n160<T, false, 4, 5>(p, s);
        }
        if (tile == 6) {
            if (!fast || p.splitk > 1 || p.act == EDTR_ACT_GEGLU) return EDTR_E_UNSUPPORTED;
            if (dry) return tile;
            return spatial ? launch_256<T, true>(p, s) : launch_256<T, false>(p, s);
        }
        if (dry) return tile;
        if (!spatial) return fast ? launch_dma<T, false, true>(p, s) : launch_dma<T, false, false>(p, s);
        return fast ? launch_dma<T, true, true>(p, s) : launch_dma<T, true, false>(p, s);
    }
    if (dry) return tile;
    if (tile == 1) return spatial ? launch<T, 2, 2, true>(p, s) : launch<T, 2, 2, false>(p, s);
    return spatial ? launch<T, 1, 1, true>(p, s) : launch<T, 1, 1, false>(p, s);
}

}  // namespace

namespace {
// Second stage of the fused GroupNorm statistics: partial[(image, row tile)][C][2] fp32 -> sums[image][group][2] fp64.
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* partial, int tiles_per_image, int C, int groups,
                                                         double* sums) {
    __shared__ double red_s[256], red_q[256];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups;
    const int total = tiles_per_image * cpg;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int t = i / cpg, c = g * cpg + (i - t * cpg);
        const float* src = partial + (((int64_t)b * tiles_per_image + t) * C + c) * 2;
        s += (double)src[0];
        q += (double)src[1];
    }
    red_s[threadIdx.x] = s;
    red_q[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red_s[threadIdx.x] += red_s[threadIdx.x + o]; red_q[threadIdx.x] += red_q[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[((int64_t)b * groups + g) * 2] = red_s[0];
        sums[((int64_t)b * groups + g) * 2 + 1] = red_q[0];
    }
}
}  // namespace

extern "C" int edtr_gn_finalize(const float* partial, int tiles_per_image, int B, int C, int groups, double* sums,
                                edtr_stream_t stream) {
    if (!partial || !sums) return EDTR_E_NULL;
    if (tiles_per_image <= 0 || B <= 0 || C <= 0 || groups <= 0 || C % groups) return EDTR_E_SHAPE;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, static_cast<hipStream_t>(stream), partial,
                       tiles_per_image, C, groups, sums);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

namespace {
// (scale, shift) per image and channel of a GroupNorm whose statistics exist as tile partials of the producing edtr_igemm or as
// the fp64 sums of edtr_gn_stats: table[b][c] = (gamma[c] rstd, beta[c] - mean gamma[c] rstd), the two numbers edtr_gn_apply
// forms per channel — for consumers that normalise the tensor themselves (edtr_igemm's a_gn).
__global__ void __launch_bounds__(256) gn_table_kernel(const float* partial, int tiles_per_image, const double* sums, int C, int groups,
                                                      int HW, const float* gamma, const float* beta, float eps, float* table) {
    __shared__ double red_s[256], red_q[256];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups;
    double s = 0.0, q = 0.0;
    if (partial) {
        const int total = tiles_per_image * cpg;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int t = i / cpg, c = g * cpg + (i - t * cpg);
            const float* src = partial + (((int64_t)b * tiles_per_image + t) * C + c) * 2;
            s += (double)src[0];
            q += (double)src[1];
        }
        red_s[threadIdx.x] = s;
        red_q[threadIdx.x] = q;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) { red_s[threadIdx.x] += red_s[threadIdx.x + o]; red_q[threadIdx.x] += red_q[threadIdx.x + o]; }
            __syncthreads();
        }
        s = red_s[0];
        q = red_q[0];
    } else {
        s = sums[((int64_t)b * groups + g) * 2];
        q = sums[((int64_t)b * groups + g) * 2 + 1];
    }
    const double inv_cnt = 1.0 / ((double)HW * cpg);
    const double mean = s * inv_cnt;
    double var = q * inv_cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = __builtin_amdgcn_rsqf((float)var + eps);
    for (int i = threadIdx.x; i < cpg; i += 256) {
        const int c = g * cpg + i;
        const float sc = gamma[c] * rstd;
        float* dst = table + ((int64_t)b * C + c) * 2;
        dst[0] = sc;
        dst[1] = beta[c] - (float)mean * sc;
    }
}
}  // namespace

extern "C" int edtr_gn_table(const float* partial, int tiles_per_image, const double* sums, int B, int C, int groups, int HW,
                             const float* gamma, const float* beta, float eps, float* table, edtr_stream_t stream) {
    if ((!partial && !sums) || !gamma || !beta || !table) return EDTR_E_NULL;
    if ((partial && tiles_per_image <= 0) || B <= 0 || C <= 0 || groups <= 0 || C % groups || HW <= 0) return EDTR_E_SHAPE;
    hipLaunchKernelGGL(gn_table_kernel, dim3(groups, B), dim3(256), 0, static_cast<hipStream_t>(stream), partial, tiles_per_image, sums,
                       C, groups, HW, gamma, beta, eps, table);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

static int igemm_run(const edtr_igemm_params* pp, edtr_stream_t stream, bool dry) {
    if (!pp) return EDTR_E_NULL;
    edtr_igemm_params p = *pp;
    if (!p.a1 || !p.w || !p.out) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.taps != 1 && p.taps != 9) return EDTR_E_UNSUPPORTED;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.Z <= 0) return EDTR_E_SHAPE;
    if (p.zdiv <= 0) p.zdiv = 1;
    if (p.C2 > 0 && !p.a2) return EDTR_E_NULL;
    if (p.C2 < 0 || p.C1 <= 0) return EDTR_E_SHAPE;
    if (p.K != p.taps * (p.C1 + p.C2)) return EDTR_E_SHAPE;
    const bool spatial = p.OH > 0;
    if (p.taps == 9 && !spatial) return EDTR_E_SHAPE;
    if (spatial) {
        if (p.OW <= 0 || p.IH <= 0 || p.IW <= 0 || p.stride <= 0) return EDTR_E_SHAPE;
        if (p.M % (p.OH * p.OW) != 0) return EDTR_E_SHAPE;
    }
    if (p.upsample2x < 0 || p.upsample2x > 2) return EDTR_E_SHAPE;
    if (p.act < EDTR_ACT_NONE || p.act > EDTR_ACT_LRELU) return EDTR_E_DTYPE;
    if (p.act == EDTR_ACT_LRELU && !(p.act_slope >= 0.0f && p.act_slope <= 1.0f)) return EDTR_E_SHAPE;
    if (p.rowvec && p.rows_per_image <= 0) return EDTR_E_SHAPE;
    // 16-byte rule
    if ((p.K & 7) || (p.N & 7) || (p.C1 & 7) || (p.C2 & 7) || (p.ld1 & 7) || (p.C2 && (p.ld2 & 7)) || (p.ldw & 7))
        return EDTR_E_ALIGN;
    const int n_out = p.act == EDTR_ACT_GEGLU ? p.N / 2 : p.N;
    if (p.act == EDTR_ACT_GEGLU && (p.N & 63)) return EDTR_E_ALIGN;
    if ((n_out & 7) || (p.ldc & (p.out_f32 ? 3 : 7)) || (p.residual && (p.ldr & (p.residual_f32 ? 3 : 7)))) return EDTR_E_ALIGN;
    if (!aligned16(p.a1) || !aligned16(p.w) || !aligned16(p.out) || (p.a2 && !aligned16(p.a2)) ||
        (p.residual && !aligned16(p.residual)))
        return EDTR_E_ALIGN;
    if ((p.a_zs_outer & 7) || (p.a_zs_inner & 7) || (p.w_zs_outer & 7) || (p.w_zs_inner & 7) ||
        (p.o_zs_outer & 3) || (p.o_zs_inner & 3))
        return EDTR_E_ALIGN;
    // the epilogue reads bias / time-embedding rows as 16-byte vectors
    if ((p.bias_n && !aligned16(p.bias_n)) || (p.rowvec && (!aligned16(p.rowvec) || (p.rowvec_ld & 3)))) return EDTR_E_ALIGN;
    if (p.n_valid < 0 || p.n_valid > p.N) return EDTR_E_SHAPE;
    if (p.residual && p.Z != 1) return EDTR_E_UNSUPPORTED;  // residual / rowvec are not z-batched
    if (p.rowvec && p.Z != 1) return EDTR_E_UNSUPPORTED;
    if (p.splitk < 0) return EDTR_E_SHAPE;
    if (p.splitk <= 1) p.splitk = 1;
    if (p.splitk > 1) {
        if (p.Z != 1 || p.act == EDTR_ACT_GEGLU) return EDTR_E_UNSUPPORTED;
        if (!p.workspace) return EDTR_E_NULL;
        if (p.workspace_bytes < (int64_t)p.splitk * p.M * p.N * 4 || !aligned16(p.workspace)) return EDTR_E_SHAPE;
        if (p.splitk > (p.K + 63) / 64) return EDTR_E_SHAPE;
    }

    if (p.vt_out) {
        if (p.Z != 1 || p.splitk > 1 || p.act == EDTR_ACT_GEGLU || p.residual || p.rowvec || p.gn_partial || p.out_f32 || spatial)
            return EDTR_E_UNSUPPORTED;
        if (p.rows_per_image <= 0 || (p.rows_per_image & 7) || (p.M & 7) || p.vt_col0 <= 0 || p.vt_col0 >= p.N) return EDTR_E_SHAPE;
        if ((p.vt_ld & 7) || p.vt_ld < p.rows_per_image || !aligned16(p.vt_out)) return EDTR_E_ALIGN;
    }

    if (p.row_stats) {
        if (p.Z != 1 || p.splitk > 1 || p.act == EDTR_ACT_GEGLU || p.vt_out) return EDTR_E_UNSUPPORTED;
        if (p.N & 31) return EDTR_E_SHAPE;
        if (!aligned16(p.row_stats)) return EDTR_E_ALIGN;
    }
    if (p.ln_stats) {
        if (p.Z != 1 || p.splitk > 1 || p.taps != 1 || spatial || p.C2) return EDTR_E_UNSUPPORTED;
        if (!p.ln_c1 || !p.ln_c2) return EDTR_E_NULL;
        // ln_C = the number of REAL columns the statistics run over (SwinIR keeps 180 channels in rows of 192 whose pad columns are
        // exactly zero: they add nothing to the slots); the slots cover the K = ln_slots * 32 stored columns
        if (p.ln_slots <= 0 || p.ln_slots * 32 != p.K || p.ln_C <= 0 || p.ln_C > p.K) return EDTR_E_SHAPE;
        if (!aligned16(p.ln_stats) || !aligned16(p.ln_c1) || !aligned16(p.ln_c2)) return EDTR_E_ALIGN;
    }

    // the LDS-DMA main loops need every 64-wide K-tile inside one tap of one source
    const bool dma_ok = p.C2 == 0 && (p.C1 & 63) == 0;
    int tile = p.tile;
    if (tile == 0) {   // default: the LDS-DMA kernels wherever they apply
        const int64_t big = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.Z;
        tile = dma_ok ? 3 : (big >= 200 ? 1 : 2);
        // 256x256 ping-pong kernel: large-K convolutions whose 256-column tiles are mostly full and fill >= half the CUs
        // (measured on MI355X: 1.17-1.32 PFLOP/s vs 0.98-1.05 for the 128x128 loop on the VAE convolutions)
        const int nbn256 = (p.N + 255) / 256;
        const int64_t nb256 = (int64_t)((p.M + 255) / 256) * nbn256 * p.Z;
        // A/B switch for measurements on ONE device (devices differ by several percent): bit i of EDTR_IGEMM_NO_AUTO
        // disables the automatic choice of tile i (6, 8, 9); explicit p.tile requests are unaffected.
        static int no_auto = -1;
        if (no_auto < 0) {
            const char* e = getenv("EDTR_IGEMM_NO_AUTO");
            no_auto = e ? atoi(e) : 0;
        }
        const bool pp_ok = dma_ok && p.splitk <= 1 && p.act != EDTR_ACT_GEGLU && (!p.upsample2x || p.stride == 1) &&
                           igemm_fast_addressable(p, spatial);
        // 128x160 tile: the SD UNet widths 320 / 640 / 1280 are multiples of 160 (no padded columns; N = 320 at M = 32768 is
        // one resident round).  Measured faster than the 128-wide grid whenever >= 200 tiles exist, except the
        // short-K GEMMs whose N the 128-wide grid also divides (two-pass epilogue).
        const int64_t nb160 = (int64_t)((p.M + 127) / 128) * (p.N / 160) * p.Z;
        if (pp_ok && !(no_auto & (1 << 8)) && p.N % 160 == 0 && nb160 >= 200 && (p.N % 128 != 0 || p.K >= 640))
            tile = 8;
        else if (pp_ok && !(no_auto & (1 << 6)) && p.K >= 1024 && nb256 >= 120 && p.N * 5 >= nbn256 * 256 * 4)
            tile = 6;
        // Halo tile: 3x3 / stride 1 convolutions on 16-pixel-aligned images.  Measured on the MI355X against the tile chosen above
        // (profiles/r02/halo_tile_experiments.log): 1.16-1.27x over the 128x128 loop on the N = 128 convolutions of the VAE's 512x512
        // level, 1.03-1.13x over the 256x256 ping-pong on its 256 / 128 / 64-pixel levels, 1.39x over the 128x160 tile at 640
        // channels; NOT where 128-column tiles pad N (N = 320: the 160-column tile wins) or with fewer than 48 units
        // (16x16-pixel patches x 128-column tiles; measured: 1.3-1.4x over the 128x160 tile at 80 units, 1.04-1.29x at 64, equal at 32).  EDTR_IGEMM_HALO=0 switches it off.
        static int halo = -1;
        if (halo < 0) {
            const char* e4 = getenv("EDTR_IGEMM_HALO");
            halo = (e4 && e4[0] == '0') ? 0 : 1;
        }
        if (halo && dma_ok && (p.N & 127) == 0 && igemm_halo_ok(p, spatial) && igemm_fast_addressable(p, spatial) &&
            (int64_t)(p.M >> 8) * (p.N >> 7) * p.splitk >= 48)      // (p.splitk was normalised to >= 1 above)
            tile = 16;
        // ... and where they DO pad N (N = 320 = 2.5 tiles) as long as the padded grid is at most ONE round of workgroups: the
        // 64x64-latent convolutions at batch <= 4 (192 units), where the 128x160 tile's 256 workgroups crawl alone on their CUs.
        // Measured (profiles/r03/halo_n320.log, one device): 37.0 / 63.1 / 88.9 us against 51.8 / 92.6 / 133.8 (K = 2880 / 5760 /
        // 8640, batch 4: 1.40 - 1.50x), 33.1 against 48.7 at batch 2; at batch 8 (384 units = 1.5 rounds) the 128x160 tile wins by 8 %.
        static int ragged = -1;
        if (ragged < 0) {
            const char* e6 = getenv("EDTR_IGEMM_HALO_RAGGED");
            ragged = (e6 && e6[0] == '0') ? 0 : 1;
        }
        if (halo && ragged && dma_ok && (p.N & 127) != 0 && p.N > 128 && p.splitk <= 1 && igemm_halo_ok(p, spatial) &&
            igemm_fast_addressable(p, spatial)) {
            const int64_t units = (int64_t)(p.M >> 8) * ((p.N + 127) >> 7);
            if (units >= 48 && units <= 256) tile = 16;
        }
        // 256x32 tile for skinny-N convolutions (the VAE decoder's 3-channel output conv: 94 % of a 128-wide tile is padding);
        // validated against tile 3 on the MI355X (profiles/r02/ab_tiles_3_vs_14*.log).  EDTR_IGEMM_SKINNY=0 switches it off.
        static int skinny = -1;
        if (skinny < 0) {
            const char* e3 = getenv("EDTR_IGEMM_SKINNY");
            skinny = (e3 && e3[0] == '0') ? 0 : 1;
        }
        if (skinny && tile == 3 && p.N <= 32 && p.M >= 65536 && p.splitk <= 1 && !p.gn_partial && p.act != EDTR_ACT_GEGLU &&
            (!p.upsample2x || p.stride == 1) && igemm_fast_addressable(p, spatial))
            tile = 14;
    }
    // Halo tile with 512-pixel units (tile 17, halo512.hip): the plain 3x3 / stride-1 convolutions whose image is a multiple of
    // 32 x 16 pixels, N a multiple of 128, no split-K / activation — wherever the 256-pixel halo tile was chosen and there are at
    // least two rounds of 512-pixel units.  EDTR_IGEMM_HALO512=0 switches the automatic choice off (A/B on one device).
    static int halo512 = -1;
    if (halo512 < 0) {
        const char* e5 = getenv("EDTR_IGEMM_HALO512");
        halo512 = e5 ? atoi(e5) : 1;
    }
    const bool h512_ok = spatial && (p.C1 & 31) == 0 && edtr_halo512_ok(p) && igemm_fast_addressable(p, spatial);
    // (whole-round rules below count the CUs of the device the launch goes to; the HIP-free edtr_igemm_plan has no device and answers
    //  for the 256 CUs of an MI355X — ADVICE r05)
    const int cus = dry ? 256 : edtr_cu_count();
    if (p.tile == 0 && tile == 16 && halo512 > 0 && h512_ok && (int64_t)(p.M >> 9) * (p.N >> 7) >= (int64_t)cus * halo512) tile = 17;
    static int halo512p = -1;
    if (halo512p < 0) {
        const char* e13 = getenv("EDTR_IGEMM_HALO512P");
        halo512p = e13 ? atoi(e13) : 0;
    }
    if (p.tile == 0 && tile == 17 && halo512p > 0 && edtr_halo512p_ok(p) && (int64_t)(p.M >> 9) * (p.N >> 7) >= (int64_t)cus * halo512p) tile = 21;
    // Halo tile of 160 columns (tile 20): N % 160 == 0 convolutions whose 16 x 16-pixel x 160-channel units fill whole rounds of the
    // chip (the 64 x 64-latent ResBlock convolutions at batch 8: 256 units = one per CU), where the 128 x 160 implicit-GEMM tile would run.
    // EDTR_IGEMM_HALO160=0 switches the automatic choice off (A/B on one device).
    static int halo160 = -1;
    if (halo160 < 0) {
        const char* e10 = getenv("EDTR_IGEMM_HALO160");
        halo160 = (e10 && e10[0] == '0') ? 0 : 1;
    }
    if (p.tile == 0 && tile == 8 && halo160 && spatial && edtr_halo160_ok(p) && igemm_fast_addressable(p, spatial)) {
        const int64_t units = (int64_t)(p.M >> 8) * (p.N / 160), tail = units % cus, most = (int64_t)cus * 3 / 4;
        if (units >= most && (tail == 0 || tail >= most)) tile = 20;       // whole rounds of one unit per CU (the last one >= 3/4 full)
    }
    if (p.a_gn) {           // GroupNorm (+ SiLU) of the input fused into the halo tiles' patch staging: 16 x 16 / 32 x 16-patch geometries only
        if (p.tile == 21 || (p.tile == 0 && tile == 21)) {
            if (!h512_ok || !edtr_halo512p_ok(p)) return EDTR_E_UNSUPPORTED;
            tile = 21;
        } else if (p.tile == 17 || (p.tile == 0 && tile == 17)) {
            if (!h512_ok) return EDTR_E_UNSUPPORTED;
            tile = 17;
        } else {
            if ((p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) || !spatial || p.upsample2x || !dma_ok || !igemm_halo_ok(p, spatial) ||
                igemm_halo_img8(p) || !igemm_fast_addressable(p, spatial) || (p.tile != 0 && p.tile != 16))
                return EDTR_E_UNSUPPORTED;
            tile = 16;
        }
        if (!aligned16(p.a_gn)) return EDTR_E_ALIGN;
    }
    if (p.out16) {
        if (!p.out_f32 || p.Z != 1 || p.act == EDTR_ACT_GEGLU || p.vt_out) return EDTR_E_UNSUPPORTED;
        if ((p.ld16 & 7) || !aligned16(p.out16)) return EDTR_E_ALIGN;
    }
    if (p.a_wrap != 0) {          // weights-exact two-part product: A[m][k mod a_wrap] against [Wh | Wl]; plain GEMMs on tiles 1 / 2 / 3 / 8
        if (p.a_wrap < 0 || p.K != 2 * p.a_wrap || (p.a_wrap & 63) || spatial || p.taps != 1 || p.C2 || p.Z != 1 || p.ln_stats) return EDTR_E_UNSUPPORTED;
        if (p.tile != 0 && !(p.tile == 1 || p.tile == 2 || p.tile == 3 || p.tile == 8)) return EDTR_E_UNSUPPORTED;
        if (!(tile == 1 || tile == 2 || tile == 3 || tile == 8)) tile = 3;       // (an automatic 256x256 / ping-pong choice)
    }
    if (p.upsample2x == 2) {      // sub-pixel form of the upsample convolution: the halo kernel only (phase-major pre-summed weights)
        if (p.tile != 0 && p.tile != 16) return EDTR_E_UNSUPPORTED;
        if (!dma_ok || !igemm_halo_ok(p, spatial) || !igemm_fast_addressable(p, spatial)) return EDTR_E_UNSUPPORTED;
        tile = 16;
    }
    if (p.vt_out) {     // the transposed V store needs whole column tiles of V: tiles 1 / 3 (128 columns) or 8 (160)
        if (p.tile == 0 && !((tile == 8 && p.vt_col0 % 160 == 0) || ((tile == 1 || tile == 3) && p.vt_col0 % 128 == 0)))
            tile = (dma_ok && p.N % 160 == 0 && p.vt_col0 % 160 == 0 && igemm_fast_addressable(p, spatial)) ? 8
                   : (p.vt_col0 % 128 == 0 ? (dma_ok ? 3 : 1) : -1);
        const int bn = tile == 8 ? 160 : 128;
        if (!(tile == 1 || tile == 3 || tile == 8) || p.vt_col0 % bn != 0) return EDTR_E_UNSUPPORTED;
    }
    if (p.ln_stats && !(tile == 1 || tile == 3)) {
        if (p.tile != 0) return EDTR_E_UNSUPPORTED;       // the folded LayerNorm's epilogue exists in the 128 x 128 tiles only
        tile = dma_ok ? 3 : 1;
        if (p.vt_out && p.vt_col0 % 128 != 0) return EDTR_E_UNSUPPORTED;
    }
    if (p.row_stats && !(tile == 1 || tile == 2 || tile == 3 || (tile == 8 && !(dbg_flags() & 2)))) {   // ... the row statistics: + the 64x64 tile 2 and the 128x160 tile
        if (p.tile != 0) return EDTR_E_UNSUPPORTED;
        tile = dma_ok ? 3 : 1;
    }
    if ((p.row_stats || p.vt_out) && (tile == 16 || tile == 17 || tile == 20 || tile == 21)) return EDTR_E_UNSUPPORTED;
    if (p.act == EDTR_ACT_GEGLU && tile == 2) tile = 1;  // value/gate pairing needs two 32-column MFMA tiles per wave
    if (tile >= 3 && !dma_ok && !((tile == 17 || tile == 20 || tile == 21) && p.C2 == 0 && (p.C1 & 31) == 0)) return EDTR_E_UNSUPPORTED;      // (tiles 17 / 20 walk 32-channel chunks)
    if (p.gn_partial) {
        const int sr = p.gn_slot_rows > 0 ? p.gn_slot_rows : 128;
        if (p.gn_ld < 0 || (p.gn_ld > 0 && p.gn_ld < p.N)) return EDTR_E_SHAPE;
        if (p.Z != 1 || p.act == EDTR_ACT_GEGLU) return EDTR_E_UNSUPPORTED;
        if (p.splitk > 1) {          // the split-K reducer writes the statistics: slots of 128 rows, or of 64 (the 8 x 8 images)
            if ((sr != 64 && sr != 128) || p.M % sr || (p.N & 31)) return EDTR_E_UNSUPPORTED;
        } else if (tile == 2 || sr != 128 || (p.M & 127)) {
            return EDTR_E_UNSUPPORTED;   // fused GroupNorm statistics need whole 128-row tiles of the 128x128 kernels
        }
    }
    // live tiles: 1, 2 (register-staged), 3 (LDS-DMA 128x128), 6 (256x256 ping-pong), 8 (128x160), 14 (256x32), 16 (halo), 17 (halo, 512-pixel units).
    // 4, 5, 7, 9 - 13, 15, 18 (and a round-2 "17") were experiments, measured (profiles/r01 - r03) and removed (15 = the 8-wave ping-pong 128x128 tile
    // for small grids and 18 = the persistent halo tile were faster in isolation and neutral on the whole path: round 4 took them out)
    if (!(tile == 1 || tile == 2 || tile == 3 || tile == 6 || tile == 8 || tile == 14 || tile == 16 || tile == 17 || tile == 20 || tile == 21)) return EDTR_E_DTYPE;
    // lockstep breaker of the two-workgroups-per-CU kernels (stagger_second_slot): only when the grid has more than one round
    // (>= 768 workgroups: below that the second slot's blocks are the tail anyway) and the tile is short enough for the epilogue
    // to matter.  EDTR_IGEMM_STAGGER = percent of the estimated half life (default 100; 0 = off).
    p.debug_flags = dbg_flags();
    p.stagger = 0;
    if ((tile == 3 || tile == 8) && p.splitk <= 1) {
        static int pct = -1;
        if (pct < 0) {
            const char* e7 = getenv("EDTR_IGEMM_STAGGER");
            pct = e7 ? atoi(e7) : 0;       // measured: no gain in isolation (the two workgroups of a CU are not slowed by their lockstep), off
        }
        const int bm = 128, bn = tile == 8 ? 160 : 128;
        const int64_t wgs = (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.Z;
        const int nkt = (p.K + 63) / 64;
        if (pct > 0 && wgs >= 768 && nkt <= 48) {
            const int life = 4000 + nkt * 1350 + (p.act == EDTR_ACT_GEGLU ? 11000 : 9000);      // cycles, from the in-kernel stamps
            p.stagger = (int)((int64_t)life * pct / 200);
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == EDTR_BF16 ? dispatch<BF16>(p, tile, spatial, s, dry) : dispatch<F16>(p, tile, spatial, s, dry);
}

extern "C" int edtr_igemm(const edtr_igemm_params* pp, edtr_stream_t stream) { return igemm_run(pp, stream, false); }

// Which kernel would edtr_igemm run for these parameters?  Every validation and every shape rule of the launch, no launch (no HIP
// call: usable without a GPU): > 0 = the tile number (edtr_igemm_params.tile), < 0 = the error edtr_igemm would return.
extern "C" int edtr_igemm_plan(const edtr_igemm_params* pp) { return igemm_run(pp, nullptr, true); }
