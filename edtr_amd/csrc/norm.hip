// HBM-bound normalisation kernels for gfx950: GroupNorm statistics / apply(+SiLU), LayerNorm,
// row softmax.  All 16-bit traffic is moved as 16-byte vectors (8 elements per lane), all
// arithmetic is fp32 (GroupNorm sums are combined across workgroups in fp64).
#include <stdlib.h>
#include "common.h"

namespace {

// Storage traits of the normalisation kernels.  IO16<T>: 16-bit in, 16-bit out (the throughput modes).  IOF32S: the
// high-precision mode (EDTR_F32_SPLIT) — the activation stream is fp32 and the normalised tensor is written as the bf16
// "split-3" GEMM operand  [hi | lo | hi]  (3*C columns: hi = bf16(x), lo = bf16(x - hi)), which the implicit GEMM
// multiplies with weights packed [Wh | Wh | Wl]:  hi*Wh + lo*Wh + hi*Wl = x*W to ~16 mantissa bits with fp32 accumulation.
template <typename T>
struct IO16 {
    using in_t = uint16_t;
    using out_t = uint16_t;
    static __device__ __forceinline__ void load8(const in_t* p, float (&f)[8]) { unpack8<T>(ldg16(p), f); }
    static __device__ __forceinline__ void store8(out_t* p, int, const float (&f)[8]) { stg16(p, pack8<T>(f)); }
};
// IOF32<OT, PARTS>: fp32 rows in, PARTS-part GEMM operand of 16-bit type OT out.  PARTS = 3: [hi | lo | hi] (bf16: the
// high-precision mode EDTR_F32_SPLIT; fp16: EDTR_F32_H3), 2: [hi | lo] (EDTR_F32_H2), 1: [x] (EDTR_F32_H1).
template <typename OT, int PARTS>
struct IOF32 {
    using in_t = float;
    using out_t = uint16_t;
    static __device__ __forceinline__ void load8(const in_t* p, float (&f)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[j + 4] = b[j]; }
    }
    static __device__ __forceinline__ void store8(out_t* p, int C, const float (&f)[8]) {
        const U4 hi = pack8<OT>(f);
        stg16(p, hi);
        if constexpr (PARTS >= 2) {
            float lo[8], back[8];
            unpack8<OT>(hi, back);
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = f[j] - back[j];
            stg16(p + C, pack8<OT>(lo));
        }
        if constexpr (PARTS == 3) stg16(p + 2 * C, hi);
    }
};
using IOF32S = IOF32<BF16, 3>;

// run `body(IO{})` for the storage traits of a dtype code (false: unknown code)
template <typename F>
bool with_io(int dtype, F&& body) {
    switch (dtype) {
        case EDTR_BF16: body(IO16<BF16>{}); return true;
        case EDTR_F16: body(IO16<F16>{}); return true;
        case EDTR_F32_SPLIT: body(IOF32<BF16, 3>{}); return true;
        case EDTR_F32_H1: body(IOF32<F16, 1>{}); return true;
        case EDTR_F32_H2: body(IOF32<F16, 2>{}); return true;
        case EDTR_F32_H3: body(IOF32<F16, 3>{}); return true;
        default: return false;
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid (pixel chunks <= 64 per image, B); block = CV * R threads (CV = C/8 channel
// vectors, R pixel rows in flight).  Each thread accumulates 8 channel sums / squares over its pixels in
// fp32 (4 independent 16-byte loads in flight), rows are folded through LDS, channels are folded into
// groups by 32 threads, and each workgroup issues ONE fp64 atomic pair per group (<= 64 arrivals per
// address per image, so the atomics never become the bottleneck).
// ------------------------------------------------------------------------------------------
template <typename IO>
__global__ void gn_stats_kernel(const edtr_gn_params p, int CV, int R, int ppb) {
    using in_t = typename IO::in_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);   // [2][R][C]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int C = p.C;
    const int cpg = C / p.groups;
    const int64_t pix0 = (int64_t)blockIdx.x * ppb;
    const int npix = (int)min((int64_t)ppb, (int64_t)p.HW - pix0);
    const in_t* xb = static_cast<const in_t*>(p.x) + ((int64_t)b * p.HW + pix0) * p.ldx;
    const int nslots = (CV + blockDim.x - 1) / blockDim.x;  // > 1 only when CV > blockDim (then R == 1)
    for (int slot = 0; slot < nslots; ++slot) {
        const int cv = R > 1 ? tid % CV : tid + slot * (int)blockDim.x;
        const int r = R > 1 ? tid / CV : 0;
        if (cv >= CV || r >= R) continue;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        const in_t* xp = xb + cv * 8;
        int pi = r;
        for (; pi + 3 * R < npix; pi += 4 * R) {
            float fv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) IO::load8(xp + (int64_t)(pi + u * R) * p.ldx, fv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += fv[u][j]; q[j] += fv[u][j] * fv[u][j]; }
            }
        }
        for (; pi < npix; pi += R) {
            float f[8];
            IO::load8(xp + (int64_t)pi * p.ldx, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[r * C + cv * 8 + j] = s[j];
            red[(R + r) * C + cv * 8 + j] = q[j];
        }
    }
    __syncthreads();
    // fold rows: channel c -> red[c], red[R*C + c]
    for (int c = tid; c < C; c += blockDim.x) {
        float a = 0.0f, bq = 0.0f;
        for (int r = 0; r < R; ++r) { a += red[r * C + c]; bq += red[(R + r) * C + c]; }
        red[c] = a;
        red[R * C + c] = bq;
    }
    __syncthreads();
    if (tid < p.groups) {
        float a = 0.0f, bq = 0.0f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += red[c]; bq += red[R * C + c]; }
        double* dst = p.sums + ((int64_t)b * p.groups + tid) * 2;
        atomicAdd(dst, (double)a);
        atomicAdd(dst + 1, (double)bq);
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm apply (+SiLU).  grid (pixel chunks, channel chunks of <= 256, B), 256 threads.  Per-channel
// scale/shift (rstd*gamma, beta - mean*rstd*gamma) of the chunk are built once per workgroup in LDS, then
// the [pixels x channels] slab is streamed as 16-byte vectors, 4 independent loads in flight per thread.
// ------------------------------------------------------------------------------------------
constexpr int GN_CC = 256;  // channels per workgroup in the apply kernel
// A/B switch for measurements: EDTR_GN_APPLY_GENERIC=1 keeps every chunk on the generic (divide per vector) loop
__device__ int g_gn_apply_generic = 0;
__device__ __forceinline__ bool gn_apply_generic_only() { return g_gn_apply_generic != 0; }

template <typename IO>
__global__ void __launch_bounds__(256) gn_apply_kernel(const edtr_gn_params p, int ppb) {
    using in_t = typename IO::in_t;
    using out_t = typename IO::out_t;
    __shared__ __attribute__((aligned(16))) float sc[GN_CC];
    __shared__ __attribute__((aligned(16))) float sh[GN_CC];
    // fused finalize: per-channel totals of the producer's tile partials, for the chunk's channels extended to whole groups
    // (the SD widths' 10 / 20 / 40 channels per group do not divide the 256-channel chunk: <= 2 x 63 extra channels)
    __shared__ double chan_s[GN_CC + 128], chan_q[GN_CC + 128];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int c0 = blockIdx.y * GN_CC;
    const int cc = min(GN_CC, p.C - c0);
    const int cpg = p.C / p.groups;
    const double inv_cnt = 1.0 / ((double)p.HW * cpg);
    const int glo = (c0 / cpg) * cpg;                                   // first channel of the first group touching the chunk
    if (p.partial) {
        // every workgroup folds the <= 64 tiles x (its channels, extended to whole groups) itself — a few KB from L2 — instead
        // of a separate finalize launch
        const int ghi = min(p.C, ((c0 + cc + cpg - 1) / cpg) * cpg);
        for (int ch = glo + tid; ch < ghi; ch += 256) {
            const float* src = p.partial + ((int64_t)b * p.tiles_per_image * p.C + ch) * 2;
            double s = 0.0, q = 0.0;
            for (int t = 0; t < p.tiles_per_image; ++t) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(src + (int64_t)t * p.C * 2);
                s += (double)v[0];
                q += (double)v[1];
            }
            chan_s[ch - glo] = s;
            chan_q[ch - glo] = q;
        }
        __syncthreads();
    }
    for (int c = tid; c < cc; c += 256) {
        double gsum[2];
        if (p.partial) {
            const int g0 = ((c0 + c) / cpg) * cpg - glo;      // first channel of this channel's group, relative to glo
            double s = 0.0, q = 0.0;
            for (int j = 0; j < cpg; ++j) { s += chan_s[g0 + j]; q += chan_q[g0 + j]; }
            gsum[0] = s;
            gsum[1] = q;
        } else {
            const double* srcd = p.sums + ((int64_t)b * p.groups + (c0 + c) / cpg) * 2;
            gsum[0] = srcd[0];
            gsum[1] = srcd[1];
        }
        // mean / variance in fp64 (the sums are fp64: no cancellation), then ONE fp32 v_rsq instead of an fp64 division + square
        // root + division per channel: with ~2048 small workgroups per launch this prologue was a third of a small tensor's time
        const double mean = gsum[0] * inv_cnt;
        double var = gsum[1] * inv_cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float rstd = __builtin_amdgcn_rsqf((float)var + p.eps);
        const float g = p.gamma[c0 + c] * rstd;
        sc[c] = g;
        sh[c] = p.beta[c0 + c] - (float)mean * g;
    }
    __syncthreads();
    const int ncv = cc >> 3;
    const int64_t pix0 = (int64_t)blockIdx.x * ppb;
    const int npix = (int)min((int64_t)ppb, (int64_t)p.HW - pix0);
    const in_t* xb = static_cast<const in_t*>(p.x) + ((int64_t)b * p.HW + pix0) * p.ldx + c0;
    out_t* yb = static_cast<out_t*>(p.y) + ((int64_t)b * p.HW + pix0) * p.ldy + c0;
    const int total = npix * ncv;
    if ((ncv & (ncv - 1)) == 0 && ncv <= 256 && !gn_apply_generic_only()) {
        // The chunk's vector count is a power of two (every SD / VAE width: 128 / 256 / 512, and 320 = 256 + 64, 640 = 2 x 256 +
        // 128, ...): a thread keeps ONE 8-channel vector for all of its pixels, so scale / shift live in registers and the
        // (pixel, vector) split is a shift — the generic loop below spends as many instructions on its integer division and
        // four LDS reads per vector as on the normalisation itself.
        const int sh_ = __builtin_ctz(ncv), cvf = tid & (ncv - 1), prow = tid >> sh_, pstep = 256 >> sh_;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc + cvf * 8), a1 = *reinterpret_cast<const f32x4*>(sc + cvf * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh + cvf * 8), b1 = *reinterpret_cast<const f32x4*>(sh + cvf * 8 + 4);
        const in_t* xt = xb + cvf * 8;
        out_t* yt = yb + cvf * 8;
        for (int px0 = prow; px0 < npix; px0 += 4 * pstep) {
            float fv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (px0 + u * pstep < npix) IO::load8(xt + (int64_t)(px0 + u * pstep) * p.ldx, fv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (px0 + u * pstep >= npix) continue;
                float (&f)[8] = fv[u];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[j] = f[j] * a0[j] + b0[j];
                    f[j + 4] = f[j + 4] * a1[j] + b1[j];
                }
                if (p.silu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
                }
                IO::store8(yt + (int64_t)(px0 + u * pstep) * p.ldy, p.C, f);
            }
        }
        return;
    }
    for (int i0 = tid; i0 < total; i0 += 4 * 256) {
        float fv[4][8];
        int pi[4], cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            pi[u] = i / ncv;
            cv[u] = i - pi[u] * ncv;
            if (i < total) IO::load8(xb + (int64_t)pi[u] * p.ldx + cv[u] * 8, fv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * 256 >= total) continue;
            float (&f)[8] = fv[u];
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc + cv[u] * 8), a1 = *reinterpret_cast<const f32x4*>(sc + cv[u] * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh + cv[u] * 8), b1 = *reinterpret_cast<const f32x4*>(sh + cv[u] * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[j] = f[j] * a0[j] + b0[j];
                f[j + 4] = f[j + 4] * a1[j] + b1[j];
            }
            if (p.silu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
            }
            IO::store8(yb + (int64_t)pi[u] * p.ldy + cv[u] * 8, p.C, f);
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (up to 4 vectors of 8 per lane: C <= 2048),
// mean / variance by wavefront shuffles.
// ------------------------------------------------------------------------------------------
// PADDED: only the first c_valid (< C) columns are real; the rest are excluded from the statistics and written as zeros.
template <typename IO, bool PADDED>
__global__ void __launch_bounds__(256) layernorm_kernel(const typename IO::in_t* x, int64_t rows, int C, int c_valid, int ldx,
                                                       const float* gamma, const float* beta, float eps,
                                                       typename IO::out_t* y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CV = C >> 3;
    const float n_cols = (float)(PADDED ? c_valid : C);
    float f[4][8];
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cv = lane + 64 * s;
        if (cv < CV) {
            IO::load8(x + row * ldx + cv * 8, f[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (PADDED && cv * 8 + j >= c_valid) f[s][j] = 0.0f;
                sum += f[s][j];
            }
        }
    }
    const float mean = wave_sum(sum) / n_cols;
    float sq = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cv = lane + 64 * s;
        if (cv < CV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = f[s][j] - mean;
                if (!PADDED || cv * 8 + j < c_valid) sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / n_cols + eps);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cv = lane + 64 * s;
        if (cv < CV) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (f[s][j] - mean) * rstd * gamma[cv * 8 + j] + beta[cv * 8 + j];
                if (PADDED && cv * 8 + j >= c_valid) o[j] = 0.0f;
            }
            IO::store8(y + row * ldy + cv * 8, C, o);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Row softmax, fp32 in -> 16-bit out.  One workgroup per row; online (max, sum) pass then a
// normalising pass (the row is re-read from L2).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* s, int cols, int64_t ld_s, uint16_t* pout,
                                                          int64_t ld_p, int cols_pad) {
    __shared__ float red_m[4], red_l[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = s + (int64_t)blockIdx.x * ld_s;
    uint16_t* orow = pout + (int64_t)blockIdx.x * ld_p;
    float m = -1e30f, l = 0.0f;
    for (int c = tid * 4; c < cols; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
        const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        const float mn = fmaxf(m, vm);
        l = l * __expf(m - mn) + __expf(v[0] - mn) + __expf(v[1] - mn) + __expf(v[2] - mn) + __expf(v[3] - mn);
        m = mn;
    }
    const float wm = wave_max(m);
    l = wave_sum(l * __expf(m - wm));
    if (lane == 0) { red_m[wave] = wm; red_l[wave] = l; }
    __syncthreads();
    const float gm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float gl = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) gl += red_l[w] * __expf(red_m[w] - gm);
    const float inv = 1.0f / gl;
    for (int c = tid * 4; c < cols; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
        uint2 o;
        o.x = pack2<T>(__expf(v[0] - gm) * inv, __expf(v[1] - gm) * inv);
        o.y = pack2<T>(__expf(v[2] - gm) * inv, __expf(v[3] - gm) * inv);
        *reinterpret_cast<uint2*>(orow + c) = o;
    }
    for (int c = cols + tid * 4; c < cols_pad; c += 1024) *reinterpret_cast<uint2*>(orow + c) = make_uint2(0u, 0u);
}

int check_gn(const edtr_gn_params& p, bool apply) {
    if (!p.x || (!p.sums && !(apply && p.partial))) return EDTR_E_NULL;
    if (apply && (!p.y || !p.gamma || !p.beta)) return EDTR_E_NULL;
    if (p.dtype < EDTR_BF16 || p.dtype > EDTR_F32_H3) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.HW <= 0 || p.C <= 0 || p.groups <= 0 || p.groups > 64) return EDTR_E_SHAPE;
    if (p.C % p.groups) return EDTR_E_SHAPE;
    if ((p.C & 7) || (p.ldx & 7) || (apply && (p.ldy & 7))) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || (apply && !aligned16(p.y))) return EDTR_E_ALIGN;
    if (p.C > 8192) return EDTR_E_UNSUPPORTED;
    return EDTR_OK;
}

}  // namespace

static int gn_pixels_per_block(int64_t HW, int B, int min_ppb, int max_chunks_per_image) {
    // enough workgroups to fill 256 CUs, but never more than max_chunks_per_image per image
    int64_t chunks = (1024 + B - 1) / B;
    if (chunks > max_chunks_per_image) chunks = max_chunks_per_image;
    int64_t ppb = (HW + chunks - 1) / chunks;
    if (ppb < min_ppb) ppb = min_ppb;
    return (int)ppb;
}

extern "C" int edtr_gn_stats(const edtr_gn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_gn_params& p = *pp;
    if (int e = check_gn(p, false)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!p.sums_zeroed) {
        hipError_t e = hipMemsetAsync(p.sums, 0, sizeof(double) * 2 * p.groups * p.B, s);
        if (e != hipSuccess) return (int)e;
    }
    const int CV = p.C >> 3;
    int R = CV >= 256 ? 1 : 256 / CV;
    if (R < 1) R = 1;
    int threads = CV >= 256 ? 256 : CV * R;
    threads = ((threads + 63) / 64) * 64;
    if (threads < 64) threads = 64;
    const int ppb = gn_pixels_per_block(p.HW, p.B, 16, 64);
    dim3 grid((unsigned)((p.HW + ppb - 1) / ppb), p.B);
    const size_t lds = sizeof(float) * 2 * R * p.C;
    if (p.dtype == EDTR_BF16)
        hipLaunchKernelGGL(gn_stats_kernel<IO16<BF16>>, grid, dim3(threads), lds, s, p, CV, R, ppb);
    else if (p.dtype == EDTR_F16)
        hipLaunchKernelGGL(gn_stats_kernel<IO16<F16>>, grid, dim3(threads), lds, s, p, CV, R, ppb);
    else      // every fp32-stream code reads the same fp32 rows
        hipLaunchKernelGGL(gn_stats_kernel<IOF32S>, grid, dim3(threads), lds, s, p, CV, R, ppb);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_gn_apply(const edtr_gn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_gn_params& p = *pp;
    if (int e = check_gn(p, true)) return e;
    if (p.partial) {      // fused finalize: few tiles; a chunk extended to whole groups must fit the LDS arrays
        const int cpg = p.C / p.groups;
        // slots of 128 rows (the main-loop epilogues) or of 64 (the split-K reducer's statistics of 8 x 8 images: edtr_igemm gn_slot_rows)
        if (p.tiles_per_image <= 0 || p.tiles_per_image > 64 || ((int64_t)p.tiles_per_image * 128 != p.HW && (int64_t)p.tiles_per_image * 64 != p.HW))
            return EDTR_E_SHAPE;
        if (cpg > 64) return EDTR_E_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    static int generic = -1;
    if (generic < 0) {
        const char* e = getenv("EDTR_GN_APPLY_GENERIC");
        generic = (e && e[0] == '1') ? 1 : 0;
        if (generic) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gn_apply_generic), &generic, sizeof(int));
    }
    const int nchunk_c = (p.C + GN_CC - 1) / GN_CC;
    // ~2048 workgroups when the tensor allows it, at least 8 pixels (>= 1 vector per thread at 256 channels) each
    int64_t want = (2048 + (int64_t)p.B * nchunk_c - 1) / ((int64_t)p.B * nchunk_c);
    if (want < 1) want = 1;
    int64_t ppb = (p.HW + want - 1) / want;
    if (ppb < 8) ppb = 8;
    if (ppb > 256) ppb = 256;
    dim3 grid((unsigned)((p.HW + ppb - 1) / ppb), nchunk_c, p.B);
    with_io(p.dtype, [&](auto io) {
        hipLaunchKernelGGL(gn_apply_kernel<decltype(io)>, grid, dim3(256), 0, s, p, (int)ppb);
    });
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

// Tiled-VAE GroupNorm pooling: sums is [T][BG][2] fp64 (per tile: sum, sum of squares of each (image, group)).
// Per tile mean / biased variance, weighted average over tiles (weights[t]), written back as the (sum, sumsq) pair
// that makes edtr_gn_apply reproduce exactly (pooled mean, pooled variance) for that tile's element count.
namespace {
__global__ void gn_pool_kernel(double* sums, const float* weights, const float* counts, int T, int BG) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BG) return;
    double mean = 0.0, var = 0.0;
    for (int t = 0; t < T; ++t) {
        const double* s = sums + ((int64_t)t * BG + i) * 2;
        const double c = (double)counts[t];
        const double m = s[0] / c;
        double v = s[1] / c - m * m;
        v = v < 0.0 ? 0.0 : v;
        mean += (double)weights[t] * m;
        var += (double)weights[t] * v;
    }
    for (int t = 0; t < T; ++t) {
        double* s = sums + ((int64_t)t * BG + i) * 2;
        const double c = (double)counts[t];
        s[0] = mean * c;
        s[1] = (var + mean * mean) * c;
    }
}
}  // namespace

extern "C" int edtr_gn_pool(double* sums, const float* weights, const float* counts, int T, int BG, edtr_stream_t stream) {
    if (!sums || !weights || !counts) return EDTR_E_NULL;
    if (T <= 0 || BG <= 0) return EDTR_E_SHAPE;
    hipLaunchKernelGGL(gn_pool_kernel, dim3((BG + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sums, weights,
                       counts, T, BG);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_layernorm(int dtype, const void* x, int64_t rows, int C, int c_valid, int ldx, const float* gamma,
                              const float* beta, float eps, void* y, int ldy, edtr_stream_t stream) {
    if (!x || !y || !gamma || !beta) return EDTR_E_NULL;
    if (dtype < EDTR_BF16 || dtype > EDTR_F32_H3) return EDTR_E_DTYPE;
    if (rows <= 0 || C <= 0 || c_valid < 0 || c_valid > C) return EDTR_E_SHAPE;
    if (C > 2048) return EDTR_E_UNSUPPORTED;
    if ((C & 7) || (ldx & 7) || (ldy & 7) || !aligned16(x) || !aligned16(y)) return EDTR_E_ALIGN;
    if (c_valid == 0) c_valid = C;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + 3) / 4));
    const uint16_t* xp = static_cast<const uint16_t*>(x);
    uint16_t* yp = static_cast<uint16_t*>(y);
    if (dtype >= EDTR_F32_SPLIT) {       // fp32 rows in, 1..3-part GEMM operand out
        if (c_valid != C) return EDTR_E_UNSUPPORTED;
        with_io(dtype, [&](auto io) {
            using IO = decltype(io);
            if constexpr (sizeof(typename IO::in_t) == 4)
                hipLaunchKernelGGL((layernorm_kernel<IO, false>), grid, dim3(256), 0, s, static_cast<const float*>(x), rows, C, C, ldx, gamma, beta, eps, yp, ldy);
        });
    } else if (dtype == EDTR_BF16) {
        if (c_valid == C) hipLaunchKernelGGL((layernorm_kernel<IO16<BF16>, false>), grid, dim3(256), 0, s, xp, rows, C, C, ldx, gamma, beta, eps, yp, ldy);
        else hipLaunchKernelGGL((layernorm_kernel<IO16<BF16>, true>), grid, dim3(256), 0, s, xp, rows, C, c_valid, ldx, gamma, beta, eps, yp, ldy);
    } else {
        if (c_valid == C) hipLaunchKernelGGL((layernorm_kernel<IO16<F16>, false>), grid, dim3(256), 0, s, xp, rows, C, C, ldx, gamma, beta, eps, yp, ldy);
        else hipLaunchKernelGGL((layernorm_kernel<IO16<F16>, true>), grid, dim3(256), 0, s, xp, rows, C, c_valid, ldx, gamma, beta, eps, yp, ldy);
    }
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_softmax_rows(int dtype, const float* sc, int64_t rows, int cols, int64_t ld_s, void* pr,
                                 int64_t ld_p, int cols_pad, edtr_stream_t stream) {
    if (!sc || !pr) return EDTR_E_NULL;
    if (dtype != EDTR_BF16 && dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (rows <= 0 || cols <= 0 || rows > 0x7fffffffLL) return EDTR_E_SHAPE;
    if ((cols & 3) || (ld_s & 3) || (ld_p & 3) || (cols_pad & 3) || !aligned16(sc) || (reinterpret_cast<uintptr_t>(pr) & 7))
        return EDTR_E_ALIGN;
    if (cols_pad < cols) cols_pad = cols;
    if (cols_pad > ld_p) return EDTR_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EDTR_BF16)
        hipLaunchKernelGGL(softmax_rows_kernel<BF16>, dim3((unsigned)rows), dim3(256), 0, s, sc, cols, ld_s,
                           static_cast<uint16_t*>(pr), ld_p, cols_pad);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<F16>, dim3((unsigned)rows), dim3(256), 0, s, sc, cols, ld_s,
                           static_cast<uint16_t*>(pr), ld_p, cols_pad);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}
