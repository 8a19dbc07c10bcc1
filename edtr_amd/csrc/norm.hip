// HBM-bound normalisation kernels for gfx950: GroupNorm statistics / apply(+SiLU), LayerNorm,
// row softmax.  All 16-bit traffic is moved as 16-byte vectors (8 elements per lane), all
// arithmetic is fp32 (GroupNorm sums are combined across workgroups in fp64).
#include "common.h"

namespace {

constexpr int GN_PIX = 256;  // pixels per workgroup (stats and apply)

// ------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid (ceil(HW / GN_PIX), B); block = CV * R threads where CV = C/8 channel
// vectors and R rows in flight; thread (row r, vector cv) accumulates 8 channel sums / squares over
// its pixels in fp32, folds them into per-group LDS accumulators, then one fp64 atomic per group.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_stats_kernel(const edtr_gn_params p, int CV, int R) {
    __shared__ float g_sum[64], g_sq[64];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    if (tid < 64) { g_sum[tid] = 0.0f; g_sq[tid] = 0.0f; }
    __syncthreads();
    const int cpg = p.C / p.groups;
    const int64_t pix0 = (int64_t)blockIdx.x * GN_PIX;
    const int npix = (int)min((int64_t)GN_PIX, (int64_t)p.HW - pix0);
    const uint16_t* xb = static_cast<const uint16_t*>(p.x) + ((int64_t)b * p.HW + pix0) * p.ldx;
    const int nslots = (CV + blockDim.x - 1) / blockDim.x;  // > 1 only when CV > blockDim (R == 1)
    for (int slot = 0; slot < nslots; ++slot) {
        const int cv = R > 1 ? tid % CV : tid + slot * blockDim.x;
        const int r = R > 1 ? tid / CV : 0;
        if (cv >= CV || r >= R) continue;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        for (int pi = r; pi < npix; pi += R) {
            float f[8];
            unpack8<T>(ldg16(xb + (int64_t)pi * p.ldx + cv * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (cv * 8 + j) / cpg;
            atomicAdd(&g_sum[g], s[j]);
            atomicAdd(&g_sq[g], q[j]);
        }
    }
    __syncthreads();
    if (tid < p.groups) {
        double* dst = p.sums + ((int64_t)b * p.groups + tid) * 2;
        atomicAdd(dst, (double)g_sum[tid]);
        atomicAdd(dst + 1, (double)g_sq[tid]);
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm apply (+SiLU).  grid (ceil(HW / GN_PIX), B), 256 threads.  Per-channel scale/shift
// (rstd*gamma, beta - mean*rstd*gamma) are built once per workgroup in LDS, then the pixel chunk is
// streamed as 16-byte vectors.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const edtr_gn_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw);
    float* sh = sc + p.C;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.groups;
    const double cnt = (double)p.HW * cpg;
    for (int c = tid; c < p.C; c += 256) {
        const double* src = p.sums + ((int64_t)b * p.groups + c / cpg) * 2;
        const double mean = src[0] / cnt;
        double var = src[1] / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float g = p.gamma[c] * rstd;
        sc[c] = g;
        sh[c] = p.beta[c] - (float)mean * g;
    }
    __syncthreads();
    const int CV = p.C >> 3;
    const int64_t pix0 = (int64_t)blockIdx.x * GN_PIX;
    const int npix = (int)min((int64_t)GN_PIX, (int64_t)p.HW - pix0);
    const uint16_t* xb = static_cast<const uint16_t*>(p.x) + ((int64_t)b * p.HW + pix0) * p.ldx;
    uint16_t* yb = static_cast<uint16_t*>(p.y) + ((int64_t)b * p.HW + pix0) * p.ldy;
    const int total = npix * CV;
    for (int i = tid; i < total; i += 256) {
        const int pi = i / CV, cv = i - pi * CV;
        float f[8];
        unpack8<T>(ldg16(xb + (int64_t)pi * p.ldx + cv * 8), f);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc + cv * 8), a1 = *reinterpret_cast<const f32x4*>(sc + cv * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh + cv * 8), b1 = *reinterpret_cast<const f32x4*>(sh + cv * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = f[j] * a0[j] + b0[j];
            f[j + 4] = f[j + 4] * a1[j] + b1[j];
        }
        if (p.silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
        }
        stg16(yb + (int64_t)pi * p.ldy + cv * 8, pack8<T>(f));
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (up to 4 vectors of 8 per lane: C <= 2048),
// mean / variance by wavefront shuffles.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) layernorm_kernel(const uint16_t* x, int64_t rows, int C, int ldx,
                                                       const float* gamma, const float* beta, float eps,
                                                       uint16_t* y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CV = C >> 3;
    float f[4][8];
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cv = lane + 64 * s;
        if (cv < CV) {
            unpack8<T>(ldg16(x + row * ldx + cv * 8), f[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += f[s][j];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cv = lane + 64 * s;
        if (cv < CV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[s][j] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cv = lane + 64 * s;
        if (cv < CV) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f[s][j] - mean) * rstd * gamma[cv * 8 + j] + beta[cv * 8 + j];
            stg16(y + row * ldy + cv * 8, pack8<T>(o));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Row softmax, fp32 in -> 16-bit out.  One workgroup per row; online (max, sum) pass then a
// normalising pass (the row is re-read from L2).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* s, int cols, int64_t ld_s, uint16_t* pout,
                                                          int64_t ld_p) {
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
}

int check_gn(const edtr_gn_params& p, bool apply) {
    if (!p.x || !p.sums) return EDTR_E_NULL;
    if (apply && (!p.y || !p.gamma || !p.beta)) return EDTR_E_NULL;
    if (p.dtype != EDTR_BF16 && p.dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (p.B <= 0 || p.HW <= 0 || p.C <= 0 || p.groups <= 0 || p.groups > 64) return EDTR_E_SHAPE;
    if (p.C % p.groups) return EDTR_E_SHAPE;
    if ((p.C & 7) || (p.ldx & 7) || (apply && (p.ldy & 7))) return EDTR_E_ALIGN;
    if (!aligned16(p.x) || (apply && !aligned16(p.y))) return EDTR_E_ALIGN;
    if (p.C > 8192) return EDTR_E_UNSUPPORTED;
    return EDTR_OK;
}

}  // namespace

extern "C" int edtr_gn_stats(const edtr_gn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_gn_params& p = *pp;
    if (int e = check_gn(p, false)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(p.sums, 0, sizeof(double) * 2 * p.groups * p.B, s);
    if (e != hipSuccess) return (int)e;
    const int CV = p.C >> 3;
    int R = CV >= 256 ? 1 : 256 / CV;
    if (R < 1) R = 1;
    int threads = CV >= 256 ? 256 : CV * R;
    threads = ((threads + 63) / 64) * 64;
    if (threads < 64) threads = 64;
    dim3 grid((unsigned)((p.HW + GN_PIX - 1) / GN_PIX), p.B);
    if (p.dtype == EDTR_BF16)
        hipLaunchKernelGGL(gn_stats_kernel<BF16>, grid, dim3(threads), 0, s, p, CV, R);
    else
        hipLaunchKernelGGL(gn_stats_kernel<F16>, grid, dim3(threads), 0, s, p, CV, R);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_gn_apply(const edtr_gn_params* pp, edtr_stream_t stream) {
    if (!pp) return EDTR_E_NULL;
    const edtr_gn_params& p = *pp;
    if (int e = check_gn(p, true)) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((p.HW + GN_PIX - 1) / GN_PIX), p.B);
    const size_t lds = sizeof(float) * 2 * p.C;
    if (p.dtype == EDTR_BF16)
        hipLaunchKernelGGL(gn_apply_kernel<BF16>, grid, dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL(gn_apply_kernel<F16>, grid, dim3(256), lds, s, p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_layernorm(int dtype, const void* x, int64_t rows, int C, int ldx, const float* gamma,
                              const float* beta, float eps, void* y, int ldy, edtr_stream_t stream) {
    if (!x || !y || !gamma || !beta) return EDTR_E_NULL;
    if (dtype != EDTR_BF16 && dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (rows <= 0 || C <= 0) return EDTR_E_SHAPE;
    if (C > 2048) return EDTR_E_UNSUPPORTED;
    if ((C & 7) || (ldx & 7) || (ldy & 7) || !aligned16(x) || !aligned16(y)) return EDTR_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == EDTR_BF16)
        hipLaunchKernelGGL(layernorm_kernel<BF16>, grid, dim3(256), 0, s, static_cast<const uint16_t*>(x), rows, C,
                           ldx, gamma, beta, eps, static_cast<uint16_t*>(y), ldy);
    else
        hipLaunchKernelGGL(layernorm_kernel<F16>, grid, dim3(256), 0, s, static_cast<const uint16_t*>(x), rows, C,
                           ldx, gamma, beta, eps, static_cast<uint16_t*>(y), ldy);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}

extern "C" int edtr_softmax_rows(int dtype, const float* sc, int64_t rows, int cols, int64_t ld_s, void* pr,
                                 int64_t ld_p, edtr_stream_t stream) {
    if (!sc || !pr) return EDTR_E_NULL;
    if (dtype != EDTR_BF16 && dtype != EDTR_F16) return EDTR_E_DTYPE;
    if (rows <= 0 || cols <= 0 || rows > 0x7fffffffLL) return EDTR_E_SHAPE;
    if ((cols & 3) || (ld_s & 3) || (ld_p & 3) || !aligned16(sc) || (reinterpret_cast<uintptr_t>(pr) & 7))
        return EDTR_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EDTR_BF16)
        hipLaunchKernelGGL(softmax_rows_kernel<BF16>, dim3((unsigned)rows), dim3(256), 0, s, sc, cols, ld_s,
                           static_cast<uint16_t*>(pr), ld_p);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<F16>, dim3((unsigned)rows), dim3(256), 0, s, sc, cols, ld_s,
                           static_cast<uint16_t*>(pr), ld_p);
    EDTR_LAUNCH_CHECK();
    return EDTR_OK;
}
