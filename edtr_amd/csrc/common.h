// Shared device-side helpers for libedtr_hip (gfx950 / CDNA4 only: wave64, MFMA 32x32x16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/edtr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct alignas(16) U4 { uint32_t x, y, z, w; };

// 16-bit storage traits: conversions + the matching MFMA.
struct BF16 {
    using vec8 = bf16x8_t;
    using elem = uint16_t;
    static constexpr uint32_t kOnePair = 0x3F803F80u;    // two 1.0 values
    static __device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float f) {
        __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
        return __builtin_bit_cast(uint16_t, b);
    }
    // two fp32 -> one dword of two bf16 (ONE v_cvt_pk_bf16_f32, RNE)
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    }
    static __device__ __forceinline__ f32x16 mfma(const U4& a, const U4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(const U4& a, const U4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
struct F16 {
    using vec8 = f16x8_t;
    using elem = uint16_t;
    static constexpr uint32_t kOnePair = 0x3C003C00u;
    static __device__ __forceinline__ float to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint16_t from_f32(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // ONE v_cvt_pk_f16_f32 (RNE)
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
    }
    static __device__ __forceinline__ f32x16 mfma(const U4& a, const U4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(const U4& a, const U4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// fp32 "storage" for the elementwise kernels' high-precision (EDTR_F32_SPLIT) variants
struct F32E {
    using elem = float;
    static __device__ __forceinline__ float from_f32(float f) { return f; }
};

template <typename T>
__device__ __forceinline__ void unpack8(const U4& v, float (&f)[8]) {
    f[0] = T::to_f32((uint16_t)(v.x & 0xffff)); f[1] = T::to_f32((uint16_t)(v.x >> 16));
    f[2] = T::to_f32((uint16_t)(v.y & 0xffff)); f[3] = T::to_f32((uint16_t)(v.y >> 16));
    f[4] = T::to_f32((uint16_t)(v.z & 0xffff)); f[5] = T::to_f32((uint16_t)(v.z >> 16));
    f[6] = T::to_f32((uint16_t)(v.w & 0xffff)); f[7] = T::to_f32((uint16_t)(v.w >> 16));
}
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return T::pack2(lo, hi); }
template <typename T>
__device__ __forceinline__ U4 pack8(const float (&f)[8]) {
    U4 v;
    v.x = pack2<T>(f[0], f[1]); v.y = pack2<T>(f[2], f[3]);
    v.z = pack2<T>(f[4], f[5]); v.w = pack2<T>(f[6], f[7]);
    return v;
}

__device__ __forceinline__ U4 ldg16(const void* p) { return *reinterpret_cast<const U4*>(p); }
__device__ __forceinline__ void stg16(void* p, const U4& v) { *reinterpret_cast<U4*>(p) = v; }
__device__ __forceinline__ U4 zero16() { U4 v; v.x = v.y = v.z = v.w = 0u; return v; }

// x * sigmoid(x); v_rcp (1 ulp) instead of the ~10-instruction IEEE division: GroupNorm-apply is VALU/HBM co-limited
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default
// erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, i.e. fp32-exact for a 16-bit result): one v_exp + one v_rcp + 6 FMAs
// instead of libdevice erff (~40 instructions) — the GEGLU epilogue applies it to every gate element.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = __builtin_fmaf(poly, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = 1.0f - poly * t * __expf(-z * z);     // erf(|x| / sqrt(2))
    return 0.5f * x + 0.5f * fabsf(x) * e;                 // 0.5 x (1 + sign(x) erf(|x|/sqrt2))
}

// gelu_erf_f over 8 values in LOCKSTEP: one stage of the evaluation for all eight before the next.  Left alone, hipcc evaluates
// value after value (least register pressure), and a wave with at most one partner on its SIMD — every MFMA kernel here —
// then runs the 13-deep dependency chain of each value at the VALU latency (measured in edtr_swin_mlp: 2.5k cycles for 16
// values, 11 cycles per instruction) instead of at the issue rate.  An empty asm that takes a stage's eight results as
// read-write operands pins the order (__builtin_amdgcn_sched_barrier does not: the arithmetic is moved across it before the
// scheduler runs).  HALF_IN: the caller passes h = x / 2 (folded into its affine constants) — one stage less.
__device__ __forceinline__ void pin8(float (&v)[8]) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
template <bool HALF_IN>
__device__ __forceinline__ void gelu_erf_lockstep(float (&h)[8]) {
    float d[8], u[8], poly[8];
    if constexpr (!HALF_IN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = 0.5f * h[i];
        pin8(h);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = __builtin_fmaf(fabsf(h[i]), 0.3275911f * 1.41421356237309504880f, 1.0f);
    pin8(d);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_rcpf(d[i]);                    // t = 1 / (1 + p |x| / sqrt 2)
    pin8(d);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = h[i] * h[i];
    pin8(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = u[i] * (-2.0f * 1.4426950408889634f);
    pin8(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_exp2f(u[i]);                   // exp(-x^2 / 2)
    pin8(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = __builtin_fmaf(1.061405429f, d[i], -1.453152027f);
    pin8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = __builtin_fmaf(poly[i], d[i], 1.421413741f);
    pin8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = __builtin_fmaf(poly[i], d[i], -0.284496736f);
    pin8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = __builtin_fmaf(poly[i], d[i], 0.254829592f);
    pin8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = u[i] * d[i];
    pin8(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = __builtin_fmaf(-poly[i], u[i], 1.0f);           // erf(|x| / sqrt 2)
    pin8(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = __builtin_fmaf(fabsf(h[i]), u[i], h[i]);        // x/2 (1 + sign(x) erf(|x| / sqrt 2))
    pin8(h);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Byte offset of 16-byte chunk `c` of row `r` in a [rows][64 x 16-bit] LDS tile (128-byte rows).
// XOR swizzle so that the 16 lanes of every ds_read_b128 lane group (rows r..r+15 pattern of the
// MFMA operand read, chunk fixed) hit 16 distinct 16-byte bank slots of the 256-byte bank row.
__device__ __forceinline__ int tile_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// One global_load_lds_dwordx4: lane i's 16 bytes land at LDS byte address lds_addr + 16*i (lds_addr wave-uniform, in
// M0).  Inline asm on purpose: hipcc neither counts it in its own vmcnt bookkeeping nor serialises later ds_reads
// behind it with a vmcnt(0), so the prefetch can stay in flight across the barrier; completion is waited for by
// the hand-placed counted s_waitcnt in the main loop.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_of(const char* p) {
    return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}

// Buffer-addressed LDS-DMA: address = SRD base + voff (per lane) + soff (wave uniform); lanes whose voff fails the
// descriptor's range check write zeros.
constexpr uint32_t kOobOffset = 0xFFFFFF00u;     // >= num_records - 15 -> always out of range
constexpr uint32_t kNumRecords = 0xFFFFFF00u;

__device__ __forceinline__ u32x4 make_srd(const void* base) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4 srd;
    srd.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    srd.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);   // stride 0 (raw buffer)
    srd.z = kNumRecords;
    srd.w = 0x00020000u;
    return srd;
}

__device__ __forceinline__ void dma16_buf(uint32_t voff, const u32x4& srd, uint32_t soff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :
                 : "v"(voff), "s"(srd), "s"(lds_addr), "s"(soff)
                 : "memory");   // M0 is free here: hipcc keeps no value in it across statements on gfx950
}

// slot stride (in channels) of edtr_igemm's fused GroupNorm partials: N, or the caller's wider buffer (gn_ld: the two halves of a
// concatenation share one buffer of slots)
__device__ __forceinline__ int gn_ld_of(const edtr_igemm_params& p) { return p.gn_ld > 0 ? p.gn_ld : p.N; }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// tile 17 of edtr_igemm lives in its own translation unit (halo512.hip)
bool edtr_halo512_ok(const edtr_igemm_params& p);
int edtr_launch_halo512(const edtr_igemm_params& p, hipStream_t stream);
bool edtr_halo512p_ok(const edtr_igemm_params& p);           // tile 21
int edtr_launch_halo512p(const edtr_igemm_params& p, hipStream_t stream);
bool edtr_halo160_ok(const edtr_igemm_params& p);            // tile 20
int edtr_launch_halo160(const edtr_igemm_params& p, hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: one flag per (launch site, device), not one per process (ADVICE r04:
// a process that drives a second GPU launched the large-LDS kernels without the attribute there).  A failure is an error, not a
// launch that fails later.
struct EdtrLdsOnce { bool done[32] = {}; };
static inline int edtr_lds_attr(const void* fn, int bytes, EdtrLdsOnce& once) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = -1;
    if (dev >= 0 && once.done[dev]) return EDTR_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return EDTR_E_UNSUPPORTED;
    if (dev >= 0) once.done[dev] = true;
    return EDTR_OK;
}
// CU count of the CURRENT device (persistent kernels size their grids with it)
static inline int edtr_cu_count() {
    static int cus[32] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

#define EDTR_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
