// What the attention tile loop can reach on gfx950 with NO memory at all: per 64-key tile and wave the
// flash_attn64_v3 loop issues 32 MFMAs (32x32x16, 1024 matrix-pipe cycles = the 100 % mark) with, per MFMA slot, 2 v_exp_f32, 2 v_add_f32
// and 1 v_cvt_pk_bf16_f32 (64 / 64 / 32 per tile: one exponential and one row-sum add per score, one conversion per pair).  This kernel
// issues exactly that stream — one wave per SIMD, 4 waves per workgroup, one workgroup per CU like the real kernel, operands in
// registers, no LDS, no global memory inside the loop — and reports cycles per MFMA slot and the pipe occupancy it implies.
//   hipcc --offload-arch=gfx950 -O3 -o attn_issue_bound tools/exp/attn_issue_bound.hip && ./attn_issue_bound
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) issue_kernel(const uint32_t* seed, float* sink, uint64_t* cycles, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    uint32_t r = seed[tid & 4095];
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) {       // random bf16 operands in [-1, 1): the clock under load depends on the data
        r = r * 1664525u + 1013904223u;
        a[i] = (__bf16)(((int)(r >> 8) & 0xffff) / 32768.0f - 1.0f);
        r = r * 1664525u + 1013904223u;
        b[i] = (__bf16)(((int)(r >> 8) & 0xffff) / 32768.0f - 1.0f);
    }
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    float x0 = -0.25f - (float)(tid & 7) * 0.01f, x1 = -0.5f, e0 = 0.0f, e1 = 0.0f, s0 = 0.0f, s1 = 0.0f;
    uint32_t pk = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {          // 32 MFMA slots per iteration = one 64-key tile of the real loop
#define SLOT(ACC)                                                                                                                \
            if constexpr (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));       \
            else if constexpr (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6" \
                                                       : "+v"(ACC), "=v"(e0), "=v"(e1) : "v"(a), "v"(b), "v"(x0), "v"(x1));     \
            else if constexpr (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %5, %6, %0\n\tv_exp_f32 %1, %7\n\tv_exp_f32 %2, %8\n\t" \
                                                       "v_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %2"                            \
                                                       : "+v"(ACC), "=&v"(e0), "=&v"(e1), "+v"(s0), "+v"(s1) : "v"(a), "v"(b), "v"(x0), "v"(x1)); \
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %1, %8\n\tv_exp_f32 %2, %9\n\t"              \
                              "v_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %2\n\tv_cvt_pk_bf16_f32 %5, %1, %2"                     \
                              : "+v"(ACC), "=&v"(e0), "=&v"(e1), "+v"(s0), "+v"(s1), "=v"(pk) : "v"(a), "v"(b), "v"(x0), "v"(x1));
            SLOT(acc0)
            SLOT(acc1)
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float out = s0 + s1 + e0 + e1 + __uint_as_float(pk & 0x3f800000u);
    for (int i = 0; i < 16; ++i) out += acc0[i] + acc1[i];
    if (out == 123.456f) sink[tid] = out;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
}

template <int MODE>
static void run(const char* what, const uint32_t* seed, float* sink, uint64_t* cyc_d, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(issue_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, seed, sink, cyc_d, 200);      // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(issue_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, seed, sink, cyc_d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> cyc(blocks * 4);
    hipMemcpy(cyc.data(), cyc_d, cyc.size() * 8, hipMemcpyDeviceToHost);
    std::sort(cyc.begin(), cyc.end());
    const double med = (double)cyc[cyc.size() / 2], per_slot = med / (iters * 32.0);
    const double tflops = 2.0 * 32 * 32 * 16 * 32.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-58s %7.1f cycles / MFMA slot  pipe occupancy %5.1f %%  %8.1f TFLOP/s  clock %.2f GHz (median wave cycles / wall)\n", what, per_slot,
           100.0 * 32.0 / per_slot, tflops, med / (ms * 1e-3) / 1e9);
}

int main() {
    const int blocks = 256;
    std::vector<uint32_t> h(4096);
    uint32_t r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = r; }
    uint32_t* seed;
    float* sink;
    uint64_t* cyc;
    hipMalloc(&seed, 4096 * 4);
    hipMalloc(&sink, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 4 * 8);
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("one wave per SIMD, 256 workgroups x 4 waves, registers only; 32 MFMA slots (32x32x16 bf16) = one 64-key tile of flash_attn64_v3\n");
    run<0>("MFMA alone", seed, sink, cyc, blocks);
    run<1>("MFMA + 2 v_exp_f32", seed, sink, cyc, blocks);
    run<2>("MFMA + 2 v_exp_f32 + 2 v_add_f32", seed, sink, cyc, blocks);
    run<3>("MFMA + 2 v_exp_f32 + 2 v_add_f32 + v_cvt_pk_bf16_f32 (the loop)", seed, sink, cyc, blocks);
    return 0;
}
