// Hardware probe (run on the MI355X box): what does ONE `buffer_load_dwordx4 ... offen lds` (1 KiB per wave) cost
// the issuing wave, (a) with an M0 write + s_nop before each, (b) with one M0 write per 4 pieces and the
// instruction's 12-bit `offset:` moving the LDS destination (the offset also moves the global address, so the
// per-lane voffset is pre-compensated), and does variant (b) land the bytes where expected?
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/dma_issue_probe.hip -o tools/exp/_build/dma_issue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint32_t* src, uint64_t* cyc, uint32_t* dump, int iters, int waves_active) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t base = (uint64_t)src;
  u32x4 srd;
  srd.x = __builtin_amdgcn_readfirstlane((uint32_t)base);
  srd.y = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffff);
  srd.z = 0xFFFF0000u;
  srd.w = 0x00020000u;
  u32x4 srdB = srd;
  {
    const uint64_t b2 = base - 4096;
    srdB.x = __builtin_amdgcn_readfirstlane((uint32_t)b2);
    srdB.y = __builtin_amdgcn_readfirstlane((uint32_t)(b2 >> 32) & 0xffff);
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem)) + wave * 8192;
  // lane -> (row = lane>>3, 16-byte slot = lane&7) of an 8-row x 128-byte piece; rows 640 bytes apart in global memory
  const uint32_t vbase = (uint32_t)((blockIdx.x * 64 + wave * 16) * 640) + (lane >> 3) * 640 + (lane & 7) * 16;
  uint64_t total = 0;
  if (wave < waves_active) {
    for (int it = 0; it < iters; ++it) {
      const uint32_t soff = (uint32_t)(it & 3) * 128;
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t vo = vbase + j * 8 * 640;
          const uint32_t ld = lds0 + j * 1024;
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" : : "v"(vo), "s"(srd), "s"(ld), "s"(soff) : "memory");
        }
      } else {
        // one M0 per 4 pieces; piece j lands at M0 + 1024*(j&3) through `offset:`; the offset also moves the global
        // address, so voffset is pre-compensated by -1024*(j&3) (+4096 against an SRD whose base is 4096 lower)
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = vbase + j * 8 * 640 - (j & 3) * 1024 + 4096;
        const uint32_t ldA = lds0, ldB = lds0 + 4096;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %0, %10 offen lds\n\t"
                     "buffer_load_dwordx4 %3, %0, %10 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %4, %0, %10 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %5, %0, %10 offen offset:3072 lds\n\t"
                     "s_mov_b32 m0, %11\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %6, %0, %10 offen lds\n\t"
                     "buffer_load_dwordx4 %7, %0, %10 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %8, %0, %10 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %9, %0, %10 offen offset:3072 lds"
                     : : "s"(srdB), "s"(ldA), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                         "s"(soff), "s"(ldB) : "memory");
      }
      total += __builtin_amdgcn_s_memtime() - t0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = total;
  if (blockIdx.x == 0 && dump) {
    for (int i = threadIdx.x; i < 2048; i += 256) dump[i] = reinterpret_cast<uint32_t*>(smem)[i];   // wave 0's 8 KiB
  }
}

int main() {
  const int nblk = 256, iters = 200;
  const size_t n = (size_t)(nblk * 64 + 64) * 640 / 4 + 4096;
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)i;
  uint32_t *src, *dump; uint64_t* cyc;
  hipMalloc(&src, n * 4); hipMalloc(&dump, 8192); hipMalloc(&cyc, nblk * 4 * 8);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> d0(2048), d1(2048);
  for (int waves = 1; waves <= 4; waves *= 2) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 32768, 0, src + 1024, cyc, dump, iters, waves);
        else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 32768, 0, src + 1024, cyc, dump, iters, waves);
        hipDeviceSynchronize();
      }
      std::vector<uint64_t> c(nblk * 4);
      hipMemcpy(c.data(), cyc, nblk * 4 * 8, hipMemcpyDeviceToHost);
      std::vector<double> per;
      for (int b = 0; b < nblk; ++b) per.push_back((double)c[b * 4] / iters / 8);
      std::sort(per.begin(), per.end());
      hipMemcpy(mode == 0 ? d0.data() : d1.data(), dump, 8192, hipMemcpyDeviceToHost);
      printf("waves/WG issuing %d  mode %d (%s): median %.1f cycles per 1-KiB piece (min %.1f max %.1f)\n", waves, mode,
             mode == 0 ? "M0 write per piece" : "M0 per 4 pieces + offset:", per[nblk / 2], per[0], per[nblk - 1]);
    }
  }
  int diff = 0;
  for (int i = 0; i < 2048; ++i) diff += d0[i] != d1[i];
  printf("LDS image of wave 0, mode 1 vs mode 0: %d differing dwords (0 = the offset: form lands the same bytes)\n", diff);
  printf("sample mode0: %08x %08x %08x | mode1: %08x %08x %08x (piece 1 first dwords at [256])  %08x vs %08x\n", d0[0], d0[4], d0[32], d1[0], d1[4], d1[32], d0[256], d1[256]);
  return 0;
}
