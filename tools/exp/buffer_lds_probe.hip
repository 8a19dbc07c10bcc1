// Hardware probe (run on the MI355X box): does `buffer_load_dwordx4 ... offen lds` (a) place lane i's 16 bytes at
// M0 + 16*i, (b) add the SGPR soffset to the address, (c) write ZEROS for lanes whose offset fails the range check?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* src, uint32_t* dst, uint32_t nbytes, uint32_t soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xABABABABu;
  __syncthreads();
  const uint64_t base = (uint64_t)src;
  u32x4 srd;
  srd.x = __builtin_amdgcn_readfirstlane((uint32_t)base);
  srd.y = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffff);
  srd.z = __builtin_amdgcn_readfirstlane(nbytes);
  srd.w = 0x00020000u;
  uint32_t voff = lane * 16;
  if (lane == 5 || lane == 40) voff = 0xfffffff0u;      // out of range
  if (lane == 7) voff = nbytes - 8;                      // straddles the end
  uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + 1024);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = reinterpret_cast<uint32_t*>(smem)[i];
}
int main() {
  const int n = 4096;
  std::vector<uint32_t> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0x1000000u + i;
  uint32_t *src, *dst;
  hipMalloc(&src, n * 4); hipMalloc(&dst, 4096);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  const uint32_t soff = 256;   // bytes
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, src, dst, (uint32_t)(n * 4), soff);
  std::vector<uint32_t> o(1024);
  hipMemcpy(o.data(), dst, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (o[i] != 0xABABABABu) { ++bad; }
  printf("first KiB untouched: %s\n", bad ? "NO" : "yes");
  for (int lane = 0; lane < 64; ++lane) {
    const uint32_t* c = &o[256 + lane * 4];
    uint32_t exp0 = 0x1000000u + (lane * 16 + soff) / 4;
    bool ok = c[0] == exp0 && c[3] == exp0 + 3;
    if (lane == 5 || lane == 40 || lane == 7 || !ok || lane < 2)
      printf("lane %2d: %08x %08x %08x %08x %s\n", lane, c[0], c[1], c[2], c[3], ok ? "(expected data)" : "");
  }
  for (int i = 512; i < 520; ++i) printf("%08x ", o[i]);
  printf("\n");
  return 0;
}
