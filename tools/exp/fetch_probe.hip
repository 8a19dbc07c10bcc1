// Round 4: what does rocprofv3's FETCH_SIZE count on gfx950 — HBM traffic, or every L2 miss incl. the ones the
// 256 MiB Infinity Cache serves?  A streaming read of a buffer (16 B per lane, fully coalesced) is launched four times back to back
// for buffer sizes below and above the Infinity Cache; the harness prints the wall time of every launch, and the same binary under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_probe
// gives FETCH_SIZE per dispatch.  If the 2nd..4th read of a 64 MiB buffer (resident on-die after the first) still reports the
// whole buffer, the counter sits on the L2's fabric side and Infinity-Cache hits are INCLUDED: a per-XCD re-fetch of a small operand
// (eight L2s each pulling the same 3 MB of weights) is then fabric traffic, not HBM traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/fetch_probe.hip -o /tmp/fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) read_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 v = src[i];
        acc ^= v;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1u;      // keeps the loads alive
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t MiB = 1 << 20;
    const size_t sizes[] = {32 * MiB, 64 * MiB, 128 * MiB, 192 * MiB, 384 * MiB, 1024 * MiB};
    unsigned* sink;
    CK(hipMalloc(&sink, 4));
    void* flush;
    CK(hipMalloc(&flush, 1024 * MiB));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (size_t sz : sizes) {
        void* buf;
        CK(hipMalloc(&buf, sz));
        CK(hipMemset(buf, 1, sz));
        CK(hipMemset(flush, 2, 1024 * MiB));            // push the buffer out of the Infinity Cache before the first read
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, static_cast<const u32x4*>(buf), sz / 16, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("buffer %5zu MiB  read %d: %8.1f us  %7.2f TB/s\n", sz / MiB, rep + 1, ms * 1e3, sz / (ms * 1e-3) / 1e12);
        }
        CK(hipFree(buf));
    }
    return 0;
}
