// Round 4: what does one dependent kernel node cost inside a hipGraph on this stack (ROCm 7.2, MI355X)?  det512s50 replays 26.8k
// launches per pass (536 per denoise step and lane) of which most are 15 - 25 us GEMMs in a dependency chain: the node-to-node
// turnaround is the floor of every such launch.  Measures, by stream capture like the product (edtr_graph_begin / _end):
//   (a) a chain of N empty one-workgroup kernels, (b) the same chain of 256-workgroup kernels that run ~T us each,
//   (c) two such chains as parallel branches (fork / join), (d) the chain launched eagerly.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/graph_node_latency.hip -o /tmp/graph_node_latency && /tmp/graph_node_latency
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) spin_kernel(int cycles, unsigned* sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (cycles < 0) *sink = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int time_graph(hipGraphExec_t ge, hipStream_t s, int reps, float* ms_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(ms_out, e0, e1));
    *ms_out /= reps;
    return 0;
}

int main() {
    unsigned* sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t s, s2;
    CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    const int N = 1000;
    const int grids[] = {1, 256, 256, 256};
    const int cycles[] = {0, 0, 10000, 40000};      // clock64 ticks at 100 MHz: 0, 0, ~100 us?  (reported from the measurement itself)
    for (int v = 0; v < 4; ++v) {
        for (int lanes = 1; lanes <= 2; ++lanes) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            hipEvent_t fork, join;
            CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
            if (lanes == 2) { CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0)); }
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin_kernel, dim3(grids[v]), dim3(256), 0, s, cycles[v], sink);
                if (lanes == 2) hipLaunchKernelGGL(spin_kernel, dim3(grids[v]), dim3(256), 0, s2, cycles[v], sink);
            }
            if (lanes == 2) { CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0)); }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float ms = 0.0f;
            if (time_graph(ge, s, 5, &ms)) return 1;
            printf("graph: %d x %d-workgroup kernels of %d ticks per lane, %d lane(s): %8.2f us per node (per lane)\n", N, grids[v], cycles[v], lanes, ms * 1e3 / N);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grids[v]), dim3(256), 0, s, cycles[v], sink);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grids[v]), dim3(256), 0, s, cycles[v], sink);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager: %d x %d-workgroup kernels of %d ticks on one stream:          %8.2f us per launch\n", N, grids[v], cycles[v], ms * 1e3 / N);
    }
    return 0;
}
