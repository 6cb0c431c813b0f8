// Launch-bound sequences: N dependent small kernels on one stream vs the same sequence replayed as an instantiated hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_graph.hip -o tools/ubench_graph
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_small(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main() {
    const int n = 64 * 1024, chain = 25, reps = 200;
    float* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, d, n);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, d, n);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("stream launches : %.2f us per kernel (%d-kernel chain: %.1f us)\n", ms * 1e3 / (reps * chain), chain, ms * 1e3 / reps);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, d, n);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("hipGraph replay : %.2f us per kernel (%d-kernel chain: %.1f us)\n", ms * 1e3 / (reps * chain), chain, ms * 1e3 / reps);
    return 0;
}
