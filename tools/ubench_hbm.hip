// Achievable HBM rates for the roofline discussion (development tool): streaming write, read and copy with 16 B/lane.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o tools/ubench_hbm
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_write(float4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i); }
__global__ void k_read(const float4* p, size_t n, float* out) {
    float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) *out = s;
}
__global__ void k_copy(const float4* a, float4* b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
int main() {
    const size_t bytes = (size_t)1610612736, n = bytes / 16;
    float4 *a, *b; float* o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n);
                if (k == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o);
                if (k == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[k], e0, e1);
            }
        }
        printf("grid %6d: write %.2f TB/s | read %.2f TB/s | copy %.2f TB/s (read+write bytes)\n", grid, bytes / ms[0] / 1e9, bytes / ms[1] / 1e9, 2.0 * bytes / ms[2] / 1e9);
    }
    hipEventRecord(e0); hipMemsetAsync(a, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1); float m; hipEventElapsedTime(&m, e0, e1);
    printf("hipMemsetAsync: %.2f TB/s\n", bytes / m / 1e9);
    return 0;
}
