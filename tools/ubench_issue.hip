// What a SIMD of gfx950 really issues: cycles per VALU wave-instruction at 1..8 waves per SIMD, measured in shader clocks inside
// the kernel (s_memtime), for independent FMAs, dependent FMAs and FMAs mixed with ds_read_b128 --
// the instruction mix of the region kernel's sample loop.  Development tool ("VALU issue rate").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define ITERS 16384
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, float a, float b, int salt) {
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(a, b, a, b);
    __syncthreads();
    float x0 = a + threadIdx.x * 1e-6f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    int s0 = salt, s1 = salt + 1;
    unsigned addr = (threadIdx.x * 16) & 16383;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0 || MODE == 3 || MODE == 4) {      // 8 independent FMAs
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        }
        if (MODE == 1) {                                              // 8 FMAs in one dependent chain
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
            x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
        }
        if (MODE == 3 || MODE == 4) {                                 // + one ds_read_b128 (MODE 4: its result feeds an FMA)
            typedef float v4 __attribute__((ext_vector_type(4)));
            v4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr + (unsigned)(unsigned long long)lds));
            addr = (addr + 4096) & 16383;
            if (MODE == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x7 = fmaf(v.x, a, x7); }
            else asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(v) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1);
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char* name, int valu_per_iter) {
    float* d; unsigned long long* t;
    (void)hipMalloc(&d, (size_t)256 * 8 * 256 * 4); (void)hipMalloc(&t, (size_t)256 * 8 * 4 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, t, 0.999f, 0.001f, 3);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, t, 0.999f, 0.001f, 3);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        (void)hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[h.size() / 2];
        double per_simd = med / ((double)ITERS * valu_per_iter) / wps;      // the wave shares its SIMD with wps - 1 others
        printf("%-34s %d waves/SIMD: %7.3f ms, median %9.0f ticks per wave -> %.2f shader clocks per VALU instruction per SIMD (clock %.2f GHz by wall time)\n",
               name, wps, ms, med, per_simd, med / (ms * 1e-3) * 1e-9);
    }
    (void)hipFree(d); (void)hipFree(t);
}
int main() {
    run<0>("8 independent FMAs", 8);
    run<1>("8 dependent FMAs", 8);
    run<3>("8 independent FMAs + ds_read_b128", 8);
    run<4>("8 FMAs + ds_read_b128 feeding one", 9);
    return 0;
}
