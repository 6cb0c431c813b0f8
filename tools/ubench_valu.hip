// micro-benchmark: issue cost of candidate inner-loop instructions on gfx950 (one number per op, cycles per wave-instruction per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int OP> __global__ void k(float* out, float a, float b) {
    float x = a + threadIdx.x * 1e-6f, y = b, z = 0.3f + threadIdx.x * 1e-7f, w = 0.9f;
    float2 p = make_float2(x, y), q = make_float2(z, w);
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if (OP == 0) { x = fmaf(x, y, z); y = fmaf(y, z, w); z = fmaf(z, w, x); w = fmaf(w, x, y); }
        if (OP == 1) { x = __builtin_amdgcn_cubeid(x, y, z); y = __builtin_amdgcn_cubesc(y, z, w); z = __builtin_amdgcn_cubetc(z, w, x); w = __builtin_amdgcn_cubema(w, x, y); }
        if (OP == 2) { x = __builtin_amdgcn_rcpf(x); y = __builtin_amdgcn_rcpf(y); z = __builtin_amdgcn_rcpf(z); w = __builtin_amdgcn_rcpf(w); }
        if (OP == 3) { x = __builtin_amdgcn_fractf(x); y = __builtin_amdgcn_fractf(y); z = __builtin_amdgcn_fractf(z); w = __builtin_amdgcn_fractf(w); }
        if (OP == 4) { x = (float)(int)x; y = (float)(int)y; z = (float)(int)z; w = (float)(int)w; }   // 2 cvt each
        if (OP == 5) { p.x = fmaf(p.x, q.x, q.y); p.y = fmaf(p.y, q.x, q.y); q.x = fmaf(q.x, p.x, p.y); q.y = fmaf(q.y, p.x, p.y); }   // pk candidates
        if (OP == 6) { x = floorf(x); y = floorf(y); z = floorf(z); w = floorf(w); }
        if (OP == 7) { int ix = __builtin_amdgcn_readlane(__float_as_int(x), i & 63); x = __int_as_float(ix) + y; int iy = __builtin_amdgcn_readlane(__float_as_int(y), (i + 1) & 63); y = __int_as_float(iy) + z; z += w; w += x; }
        if (OP >= 8) {   // integer multiplies (tap addresses): 32-bit, 32-bit multiply-add, 24-bit, 24-bit multiply-add
            int ix = __float_as_int(x), iy = __float_as_int(y), iz = __float_as_int(z), iw = __float_as_int(w);
            if (OP == 8) { ix = ix * iy; iy = iy * iz; iz = iz * iw; iw = iw * ix; }
            if (OP == 9) { ix = ix * iy + iz; iy = iy * iz + iw; iz = iz * iw + ix; iw = iw * ix + iy; }
            if (OP == 10) { ix = __mul24(ix, iy); iy = __mul24(iy, iz); iz = __mul24(iz, iw); iw = __mul24(iw, ix); }
            if (OP == 11) { ix = __mul24(ix, iy) + iz; iy = __mul24(iy, iz) + iw; iz = __mul24(iz, iw) + ix; iw = __mul24(iw, ix) + iy; }
            x = __int_as_float(ix); y = __int_as_float(iy); z = __int_as_float(iz); w = __int_as_float(iw);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w + p.x + p.y + q.x + q.y;
}
template <int OP> void run(const char* name, int ops_per_iter) {
    float* d; hipMalloc(&d, (size_t)256 * 2048 * 4 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        int blocks = 256 * waves_per_simd;               // 256 threads = 4 waves = 1 per SIMD per block; one block per CU per step
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 0.5f, 0.25f);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 0.5f, 0.25f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double wave_instr_per_simd = (double)N * ops_per_iter * waves_per_simd;
        printf("%-10s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
    }
    hipFree(d);
}
int main() {
    run<0>("fma", 4); run<1>("cube", 4); run<2>("rcp", 4); run<3>("fract", 4); run<4>("cvt2", 8); run<5>("fma_vec2", 4); run<6>("floor", 4); run<7>("readlane", 6);
    run<8>("mul_lo", 4); run<9>("mul_add32", 4); run<10>("mul_i24", 4); run<11>("mad_i24", 4);
    return 0;
}
