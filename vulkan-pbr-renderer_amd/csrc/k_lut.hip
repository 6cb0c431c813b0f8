// k_lut.hip -- K1: split-sum BRDF integration map (gfx950).
//
// Replaces shaders/gen_brdf_integration_map.glsl:142-210 (Beckmann D :34-39, Mikkelsen G :57-59,
// Schlick Fc :196; uniform Fibonacci-hemisphere quadrature :170-179) as dispatched by
// render.cpp:607-613 (whose 6 z-slices all write the same texel; computed once here).
//
// The quadrature is numerically ill-conditioned at low roughness: the Beckmann lobe is a function of
// 1 - N.H, so a one-ulp change of H moves a sample's weight by percents.  To stay texel-for-texel
// comparable the kernel keeps the shader's operation order for everything that feeds N.H (Rotate,
// normalize: separately rounded fp32, correctly rounded divide / sqrt), takes sample angles and the
// per-column view angle from host tables, and evaluates acos/tan/exp/pow with the device libm.
// One thread per texel, sequential sample loop (same summation order as the shader).
#include "pbr_device.h"
#include "pbr_kernels.h"
#include <hip/hip_fp16.h>

// EXACT: Rotate() of gen_brdf_integration_map.glsl:61-64
__device__ __forceinline__ f3 rotate_exact(f3 v, f3 n, float c, float s) {
    float d = dot3(v, n);
    f3 a = scale3(sub3(v, scale3(n, d)), c);
    f3 b = scale3(cross3(n, v), s);
    f3 cc = scale3(n, d);
    return add3(add3(a, b), cc);
}

__device__ __forceinline__ float beckmann_dev(float ndoth, float m) {
    float m2 = m * m;
    float a = tanf(acosf(ndoth));
    float n2 = ndoth * ndoth;
    return expf(-(a * a) / m2) / (PBR_PI * m2 * n2 * n2);
}

__global__ __launch_bounds__(64) void k_brdf_lut(void* __restrict__ out, int fmt, int size, int nsamples,
                                                 const float4* __restrict__ angles, const float2* __restrict__ view_cs,
                                                 int y0, int rows) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * size) return;
    int x = id % size, y = y0 + id / size;
    float NdotV = ((float)x + 0.5f) / (float)size;           // :143, :154
    float rough = ((float)y + 0.5f) / (float)size;           // :155
    const f3 N = mk3(0.0f, 0.0f, 1.0f);
    const f3 X = mk3(1.0f, 0.0f, 0.0f);
    float2 vcs = view_cs[x];
    f3 V = rotate_exact(N, X, vcs.x, vcs.y);                 // :160
    float dw = 2 * PBR_PI / (float)nsamples;                 // :168
    float scale = 0.0f, bias = 0.0f;
    for (int i = 0; i < nsamples; ++i) {
        float4 a = angles[i];
        f3 L = rotate_exact(N, X, a.x, a.y);                 // :177
        L = rotate_exact(L, N, a.z, a.w);                    // :178
        f3 H = normalize3(add3(L, V));                       // :179
        float NdotL = dot3(N, L);
        float NdotH = dot3(N, H);
        float VdotH = dot3(V, H);
        float D = beckmann_dev(NdotH, rough);                // :192
        float G = fminf(1.0f, fminf(2.0f * NdotH * NdotV / VdotH, 2.0f * NdotH * NdotL / VdotH));   // :193
        float Fc = powf(1.0f - VdotH, 5.0f);                 // :196
        scale += D * G * (1 - Fc) * dw / (4.0f * NdotV);     // :198
        bias += D * G * (0 + Fc) * dw / (4.0f * NdotV);      // :199
    }
    size_t o = (size_t)y * size + x;
    if (fmt == PBRK_FMT_RG16F) {
        ((__half2*)out)[o] = __halves2half2(__float2half_rn(scale), __float2half_rn(bias));
    } else if (fmt == PBRK_FMT_RG32F) {
        ((float2*)out)[o] = make_float2(scale, bias);
    } else {
        ((float4*)out)[o] = make_float4(scale, bias, 0.0f, 1.0f);   // :209
    }
}

extern "C" int pbrk_brdf_lut(void* out, int out_format, int size, int nsamples, const void* angles4,
                             const void* view_cs, int y0, int y1, void* stream) {
    if (!out || !angles4 || !view_cs || size < 1 || nsamples < 1) return PBRK_E_ARG;
    if (y0 < 0 || y1 > size || y0 >= y1) return PBRK_E_ARG;
    if (out_format != PBRK_FMT_RG16F && out_format != PBRK_FMT_RG32F && out_format != PBRK_FMT_RGBA32F) return PBRK_E_FORMAT;
    int rows = y1 - y0;
    int total = rows * size;
    hipLaunchKernelGGL(k_brdf_lut, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       out, out_format, size, nsamples, (const float4*)angles4, (const float2*)view_cs, y0, rows);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
