// k_equirect.hip -- K6 (an extension: the reference only loads pre-converted cube strips,
// asset_import.cpp:17-27): equirectangular (lat-long, 2:1) RGBA32F panorama -> level 0 of an RGBA32F cubemap.
//
// Convention (the reference's world is Z-up, utils/camera.h): for the direction d through a cube texel centre
//   u = atan2(d.y, d.x) / (2 pi) + 0.5      (longitude, wraps)
//   v = acos(d.z / |d|) / pi                (colatitude: v = 0 is +Z, clamps)
// and the panorama is sampled bilinearly at (u*w - 0.5, v*h - 0.5).  The two transcendentals are evaluated in
// fp64 so that host (oracle) and device agree on the sample position to the last fp32 bit: next to an HDR sun
// texel a 1e-7 coordinate error is a visible difference.  Streaming kernel: 16 B written per texel, the
// panorama is read through the caches.
#include "pbr_device.h"
#include "pbr_kernels.h"

__global__ __launch_bounds__(256) void k_equirect_to_cube(const float4* __restrict__ eq, int w, int h,
                                                          float4* __restrict__ out, int size) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = blockIdx.z;
    if (x >= size || y >= size) return;
    f3 d = face_texel_dir(f, x, y, size);
    const double inv_2pi = 0.15915494309189535, inv_pi = 0.3183098861837907;
    double len = sqrt((double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z);
    float u = (float)(atan2((double)d.y, (double)d.x) * inv_2pi + 0.5);
    float v = (float)(acos(fmin(fmax((double)d.z / len, -1.0), 1.0)) * inv_pi);
    float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float a = fx - flx, b = fy - fly;
    int i0 = (int)flx, j0 = (int)fly;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = ((i0 % w) + w) % w; i1 = ((i1 % w) + w) % w;                 // longitude wraps
    j0 = min(max(j0, 0), h - 1); j1 = min(max(j1, 0), h - 1);          // colatitude clamps
    float4 t00 = eq[(size_t)j0 * w + i0], t10 = eq[(size_t)j0 * w + i1];
    float4 t01 = eq[(size_t)j1 * w + i0], t11 = eq[(size_t)j1 * w + i1];
    float4 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
    r.z = lerp_fma(lerp_fma(t00.z, t10.z, a), lerp_fma(t01.z, t11.z, a), b);
    r.w = lerp_fma(lerp_fma(t00.w, t10.w, a), lerp_fma(t01.w, t11.w, a), b);
    out[((size_t)f * size + y) * size + x] = r;
}

extern "C" int pbrk_equirect_to_cube(const void* equirect_rgba32f, int w, int h, void* cube_level0, int size, void* stream) {
    if (!equirect_rgba32f || !cube_level0 || w < 1 || h < 1 || size < 1) return PBRK_E_ARG;
    hipLaunchKernelGGL(k_equirect_to_cube, dim3((size + 63) / 64, (size + 3) / 4, 6), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)equirect_rgba32f, w, h, (float4*)cube_level0, size);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
