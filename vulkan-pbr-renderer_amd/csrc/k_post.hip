// k_post.hip -- K8 / K9: the post-process tail that consumes the shade pass.
//   K8 k_taa_resolve        shaders/taa_resolve.glsl:180-287   (full-screen draw, render.cpp:1131-1137)
//   K9 k_final_post_process shaders/final_post_process.glsl:2-10,31-34 (render.cpp:1181-1187)
//
// Sampler (SAMPLER_LINEAR_CLAMP on 2-D textures): texel coordinates are snapped to 1/256 texel before the bilinear
// split -- Vulkan's subTexelPrecisionBits = 8, what the reference's target GPUs do -- then weights and lerps are exact
// fp32 (a + t*(b-a), x then y); edges clamp.  A tap aimed at a texel centre therefore returns that texel bit for bit,
// which is what lets K8 read its 3x3 neighbourhood, its velocity tap and 5 of the 9 Catmull-Rom history taps as plain
// texel fetches (kCentreExact, valid while the snap absorbs the fp32 coordinate error: extents <= 8192).
// Arithmetic follows the shader statement by statement (no contraction, correctly rounded divide / sqrt), so K8 is
// bit-identical to the oracle and, through Oracle-A, to the shader text executed on the CPU.
// Both kernels are HBM-shaped by bytes (K8: 36 B per pixel = 8 lighting + 4 depth + 4 + 4 velocity + 8 history + 8
// written; K9: 8 B read + 4 B written); K8 issues 31 vector loads per pixel and is bound by the L1 path.
#include "pbr_device.h"
#include "pbr_kernels.h"

#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float h2f(unsigned short h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float lerpx(float p, float q, float t) { return p + t * (q - p); }

struct Rgba { float x, y, z, w; };

__device__ __forceinline__ Rgba texel_rgba16f(const PbrkTex2D& t, int i, int j) {
    uint2 v = ((const uint2*)t.data)[(size_t)j * t.width + i];
    return Rgba{h2f(v.x & 0xffff), h2f(v.x >> 16), h2f(v.y & 0xffff), h2f(v.y >> 16)};
}
__device__ __forceinline__ float2 texel_rg16f(const PbrkTex2D& t, int i, int j) {
    unsigned v = ((const unsigned*)t.data)[(size_t)j * t.width + i];
    return make_float2(h2f(v & 0xffff), h2f(v >> 16));
}

// snapped bilinear split of one axis: returns the two clamped texel indices and the weight of the second
__device__ __forceinline__ void split_axis(float coord01, int extent, int& i0, int& i1, float& a) {
    float f = coord01 * (float)extent - 0.5f;
    f = floorf(f * 256.0f + 0.5f) * (1.0f / 256.0f);
    float fl = floorf(f);
    a = f - fl;
    int i = (int)fl;
    i0 = clampi(i, 0, extent - 1); i1 = clampi(i + 1, 0, extent - 1);
}

__device__ __forceinline__ Rgba sample_rgba16f(const PbrkTex2D& t, float u, float v) {
    int i0, i1, j0, j1; float a, b;
    split_axis(u, t.width, i0, i1, a); split_axis(v, t.height, j0, j1, b);
    Rgba t00 = texel_rgba16f(t, i0, j0), t10 = texel_rgba16f(t, i1, j0), t01 = texel_rgba16f(t, i0, j1), t11 = texel_rgba16f(t, i1, j1);
    return Rgba{lerpx(lerpx(t00.x, t10.x, a), lerpx(t01.x, t11.x, a), b), lerpx(lerpx(t00.y, t10.y, a), lerpx(t01.y, t11.y, a), b),
                lerpx(lerpx(t00.z, t10.z, a), lerpx(t01.z, t11.z, a), b), lerpx(lerpx(t00.w, t10.w, a), lerpx(t01.w, t11.w, a), b)};
}
__device__ __forceinline__ float2 sample_rg16f(const PbrkTex2D& t, float u, float v) {
    int i0, i1, j0, j1; float a, b;
    split_axis(u, t.width, i0, i1, a); split_axis(v, t.height, j0, j1, b);
    float2 t00 = texel_rg16f(t, i0, j0), t10 = texel_rg16f(t, i1, j0), t01 = texel_rg16f(t, i0, j1), t11 = texel_rg16f(t, i1, j1);
    return make_float2(lerpx(lerpx(t00.x, t10.x, a), lerpx(t01.x, t11.x, a), b), lerpx(lerpx(t00.y, t10.y, a), lerpx(t01.y, t11.y, a), b));
}
__device__ __forceinline__ float sample_r32f(const PbrkTex2D& t, float u, float v) {
    int i0, i1, j0, j1; float a, b;
    split_axis(u, t.width, i0, i1, a); split_axis(v, t.height, j0, j1, b);
    const float* p = (const float*)t.data;
    float t00 = p[(size_t)j0 * t.width + i0], t10 = p[(size_t)j0 * t.width + i1], t01 = p[(size_t)j1 * t.width + i0], t11 = p[(size_t)j1 * t.width + i1];
    return lerpx(lerpx(t00, t10, a), lerpx(t01, t11, a), b);
}

__device__ __forceinline__ float mitchell_netravali(float x) {            // taa_resolve.glsl:13-26
    const float B = 1.0f / 3.0f, C = 1.0f / 3.0f;
    float ax = fabsf(x);
    if (ax < 1.0f)
        return ((12.0f - 9.0f * B - 6.0f * C) * ax * ax * ax + (-18.0f + 12.0f * B + 6.0f * C) * ax * ax + (6.0f - 2.0f * B)) / 6.0f;
    else if (ax >= 1.0f && ax < 2.0f)
        return ((-B - 6.0f * C) * ax * ax * ax + (6.0f * B + 30.0f * C) * ax * ax + (-12.0f * B - 48.0f * C) * ax + (8.0f * B + 24.0f * C)) / 6.0f;
    return 0.0f;
}

// one axis of SampleHistoryTextureCatmullRom (:139-167): tap coordinates (already divided by the size) and weights
struct CrAxis { float p0, p12, p3, w0, w12, w3; };
__device__ __forceinline__ CrAxis catmull_rom_axis(float uv, float ts) {
    float sp = uv * ts;
    float tp1 = floorf(sp - 0.5f) + 0.5f;
    float f = sp - tp1;
    CrAxis r;
    r.w0 = f * (-0.5f + f * (1.0f - 0.5f * f));
    float w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
    float w2 = f * (0.5f + f * (2.0f - 1.5f * f));
    r.w3 = f * f * (-0.5f + 0.5f * f);
    r.w12 = w1 + w2;
    float offset12 = w2 / (w1 + w2);
    r.p0 = (tp1 - 1.0f) / ts; r.p3 = (tp1 + 2.0f) / ts; r.p12 = (tp1 + offset12) / ts;
    return r;
}

template <bool kCentreExact, bool kHalfOut>
__global__ __launch_bounds__(256) void k_taa_resolve(PbrkTaaArgs A) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = A.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= A.width || py >= A.y1) return;
    const PbrkTex2D& LR = A.lighting_result;
    const float tsx = (float)LR.width, tsy = (float)LR.height;                 // :189
    const float psx = 1.0f / tsx, psy = 1.0f / tsy;                             // :190
    const float uvx = ((float)px + 0.5f) * psx, uvy = ((float)py + 0.5f) * psy; // :192

    float tot[3] = {0, 0, 0}, wsum = 0.0f, m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
    float closest_depth = 10000.0f, cdu = 0.0f, cdv = 0.0f;
    int cdi = 0, cdj = 0;
    const float depth = kCentreExact ? ((const float*)A.gbuffer_depth.data)[(size_t)py * A.gbuffer_depth.width + px]
                                     : sample_r32f(A.gbuffer_depth, uvx, uvy);  // :221 (same tap nine times)
#pragma unroll
    for (int x = -1; x <= 1; ++x)                                               // :205-227
#pragma unroll
        for (int y = -1; y <= 1; ++y) {
            float su = uvx + (float)x * psx, sv = uvy + (float)y * psy;
            Rgba nb = kCentreExact ? texel_rgba16f(LR, clampi(px + x, 0, LR.width - 1), clampi(py + y, 0, LR.height - 1))
                                   : sample_rgba16f(LR, su, sv);
            float w = mitchell_netravali(sqrtf((float)x * (float)x + (float)y * (float)y));
            float n[3] = {nb.x, nb.y, nb.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tot[k] = tot[k] + n[k] * w;
                m1[k] = m1[k] + n[k];
                m2[k] = m2[k] + n[k] * n[k];
            }
            wsum = wsum + w;
            if (depth < closest_depth) { closest_depth = depth; cdu = su; cdv = sv; cdi = px + x; cdj = py + y; }
        }
    float src[3] = {tot[0] / wsum, tot[1] / wsum, tot[2] / wsum};               // :228
    float2 vel;
    if (kCentreExact && closest_depth < 10000.0f)
        vel = texel_rg16f(A.gbuffer_velocity, clampi(cdi, 0, A.gbuffer_velocity.width - 1), clampi(cdj, 0, A.gbuffer_velocity.height - 1));
    else
        vel = sample_rg16f(A.gbuffer_velocity, cdu, cdv);                       // :230
    const float ru = uvx - vel.x * 0.5f, rv = uvy - vel.y * 0.5f;               // :231
    const float2 pvel = sample_rg16f(A.gbuffer_velocity_prev, ru, rv);          // :232

    // :234 history, 9 bilinear taps in the shader's order; with kCentreExact the p0 / p3 taps are single texels
    const PbrkTex2D& HI = A.prev_frame_result;
    CrAxis cx = catmull_rom_axis(ru, tsx), cy = catmull_rom_axis(rv, tsy);
    float prev[3] = {0, 0, 0};
    {
        const float pxs[3] = {cx.p0, cx.p12, cx.p3}, wxs[3] = {cx.w0, cx.w12, cx.w3};
        const float pys[3] = {cy.p0, cy.p12, cy.p3}, wys[3] = {cy.w0, cy.w12, cy.w3};
        int ci0[3], ci1[3], cj0[3], cj1[3]; float ca[3], cb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { split_axis(pxs[k], HI.width, ci0[k], ci1[k], ca[k]); split_axis(pys[k], HI.height, cj0[k], cj1[k], cb[k]); }
#pragma unroll
        for (int row = 0; row < 3; ++row)
#pragma unroll
            for (int col = 0; col < 3; ++col) {
                const bool x_exact = kCentreExact && col != 1, y_exact = kCentreExact && row != 1;
                Rgba t00 = texel_rgba16f(HI, ci0[col], cj0[row]);
                Rgba top = t00, bot;
                if (!x_exact) { Rgba t10 = texel_rgba16f(HI, ci1[col], cj0[row]); top = Rgba{lerpx(t00.x, t10.x, ca[col]), lerpx(t00.y, t10.y, ca[col]), lerpx(t00.z, t10.z, ca[col]), 0.f}; }
                Rgba s = top;
                if (!y_exact) {
                    Rgba t01 = texel_rgba16f(HI, ci0[col], cj1[row]);
                    bot = t01;
                    if (!x_exact) { Rgba t11 = texel_rgba16f(HI, ci1[col], cj1[row]); bot = Rgba{lerpx(t01.x, t11.x, ca[col]), lerpx(t01.y, t11.y, ca[col]), lerpx(t01.z, t11.z, ca[col]), 0.f}; }
                    s = Rgba{lerpx(top.x, bot.x, cb[row]), lerpx(top.y, bot.y, cb[row]), lerpx(top.z, bot.z, cb[row]), 0.f};
                }
                prev[0] = prev[0] + (s.x * wxs[col]) * wys[row];
                prev[1] = prev[1] + (s.y * wxs[col]) * wys[row];
                prev[2] = prev[2] + (s.z * wxs[col]) * wys[row];
            }
    }
    const float inv9 = 1.0f / 9.0f;                                             // :237
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                               // :239-244 (gamma = 1)
        float avg = m1[k] * inv9;
        float sigma = sqrtf(fabsf(m2[k] * inv9 - avg * avg));
        float minc = avg - sigma, maxc = avg + sigma;
        prev[k] = fminf(fmaxf(prev[k], minc), maxc);
    }
    float wB = 0.05f, wA = 1.0f - wB;                                           // :252-253
    float dvx = pvel.x - vel.x, dvy = pvel.y - vel.y;
    wB = wB + 1000.0f * sqrtf(dvx * dvx + dvy * dvy);                           // :269-270
    if (ru != fminf(fmaxf(ru, 0.0f), 1.0f) || rv != fminf(fmaxf(rv, 0.0f), 1.0f)) { wA = 0.0f; wB = 1.0f; }   // :272-275
    const float den = fmaxf(wB + wA, 0.00001f);
    float r0 = (src[0] * wB + prev[0] * wA) / den, r1 = (src[1] * wB + prev[1] * wA) / den, r2 = (src[2] * wB + prev[2] * wA) / den;   // :277
    const size_t o = (size_t)py * A.width + px;
    if (kHalfOut) {
        uint2 v;
        v.x = (unsigned)__half_as_ushort(__float2half_rn(r0)) | ((unsigned)__half_as_ushort(__float2half_rn(r1)) << 16);
        v.y = (unsigned)__half_as_ushort(__float2half_rn(r2)) | (0x3c00u << 16);
        ((uint2*)A.out)[o] = v;
    } else {
        ((float4*)A.out)[o] = make_float4(r0, r1, r2, 1.0f);                    // :290
    }
}

__device__ __forceinline__ float aces_approx(float v) {                   // final_post_process.glsl:2-10
    v = v * 0.6f;
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(fmaxf((v * (a * v + b)) / (v * (c * v + d) + e), 0.0f), 1.0f);
}
__device__ __forceinline__ unsigned to_unorm8(float v) {                   // render-target conversion, round to nearest even
    return (unsigned)__float2int_rn(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f);
}

template <int kOutFmt>
__global__ __launch_bounds__(256) void k_final_post_process(PbrkFinalArgs A) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = A.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= A.width || py >= A.y1) return;
    Rgba s;
    if (A.src.width == A.width && A.src.height == A.height && A.width <= 8192 && A.height <= 8192)
        s = texel_rgba16f(A.src, px, py);                                       // centre tap of an equal-size source
    else
        s = sample_rgba16f(A.src, ((float)px + 0.5f) / (float)A.width, ((float)py + 0.5f) / (float)A.height);
    const float g = 1.0f / 2.2f;
    float r0 = powf(aces_approx(2.0f * s.x), g), r1 = powf(aces_approx(2.0f * s.y), g), r2 = powf(aces_approx(2.0f * s.z), g);   // :32-33
    const size_t o = (size_t)py * A.width + px;
    if (kOutFmt == PBRK_FMT_RGBA8UN) ((unsigned*)A.out)[o] = to_unorm8(r0) | (to_unorm8(r1) << 8) | (to_unorm8(r2) << 16) | 0xff000000u;
    else if (kOutFmt == PBRK_FMT_BGRA8UN) ((unsigned*)A.out)[o] = to_unorm8(r2) | (to_unorm8(r1) << 8) | (to_unorm8(r0) << 16) | 0xff000000u;
    else if (kOutFmt == PBRK_FMT_RGBA16F) {
        uint2 v;
        v.x = (unsigned)__half_as_ushort(__float2half_rn(r0)) | ((unsigned)__half_as_ushort(__float2half_rn(r1)) << 16);
        v.y = (unsigned)__half_as_ushort(__float2half_rn(r2)) | (0x3c00u << 16);
        ((uint2*)A.out)[o] = v;
    } else ((float4*)A.out)[o] = make_float4(r0, r1, r2, 1.0f);
}

bool tex_ok(const PbrkTex2D& t, int fmt) { return t.data && t.format == fmt && t.width > 0 && t.height > 0; }
}  // namespace

extern "C" int pbrk_taa_resolve(const PbrkTaaArgs* a, void* stream) {
    if (!a || !a->out || a->width < 1 || a->height < 1 || a->y0 < 0 || a->y0 >= a->y1 || a->y1 > a->height) return PBRK_E_ARG;
    if (!tex_ok(a->lighting_result, PBRK_FMT_RGBA16F) || !tex_ok(a->prev_frame_result, PBRK_FMT_RGBA16F) || !tex_ok(a->gbuffer_depth, PBRK_FMT_R32F) ||
        !tex_ok(a->gbuffer_velocity, PBRK_FMT_RG16F) || !tex_ok(a->gbuffer_velocity_prev, PBRK_FMT_RG16F)) return PBRK_E_FORMAT;
    if (a->out_format != PBRK_FMT_RGBA16F && a->out_format != PBRK_FMT_RGBA32F) return PBRK_E_FORMAT;
    // the frame, its G-buffer planes and the render target share one size in the reference (render.cpp:680-697, 732-739)
    const PbrkTex2D* same[4] = {&a->lighting_result, &a->gbuffer_depth, &a->gbuffer_velocity, &a->gbuffer_velocity_prev};
    for (int k = 0; k < 4; ++k) if (same[k]->width != a->width || same[k]->height != a->height) return PBRK_E_ARG;
    if (a->out == a->prev_frame_result.data || a->out == a->lighting_result.data) return PBRK_E_ARG;
    const bool centre = a->width <= 8192 && a->height <= 8192;
    const bool half = a->out_format == PBRK_FMT_RGBA16F;
    dim3 grid((a->width + 63) / 64, (a->y1 - a->y0 + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (centre && half) hipLaunchKernelGGL((k_taa_resolve<true, true>), grid, block, 0, st, *a);
    else if (centre) hipLaunchKernelGGL((k_taa_resolve<true, false>), grid, block, 0, st, *a);
    else if (half) hipLaunchKernelGGL((k_taa_resolve<false, true>), grid, block, 0, st, *a);
    else hipLaunchKernelGGL((k_taa_resolve<false, false>), grid, block, 0, st, *a);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" int pbrk_final_post_process(const PbrkFinalArgs* a, void* stream) {
    if (!a || !a->out || a->width < 1 || a->height < 1 || a->y0 < 0 || a->y0 >= a->y1 || a->y1 > a->height) return PBRK_E_ARG;
    if (!tex_ok(a->src, PBRK_FMT_RGBA16F)) return PBRK_E_FORMAT;
    if (a->out == a->src.data) return PBRK_E_ARG;
    dim3 grid((a->width + 63) / 64, (a->y1 - a->y0 + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (a->out_format) {
    case PBRK_FMT_RGBA8UN: hipLaunchKernelGGL((k_final_post_process<PBRK_FMT_RGBA8UN>), grid, block, 0, st, *a); break;
    case PBRK_FMT_BGRA8UN: hipLaunchKernelGGL((k_final_post_process<PBRK_FMT_BGRA8UN>), grid, block, 0, st, *a); break;
    case PBRK_FMT_RGBA16F: hipLaunchKernelGGL((k_final_post_process<PBRK_FMT_RGBA16F>), grid, block, 0, st, *a); break;
    case PBRK_FMT_RGBA32F: hipLaunchKernelGGL((k_final_post_process<PBRK_FMT_RGBA32F>), grid, block, 0, st, *a); break;
    default: return PBRK_E_FORMAT;
    }
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
