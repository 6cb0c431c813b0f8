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
// clamp to [0, hi], hi >= 0 and wave-uniform (a texture extent - 1): the median of (v, 0, hi) in one instruction -- the compare /
// select / min the plain expression compiles to is three, and the samplers below clamp two indices per axis and tap
__device__ __forceinline__ int clampi(int v, int /*lo == 0*/, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
}
__device__ __forceinline__ float lerpx(float p, float q, float t) { return p + t * (q - p); }
// The same value for a COMPILE-TIME fraction: when t is a power of two, t * (q - p) is exact (texels are fp16: no fp32 underflow within
// reach), so one FMA rounds exactly like the multiply-add pair; other fractions keep the sampler's two roundings.
__device__ __forceinline__ float lerpc(float p, float q, float t) { return (t == 0.5f || t == 0.25f) ? fmaf(t, q - p, p) : p + t * (q - p); }

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
    int i = (int)fminf(fmaxf(fl, -1.0f), (float)extent);       // clamp before the conversion: no saturation / overflow for wild coordinates
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

__device__ __forceinline__ uint2 pack_half4(float r0, float r1, float r2) {
    uint2 v;
    v.x = (unsigned)__half_as_ushort(__float2half_rn(r0)) | ((unsigned)__half_as_ushort(__float2half_rn(r1)) << 16);
    v.y = (unsigned)__half_as_ushort(__float2half_rn(r2)) | (0x3c00u << 16);
    return v;
}

__device__ __forceinline__ Rgba unpack_rgba16f(uint2 v) { return Rgba{h2f(v.x & 0xffff), h2f(v.x >> 16), h2f(v.y & 0xffff), h2f(v.y >> 16)}; }

// buffer-addressed texel fetches: one 32-bit offset per load instead of a 64-bit address (textures are < 2 GiB, checked on the host)
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tex_rsrc(const PbrkTex2D& t, int texel_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)t.data, 0, t.width * t.height * texel_bytes, 0x00020000);
}
__device__ __forceinline__ Rgba fetch_rgba16f(__amdgpu_buffer_rsrc_t rs, int w, int i, int j) {
    u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rs, (j * w + i) * 8, 0, 0);
    return Rgba{h2f(v.x & 0xffff), h2f(v.x >> 16), h2f(v.y & 0xffff), h2f(v.y >> 16)};
}
__device__ __forceinline__ float2 fetch_rg16f(__amdgpu_buffer_rsrc_t rs, int w, int i, int j) {
    unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, (j * w + i) * 4, 0, 0);
    return make_float2(h2f(v & 0xffff), h2f(v >> 16));
}

// kCentreExact (frames up to 8192^2): every tap the shader aims at a texel centre -- the 3x3 neighbourhood, depth, the
// velocity tap and the outer Catmull-Rom taps -- is one texel fetch, and the outer taps need neither their division by the
// texture size nor the bilinear split (their texel is floor(sp - 0.5) - 1 / + 2).  Otherwise every tap goes through the sampler.
struct TaaParams { PbrkTaaArgs a; float psx, psy; };       // 1 / width, 1 / height: the shader's divisions (:190), done once on the host
template <bool kCentreExact, bool kHalfOut>
__global__ __launch_bounds__(256) void k_taa_resolve(TaaParams P) {
    const PbrkTaaArgs& A = P.a;
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = A.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= A.width || py >= A.y1) return;
    const PbrkTex2D& LR = A.lighting_result;
    const PbrkTex2D& HI = A.prev_frame_result;
    const int W = LR.width, H = LR.height;
    const __amdgpu_buffer_rsrc_t rs_lr = tex_rsrc(LR, 8), rs_hi = tex_rsrc(HI, 8), rs_v = tex_rsrc(A.gbuffer_velocity, 4), rs_pv = tex_rsrc(A.gbuffer_velocity_prev, 4);
    const float tsx = (float)W, tsy = (float)H;                                 // :189
    const float psx = P.psx, psy = P.psy;                                       // :190 (1.0f / tsx, 1.0f / tsy: IEEE divisions, identical on the host)
    const float uvx = ((float)px + 0.5f) * psx, uvy = ((float)py + 0.5f) * psy; // :192

    float tot[3] = {0, 0, 0}, wsum = 0.0f, m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
    float closest_depth = 10000.0f, cdu = 0.0f, cdv = 0.0f;
    const float depth = kCentreExact ? ((const float*)A.gbuffer_depth.data)[(size_t)py * A.gbuffer_depth.width + px]
                                     : sample_r32f(A.gbuffer_depth, uvx, uvy);  // :221 (same tap nine times)
#pragma unroll
    for (int x = -1; x <= 1; ++x)                                               // :205-227
#pragma unroll
        for (int y = -1; y <= 1; ++y) {
            float su = uvx + (float)x * psx, sv = uvy + (float)y * psy;
            Rgba nb = kCentreExact ? fetch_rgba16f(rs_lr, W, clampi(px + x, 0, W - 1), clampi(py + y, 0, H - 1)) : sample_rgba16f(LR, su, sv);
            float w = mitchell_netravali(sqrtf((float)x * (float)x + (float)y * (float)y));
            float n[3] = {nb.x, nb.y, nb.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tot[k] = tot[k] + n[k] * w;
                m1[k] = m1[k] + n[k];
                m2[k] = m2[k] + n[k] * n[k];
            }
            wsum = wsum + w;
            if (depth < closest_depth) { closest_depth = depth; cdu = su; cdv = sv; }   // :222-225: only the first tap (-1,-1) can win
        }
    const SharedRcp rws = shared_rcp(wsum);                                      // one reciprocal, three exact quotients (pbr_device.h)
    float src[3] = {div_by(tot[0], rws), div_by(tot[1], rws), div_by(tot[2], rws)};   // :228
    float2 vel;
    if (kCentreExact && closest_depth < 10000.0f) vel = fetch_rg16f(rs_v, W, clampi(px - 1, 0, W - 1), clampi(py - 1, 0, H - 1));
    else vel = sample_rg16f(A.gbuffer_velocity, cdu, cdv);                      // :230
    const float ru = uvx - vel.x * 0.5f, rv = uvy - vel.y * 0.5f;               // :231
    float2 pvel;                                                                // :232
    {
        int i0, i1, j0, j1; float a, b;
        split_axis(ru, W, i0, i1, a); split_axis(rv, H, j0, j1, b);
        float2 t00 = fetch_rg16f(rs_pv, W, i0, j0), t10 = fetch_rg16f(rs_pv, W, i1, j0), t01 = fetch_rg16f(rs_pv, W, i0, j1), t11 = fetch_rg16f(rs_pv, W, i1, j1);
        pvel = make_float2(lerpx(lerpx(t00.x, t10.x, a), lerpx(t01.x, t11.x, a), b), lerpx(lerpx(t00.y, t10.y, a), lerpx(t01.y, t11.y, a), b));
    }

    // :234 history, 9 bilinear taps in the shader's order
    float prev[3] = {0, 0, 0};
    {
        const int HW = HI.width, HH = HI.height;
        float wxs[3], wys[3], ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
        int ci0[3], ci1[3], cj0[3], cj1[3];
        const float uvs[2] = {ru, rv}, tss[2] = {tsx, tsy};
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {                                  // :139-167
            int* i0 = axis ? cj0 : ci0; int* i1 = axis ? cj1 : ci1; float* wt = axis ? wys : wxs; float* frac = axis ? cb : ca;
            const int extent = axis ? HH : HW;
            float sp = uvs[axis] * tss[axis];
            float fl = floorf(sp - 0.5f);
            float tp1 = fl + 0.5f;
            float f = sp - tp1;
            wt[0] = f * (-0.5f + f * (1.0f - 0.5f * f));
            float w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
            float w2 = f * (0.5f + f * (2.0f - 1.5f * f));
            wt[2] = f * f * (-0.5f + 0.5f * f);
            wt[1] = w1 + w2;
            float offset12 = w2 / (w1 + w2);
            split_axis((tp1 + offset12) / tss[axis], extent, i0[1], i1[1], frac[1]);
            if (kCentreExact) {
                int t = (int)fl;
                i0[0] = i1[0] = clampi(t - 1, 0, extent - 1); i0[2] = i1[2] = clampi(t + 2, 0, extent - 1);
            } else {
                split_axis((tp1 - 1.0f) / tss[axis], extent, i0[0], i1[0], frac[0]);
                split_axis((tp1 + 2.0f) / tss[axis], extent, i0[2], i1[2], frac[2]);
            }
        }
#pragma unroll
        for (int row = 0; row < 3; ++row)
#pragma unroll
            for (int col = 0; col < 3; ++col) {
                const bool x_exact = kCentreExact && col != 1, y_exact = kCentreExact && row != 1;
                Rgba t00 = fetch_rgba16f(rs_hi, HW, ci0[col], cj0[row]);
                Rgba top = t00;
                if (!x_exact) { Rgba t10 = fetch_rgba16f(rs_hi, HW, ci1[col], cj0[row]); top = Rgba{lerpx(t00.x, t10.x, ca[col]), lerpx(t00.y, t10.y, ca[col]), lerpx(t00.z, t10.z, ca[col]), 0.f}; }
                Rgba s = top;
                if (!y_exact) {
                    Rgba t01 = fetch_rgba16f(rs_hi, HW, ci0[col], cj1[row]);
                    Rgba bot = t01;
                    if (!x_exact) { Rgba t11 = fetch_rgba16f(rs_hi, HW, ci1[col], cj1[row]); bot = Rgba{lerpx(t01.x, t11.x, ca[col]), lerpx(t01.y, t11.y, ca[col]), lerpx(t01.z, t11.z, ca[col]), 0.f}; }
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
    // true divisions: `den` is data (a shared-reciprocal quotient would be 1 ulp off whenever its significand is all ones:
    // a handful of pixels per 1080p frame, and this kernel is bit-exact)
    float r0 = (src[0] * wB + prev[0] * wA) / den, r1 = (src[1] * wB + prev[1] * wA) / den, r2 = (src[2] * wB + prev[2] * wA) / den;   // :277
    const size_t o = (size_t)py * A.width + px;
    if (kHalfOut) ((uint2*)A.out)[o] = pack_half4(r0, r1, r2);
    else ((float4*)A.out)[o] = make_float4(r0, r1, r2, 1.0f);                   // :290
}

__device__ __forceinline__ float aces_approx(float v) {                   // final_post_process.glsl:2-10
    v = v * 0.6f;
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(fmaxf((v * (a * v + b)) / (v * (c * v + d) + e), 0.0f), 1.0f);
}
__device__ __forceinline__ unsigned to_unorm8(float v) {                   // render-target conversion, round to nearest even
    return (unsigned)__float2int_rn(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f);
}

// pow(x, 1/2.2) for x in [0,1] through the hardware log2 / exp2 (1 ulp each): relative error < 1e-6, far inside the
// 8-bit target's step and the 1e-5 the float targets are checked to; libm's powf costs ~10x more instructions.
__device__ __forceinline__ float pow_gamma(float x) {
    return __builtin_amdgcn_exp2f((1.0f / 2.2f) * __builtin_amdgcn_logf(x));
}
__device__ __forceinline__ void tone_map(const Rgba& s, float& r0, float& r1, float& r2) {     // :32-33
    r0 = pow_gamma(aces_approx(2.0f * s.x)); r1 = pow_gamma(aces_approx(2.0f * s.y)); r2 = pow_gamma(aces_approx(2.0f * s.z));
}
template <int kOutFmt> __device__ __forceinline__ unsigned pack8(float r0, float r1, float r2) {
    return kOutFmt == PBRK_FMT_RGBA8UN ? (to_unorm8(r0) | (to_unorm8(r1) << 8) | (to_unorm8(r2) << 16) | 0xff000000u)
                                       : (to_unorm8(r2) | (to_unorm8(r1) << 8) | (to_unorm8(r0) << 16) | 0xff000000u);
}

// general form: any source size (bilinear), one pixel per thread
template <int kOutFmt>
__global__ __launch_bounds__(256) void k_final_post_process(PbrkFinalArgs A) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = A.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= A.width || py >= A.y1) return;
    Rgba s = sample_rgba16f(A.src, ((float)px + 0.5f) / (float)A.width, ((float)py + 0.5f) / (float)A.height);
    float r0, r1, r2;
    tone_map(s, r0, r1, r2);
    const size_t o = (size_t)py * A.width + px;
    if (kOutFmt == PBRK_FMT_RGBA8UN || kOutFmt == PBRK_FMT_BGRA8UN) ((unsigned*)A.out)[o] = pack8<kOutFmt>(r0, r1, r2);
    else if (kOutFmt == PBRK_FMT_RGBA16F) ((uint2*)A.out)[o] = pack_half4(r0, r1, r2);
    else ((float4*)A.out)[o] = make_float4(r0, r1, r2, 1.0f);
}

// the reference's case: source and target have one size, every tap is a texel centre -> a streaming map, 4 pixels
// (32 B in, 16 B out for the 8-bit targets) per thread; width % 4 == 0
template <int kOutFmt>
__global__ __launch_bounds__(256) void k_final_post_process_stream(PbrkFinalArgs A) {
    const int quads_per_row = A.width >> 2;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nq = (size_t)quads_per_row * (A.y1 - A.y0);
    if (q >= nq) return;
    const size_t first = (size_t)A.y0 * A.width + q * 4;                        // rows are contiguous: a flat range of pixels
    const uint4* src = (const uint4*)((const uint2*)A.src.data + first);
    uint4 v0 = src[0], v1 = src[1];
    const unsigned raw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float r[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        Rgba s{h2f(raw[2 * k] & 0xffff), h2f(raw[2 * k] >> 16), h2f(raw[2 * k + 1] & 0xffff), 1.0f};
        tone_map(s, r[k][0], r[k][1], r[k][2]);
    }
    if (kOutFmt == PBRK_FMT_RGBA8UN || kOutFmt == PBRK_FMT_BGRA8UN) {
        ((uint4*)((unsigned*)A.out + first))[0] = make_uint4(pack8<kOutFmt>(r[0][0], r[0][1], r[0][2]), pack8<kOutFmt>(r[1][0], r[1][1], r[1][2]),
                                                             pack8<kOutFmt>(r[2][0], r[2][1], r[2][2]), pack8<kOutFmt>(r[3][0], r[3][1], r[3][2]));
    } else if (kOutFmt == PBRK_FMT_RGBA16F) {
        uint2 a = pack_half4(r[0][0], r[0][1], r[0][2]), b = pack_half4(r[1][0], r[1][1], r[1][2]);
        uint2 c = pack_half4(r[2][0], r[2][1], r[2][2]), d = pack_half4(r[3][0], r[3][1], r[3][2]);
        uint4* o = (uint4*)((uint2*)A.out + first);
        o[0] = make_uint4(a.x, a.y, b.x, b.y); o[1] = make_uint4(c.x, c.y, d.x, d.y);
    } else {
        float4* o = (float4*)A.out + first;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = make_float4(r[k][0], r[k][1], r[k][2], 1.0f);
    }
}

// K10 / K11: bloom passes.  Column / row tap coordinates repeat (5 distinct offsets for the 13-tap downsample, 3 for the
// tent), so the snapped bilinear split is done once per distinct offset and the taps index the results.
struct BloomParams { PbrkBloomArgs a; float x_step, y_step, rcp_dw, rcp_dh; int exact2to1; };
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// The general path of a bloom pass: every tap through the snapped bilinear sampler (any size ratio, border pixels of the exact passes).
template <bool kUp>
__device__ __forceinline__ void bloom_general(const BloomParams& P, const __amdgpu_buffer_rsrc_t rs, const int px, const int py, float* r) {
    const PbrkBloomArgs& A = P.a;
    const int SW = A.src.width, SH = A.src.height;
    SharedRcp rw, rh;
    rw.d = (float)A.dst_width; rw.r = P.rcp_dw; rh.d = (float)A.dst_height; rh.r = P.rcp_dh;
    const float u = div_by((float)px + 0.5f, rw), v = div_by((float)py + 0.5f, rh);       // fs_uv (full-screen triangle)
    constexpr int kOffs = kUp ? 3 : 5;                                                  // offsets -1..1 (x radius) or -2..2 (x texel)
    int ci0[kOffs], ci1[kOffs], cj0[kOffs], cj1[kOffs]; float ca[kOffs], cb[kOffs];
#pragma unroll
    for (int k = 0; k < kOffs; ++k) {
        const float off = (float)(k - kOffs / 2);
        split_axis(u + off * P.x_step, SW, ci0[k], ci1[k], ca[k]);
        split_axis(v + off * P.y_step, SH, cj0[k], cj1[k], cb[k]);
    }
    auto tap = [&](int kx, int ky, float* o) {
        Rgba t00 = fetch_rgba16f(rs, SW, ci0[kx], cj0[ky]), t10 = fetch_rgba16f(rs, SW, ci1[kx], cj0[ky]);
        Rgba t01 = fetch_rgba16f(rs, SW, ci0[kx], cj1[ky]), t11 = fetch_rgba16f(rs, SW, ci1[kx], cj1[ky]);
        o[0] = lerpx(lerpx(t00.x, t10.x, ca[kx]), lerpx(t01.x, t11.x, ca[kx]), cb[ky]);
        o[1] = lerpx(lerpx(t00.y, t10.y, ca[kx]), lerpx(t01.y, t11.y, ca[kx]), cb[ky]);
        o[2] = lerpx(lerpx(t00.z, t10.z, ca[kx]), lerpx(t01.z, t11.z, ca[kx]), cb[ky]);
    };
    if (kUp) {                                                                          // bloom_upsample.glsl:43-58
        float t[9][3];
#pragma unroll
        for (int k = 0; k < 9; ++k) tap(k % 3, k / 3, t[k]);                            // a b c / d e f / g h i
        const float factor = A.dst_mip_level == 0 ? 0.06f : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sum = t[4][c] * 4.0f;
            sum = sum + (((t[1][c] + t[3][c]) + t[5][c]) + t[7][c]) * 2.0f;
            sum = sum + (((t[0][c] + t[2][c]) + t[6][c]) + t[8][c]);
            r[c] = sum * factor / 16.0f;
        }
    } else {                                                                            // bloom_downsample.glsl:50-97
        float t[13][3];
#pragma unroll
        for (int k = 0; k < 9; ++k) tap(2 * (k % 3), 2 * (k / 3), t[k]);                // a..i at offsets -2, 0, 2
        tap(1, 1, t[9]); tap(3, 1, t[10]); tap(1, 3, t[11]); tap(3, 3, t[12]);          // j k l m at (+-1, +-1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sum = t[4][c] * 0.125f;
            sum = sum + (((t[0][c] + t[2][c]) + t[6][c]) + t[8][c]) * 0.03125f;
            sum = sum + (((t[1][c] + t[3][c]) + t[5][c]) + t[7][c]) * 0.0625f;
            sum = sum + (((t[9][c] + t[10][c]) + t[11][c]) + t[12][c]) * 0.125f;
            if (A.dst_mip_level == 1) sum = fminf(sum, 1.0f);
            r[c] = sum;
        }
    }
}
template <bool kUp>
__global__ __launch_bounds__(256) void k_bloom_pass(BloomParams P) {
    const PbrkBloomArgs& A = P.a;
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = A.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= A.dst_width || py >= A.y1) return;
    const int SW = A.src.width, SH = A.src.height;
    const __amdgpu_buffer_rsrc_t rs = tex_rsrc(A.src, 8);
    float r[3];
    bool done = false;
    // Exact 2:1 passes (every level of an even-sized chain; the two full-size passes of a 1080p frame): the snapped tap
    // coordinates are known in closed form -- texel pairs at fraction 1/2 (down) or 1/4, 3/4 by pixel parity (up) -- so an
    // interior pixel reads its 6x6 / 5x5 source window as whole rows (18 / 15 loads instead of 52 / 36) and needs no
    // coordinate arithmetic.  Same lerps in the same order: bit-identical to the general path below (border pixels take it).
    if (P.exact2to1) {
        if (!kUp) {
            if (px >= 1 && px <= A.dst_width - 2 && py >= 1 && py <= A.dst_height - 2) {
                float hx[6][5][3];                                                      // rows x horizontal taps (offsets -2..2) x rgb
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) {
                    const int off = ((2 * py - 2 + rr) * SW + 2 * px - 2) * 8;
                    u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
                    u32x4v q2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0);
                    const unsigned w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
                    float t[6][3];
#pragma unroll
                    for (int c = 0; c < 6; ++c) { t[c][0] = h2f(w[2 * c] & 0xffff); t[c][1] = h2f(w[2 * c] >> 16); t[c][2] = h2f(w[2 * c + 1] & 0xffff); }
#pragma unroll
                    for (int k = 0; k < 5; ++k)
#pragma unroll
                        for (int c = 0; c < 3; ++c) hx[rr][k][c] = lerpc(t[k][c], t[k + 1][c], 0.5f);
                }
                auto tapd = [&](int kx, int ky, int c) { return lerpc(hx[ky][kx][c], hx[ky + 1][kx][c], 0.5f); };
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sum = tapd(2, 2, c) * 0.125f;
                    sum = sum + (((tapd(0, 0, c) + tapd(4, 0, c)) + tapd(0, 4, c)) + tapd(4, 4, c)) * 0.03125f;
                    sum = sum + (((tapd(2, 0, c) + tapd(0, 2, c)) + tapd(4, 2, c)) + tapd(2, 4, c)) * 0.0625f;
                    sum = sum + (((tapd(1, 1, c) + tapd(3, 1, c)) + tapd(1, 3, c)) + tapd(3, 3, c)) * 0.125f;
                    if (A.dst_mip_level == 1) sum = fminf(sum, 1.0f);
                    r[c] = sum;
                }
                done = true;
            }
        } else {
            const int i = px >> 1, j = py >> 1;
            if (i >= 2 && i <= SW - 3 && j >= 2 && j <= SH - 3) {
                const bool ox = px & 1, oy = py & 1;
                const float fx_side = ox ? 0.75f : 0.25f, fx_mid = ox ? 0.25f : 0.75f;      // fractions of the taps at -1.5 / +1.5 and at 0
                const float fy_side = oy ? 0.75f : 0.25f, fy_mid = oy ? 0.25f : 0.75f;
                float hx[5][3][3];                                                      // rows x horizontal taps (left, centre, right) x rgb
#pragma unroll
                for (int rr = 0; rr < 5; ++rr) {
                    const int off = ((j - 2 + rr) * SW + i - 2) * 8;
                    u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
                    u32x2v q2 = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 32, 0, 0);
                    const unsigned w[10] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
                    float t[5][3];
#pragma unroll
                    for (int c = 0; c < 5; ++c) { t[c][0] = h2f(w[2 * c] & 0xffff); t[c][1] = h2f(w[2 * c] >> 16); t[c][2] = h2f(w[2 * c + 1] & 0xffff); }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        hx[rr][0][c] = lerpx(t[0][c], t[1][c], fx_side);
                        hx[rr][1][c] = lerpx(ox ? t[2][c] : t[1][c], ox ? t[3][c] : t[2][c], fx_mid);
                        hx[rr][2][c] = lerpx(t[3][c], t[4][c], fx_side);
                    }
                }
                auto tapu = [&](int kx, int ky, int c) {
                    if (ky == 0) return lerpx(hx[0][kx][c], hx[1][kx][c], fy_side);
                    if (ky == 2) return lerpx(hx[3][kx][c], hx[4][kx][c], fy_side);
                    return lerpx(oy ? hx[2][kx][c] : hx[1][kx][c], oy ? hx[3][kx][c] : hx[2][kx][c], fy_mid);
                };
                const float factor = A.dst_mip_level == 0 ? 0.06f : 1.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sum = tapu(1, 1, c) * 4.0f;
                    sum = sum + (((tapu(1, 0, c) + tapu(0, 1, c)) + tapu(2, 1, c)) + tapu(1, 2, c)) * 2.0f;
                    sum = sum + (((tapu(0, 0, c) + tapu(2, 0, c)) + tapu(0, 2, c)) + tapu(2, 2, c));
                    r[c] = sum * factor / 16.0f;
                }
                done = true;
            }
        }
    }
    if (!done) bloom_general<kUp>(P, rs, px, py, r);
    uint2* o = (uint2*)A.dst + (size_t)py * A.dst_width + px;
    if (A.blend_additive) {
        Rgba d = unpack_rgba16f(A.blend_src ? ((const uint2*)A.blend_src)[(size_t)py * A.dst_width + px] : *o);
        r[0] = r[0] + d.x; r[1] = r[1] + d.y; r[2] = r[2] + d.z;
    }
    *o = pack_half4(r[0], r[1], r[2]);
}

// K10 / K11, SMALL levels through the general sampler (the passes of a 1080p frame below 240 x 135, whose sizes are not 2 : 1): a level
// of a few thousand pixels cannot fill the chip, so a pass lasts as long as ONE thread's chain of 52 / 36 dependent-address fetches
// and ~1000 instructions.  Here four lanes share a pixel, one per group of the shader's sum (downsample: centre | four corners at
// +-2 | four edges at +-2 | four inner taps at +-1; upsample: centre | four edges | four corners | idle): a lane runs at most four
// taps, sums them in the shader's order, and lane 0 of the pixel folds the group sums -- again in the shader's order -- after three
// shuffles per channel.  Same taps, same operations, same order: bit-identical; the chain is a third as long.
template <bool kUp>
__global__ __launch_bounds__(256) void k_bloom_small(BloomParams P) {
    const PbrkBloomArgs& A = P.a;
    const int q = threadIdx.x & 3;
    const int px = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int py = A.y0 + blockIdx.y;
    const bool live = px < A.dst_width;                                                 // whole quads of lanes: the shuffles below stay inside a pixel
    const int pxc = min(px, A.dst_width - 1);
    const int SW = A.src.width, SH = A.src.height;
    const __amdgpu_buffer_rsrc_t rs = tex_rsrc(A.src, 8);
    SharedRcp rw, rh;
    rw.d = (float)A.dst_width; rw.r = P.rcp_dw; rh.d = (float)A.dst_height; rh.r = P.rcp_dh;
    const float u = div_by((float)pxc + 0.5f, rw), v = div_by((float)py + 0.5f, rh);      // fs_uv, as in bloom_general
    const bool centre = q == 0 || (kUp && q == 3);
    const bool edge = kUp ? q == 1 : q == 2;
    const float mag = centre ? 0.0f : (kUp || q == 3) ? 1.0f : 2.0f;                     // tap offsets in units of x_step / y_step
    float g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // corners (and the inner taps): (-,-) (+,-) (-,+) (+,+);  edges: (0,-) (-,0) (+,0) (0,+);  centre: (0,0)
        const float cx = (j & 1) ? 1.0f : -1.0f, cy = (j & 2) ? 1.0f : -1.0f;
        const float ex = j == 1 ? -1.0f : j == 2 ? 1.0f : 0.0f, ey = j == 0 ? -1.0f : j == 3 ? 1.0f : 0.0f;
        const float offx = (edge ? ex : cx) * mag, offy = (edge ? ey : cy) * mag;        // exact: +-2, +-1 or 0 (a centre lane repeats its tap)
        int i0, i1, j0, j1; float a, b;
        split_axis(u + offx * P.x_step, SW, i0, i1, a);
        split_axis(v + offy * P.y_step, SH, j0, j1, b);
        Rgba t00 = fetch_rgba16f(rs, SW, i0, j0), t10 = fetch_rgba16f(rs, SW, i1, j0);
        Rgba t01 = fetch_rgba16f(rs, SW, i0, j1), t11 = fetch_rgba16f(rs, SW, i1, j1);
        const float t[3] = {lerpx(lerpx(t00.x, t10.x, a), lerpx(t01.x, t11.x, a), b), lerpx(lerpx(t00.y, t10.y, a), lerpx(t01.y, t11.y, a), b),
                            lerpx(lerpx(t00.z, t10.z, a), lerpx(t01.z, t11.z, a), b)};
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = j == 0 ? t[c] : (centre ? g[c] : g[c] + t[c]);   // ((t0 + t1) + t2) + t3; the centre group is its one tap
    }
    const int base = (threadIdx.x & 63) & ~3;
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g1 = __shfl(g[c], base + 1), g2 = __shfl(g[c], base + 2), g3 = __shfl(g[c], base + 3);
        if (kUp) {                                                                      // bloom_upsample.glsl:43-58
            const float factor = A.dst_mip_level == 0 ? 0.06f : 1.0f;
            float sum = g[c] * 4.0f;
            sum = sum + g1 * 2.0f;
            sum = sum + g2;
            r[c] = sum * factor / 16.0f;
        } else {                                                                        // bloom_downsample.glsl:50-97
            float sum = g[c] * 0.125f;
            sum = sum + g1 * 0.03125f;
            sum = sum + g2 * 0.0625f;
            sum = sum + g3 * 0.125f;
            if (A.dst_mip_level == 1) sum = fminf(sum, 1.0f);
            r[c] = sum;
        }
    }
    if (!live || q != 0) return;
    uint2* o = (uint2*)A.dst + (size_t)py * A.dst_width + px;
    if (A.blend_additive) {
        Rgba d = unpack_rgba16f(A.blend_src ? ((const uint2*)A.blend_src)[(size_t)py * A.dst_width + px] : *o);
        r[0] = r[0] + d.x; r[1] = r[1] + d.y; r[2] = r[2] + d.z;
    }
    *o = pack_half4(r[0], r[1], r[2]);
}

// N consecutive texels (rgb) of row y starting at column x0: whole-row loads when the run lies inside the level, else texel by texel
// with the sampler's clamp to edge (a tap whose two texels clamp to the same one returns it whatever its fraction: p + t (p - p)).
template <int N>
__device__ __forceinline__ void bloom_row(const __amdgpu_buffer_rsrc_t rs, const int SW, const int SH, const int x0, const int y, const bool inside, float (*t)[3]) {
    static_assert(N == 5 || N == 8, "window width");
    if (inside) {
        const int off = (y * SW + x0) * 8;
        unsigned w[16];
        u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
        w[0] = q0.x; w[1] = q0.y; w[2] = q0.z; w[3] = q0.w; w[4] = q1.x; w[5] = q1.y; w[6] = q1.z; w[7] = q1.w;
        if (N == 5) { u32x2v q2 = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 32, 0, 0); w[8] = q2.x; w[9] = q2.y; }
        else {
            u32x4v q2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0), q3 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
            w[8] = q2.x; w[9] = q2.y; w[10] = q2.z; w[11] = q2.w; w[12] = q3.x; w[13] = q3.y; w[14] = q3.z; w[15] = q3.w;
        }
#pragma unroll
        for (int c = 0; c < N; ++c) { t[c][0] = h2f(w[2 * c] & 0xffff); t[c][1] = h2f(w[2 * c] >> 16); t[c][2] = h2f(w[2 * c + 1] & 0xffff); }
    } else {
        const int yc = clampi(y, 0, SH - 1);
#pragma unroll
        for (int c = 0; c < N; ++c) { const Rgba v = fetch_rgba16f(rs, SW, clampi(x0 + c, 0, SW - 1), yc); t[c][0] = v.x; t[c][1] = v.y; t[c][2] = v.z; }
    }
}

// K11, exact 1 : 2 upsample of LARGE levels: one thread per 2 x 2 block of target pixels.  The four pixels (2i + ox, 2j + oy) read the SAME
// 5 x 5 source window around texel (i, j); only their tap fractions differ (1/4 or 3/4 by parity): 15 loads and 75 conversions per four
// pixels instead of per pixel.  Every tap is the same lerp of the same texels with the same fraction as in k_bloom_pass, and every pixel's sum
// runs in the shader's order: bit-identical (tests/test_post.py compares all twelve targets with the shader text at 1920 x 1080).
// Blocks at the border of the level read their window texel by texel with the sampler's edge clamp (the closed-form fractions hold for
// every pixel; only the whole-row loads need the window inside the level).  Small levels stay with one pixel per thread: they are
// latency-bound and a four-pixel thread is four times as long.  (The downsample had a 2 x 2 form too -- 8 x 8 window, 25 distinct taps for
// 52 -- which kept 158 registers and two waves per SIMD on the 960 x 540 pass, 65 % of their lifetime waiting for 32 loads: the pair form
// below replaced it, 16.8 -> 14.2 us.)
__global__ __launch_bounds__(256) void k_bloom_quad(BloomParams P) {
    const PbrkBloomArgs& A = P.a;
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int qy = (A.y0 >> 1) + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * qx >= A.dst_width || 2 * qy >= A.y1) return;
    const int SW = A.src.width, SH = A.src.height;
    const __amdgpu_buffer_rsrc_t rs = tex_rsrc(A.src, 8);
    float r[2][2][3];                                                                   // [oy][ox][rgb]
    const bool interior = qx >= 2 && qx <= SW - 3 && qy >= 2 && qy <= SH - 3;
    {
        // horizontal taps of a row: [parity][left, centre, right]
        float hx[5][2][3][3];
#pragma unroll
        for (int rr = 0; rr < 5; ++rr) {
            float t[5][3];
            bloom_row<5>(rs, SW, SH, qx - 2, qy - 2 + rr, interior, t);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                hx[rr][0][0][c] = lerpc(t[0][c], t[1][c], 0.25f); hx[rr][1][0][c] = lerpc(t[0][c], t[1][c], 0.75f);
                hx[rr][0][1][c] = lerpc(t[1][c], t[2][c], 0.75f); hx[rr][1][1][c] = lerpc(t[2][c], t[3][c], 0.25f);
                hx[rr][0][2][c] = lerpc(t[3][c], t[4][c], 0.25f); hx[rr][1][2][c] = lerpc(t[3][c], t[4][c], 0.75f);
            }
        }
        const float factor = A.dst_mip_level == 0 ? 0.06f : 1.0f;
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                const float fy_side = oy ? 0.75f : 0.25f, fy_mid = oy ? 0.25f : 0.75f;
                auto tapu = [&](int kx, int ky, int c) {
                    if (ky == 0) return lerpc(hx[0][ox][kx][c], hx[1][ox][kx][c], fy_side);
                    if (ky == 2) return lerpc(hx[3][ox][kx][c], hx[4][ox][kx][c], fy_side);
                    return lerpc(hx[oy ? 2 : 1][ox][kx][c], hx[oy ? 3 : 2][ox][kx][c], fy_mid);
                };
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sum = tapu(1, 1, c) * 4.0f;
                    sum = sum + (((tapu(1, 0, c) + tapu(0, 1, c)) + tapu(2, 1, c)) + tapu(1, 2, c)) * 2.0f;
                    sum = sum + (((tapu(0, 0, c) + tapu(2, 0, c)) + tapu(0, 2, c)) + tapu(2, 2, c));
                    r[oy][ox][c] = sum * factor / 16.0f;
                }
            }
    }
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
        uint4* o = (uint4*)((uint2*)A.dst + (size_t)(2 * qy + oy) * A.dst_width + 2 * qx);
        if (A.blend_additive) {
            const uint4 d = A.blend_src ? *(const uint4*)((const uint2*)A.blend_src + (size_t)(2 * qy + oy) * A.dst_width + 2 * qx) : *o;
            const Rgba d0 = unpack_rgba16f(make_uint2(d.x, d.y)), d1 = unpack_rgba16f(make_uint2(d.z, d.w));
            r[oy][0][0] = r[oy][0][0] + d0.x; r[oy][0][1] = r[oy][0][1] + d0.y; r[oy][0][2] = r[oy][0][2] + d0.z;
            r[oy][1][0] = r[oy][1][0] + d1.x; r[oy][1][1] = r[oy][1][1] + d1.y; r[oy][1][2] = r[oy][1][2] + d1.z;
        }
        const uint2 a = pack_half4(r[oy][0][0], r[oy][0][1], r[oy][0][2]), b = pack_half4(r[oy][1][0], r[oy][1][1], r[oy][1][2]);
        *o = make_uint4(a.x, a.y, b.x, b.y);
    }
}

// K10, exact 2 : 1 downsample of LARGE levels: two horizontally adjacent target pixels per thread.  They share an 8 x 6 source window (24
// loads instead of 36) and 8 of their 2 x 13 taps coincide (a pixel's tap two texels to the right is its neighbour's centre tap): 18 distinct
// taps.  E[b][a]: tap at window position (2a, 2b); O[b][a]: tap at (2a + 1, 2b + 1); rows stream through (a tap needs two adjacent rows).
// Same lerps, same order of every pixel's sum as k_bloom_pass: bit-identical.
__global__ __launch_bounds__(256) void k_bloom_down_pair(BloomParams P) {
    const PbrkBloomArgs& A = P.a;
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = A.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * qx >= A.dst_width || py >= A.y1) return;
    const int SW = A.src.width, SH = A.src.height;
    const __amdgpu_buffer_rsrc_t rs = tex_rsrc(A.src, 8);
    const bool interior = qx >= 1 && 2 * qx + 1 <= A.dst_width - 2 && py >= 1 && py <= A.dst_height - 2;
    float E[3][4][3], O[2][3][3], prev[7][3];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        float t[8][3], cur[7][3];
        bloom_row<8>(rs, SW, SH, 4 * qx - 2, 2 * py - 2 + rr, interior, t);
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) cur[k][c] = lerpc(t[k][c], t[k + 1][c], 0.5f);
        if (rr >= 1) {
            const int wy = rr - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if ((wy & 1) == 0) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) E[wy >> 1][a][c] = lerpc(prev[2 * a][c], cur[2 * a][c], 0.5f);
                } else {
#pragma unroll
                    for (int a = 0; a < 3; ++a) O[wy >> 1][a][c] = lerpc(prev[2 * a + 1][c], cur[2 * a + 1][c], 0.5f);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) prev[k][c] = cur[k][c];
    }
    float r[2][3];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            auto ev = [&](int kx, int ky) { return E[ky >> 1][dx + (kx >> 1)][c]; };
            auto od = [&](int kx, int ky) { return O[ky >> 1][dx + (kx >> 1)][c]; };
            float sum = ev(2, 2) * 0.125f;
            sum = sum + (((ev(0, 0) + ev(4, 0)) + ev(0, 4)) + ev(4, 4)) * 0.03125f;
            sum = sum + (((ev(2, 0) + ev(0, 2)) + ev(4, 2)) + ev(2, 4)) * 0.0625f;
            sum = sum + (((od(1, 1) + od(3, 1)) + od(1, 3)) + od(3, 3)) * 0.125f;
            if (A.dst_mip_level == 1) sum = fminf(sum, 1.0f);
            r[dx][c] = sum;
        }
    const uint2 a = pack_half4(r[0][0], r[0][1], r[0][2]), b = pack_half4(r[1][0], r[1][1], r[1][2]);
    *(uint4*)((uint2*)A.dst + (size_t)py * A.dst_width + 2 * qx) = make_uint4(a.x, a.y, b.x, b.y);
}

__global__ void k_debug_sample_post(PbrkTex2D t, const float* __restrict__ coords, int count, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Rgba r = sample_rgba16f(t, coords[i * 3], coords[i * 3 + 1]);
    out[i] = make_float4(r.x, r.y, r.z, r.w);
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
    if ((long long)a->width * a->height > (1ll << 27) || (long long)a->prev_frame_result.width * a->prev_frame_result.height > (1ll << 27)) return PBRK_E_ARG;   // 32-bit byte offsets
    const bool centre = a->width <= 8192 && a->height <= 8192;
    const bool half = a->out_format == PBRK_FMT_RGBA16F;
    dim3 grid((a->width + 63) / 64, (a->y1 - a->y0 + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    TaaParams tp; tp.a = *a; tp.psx = 1.0f / (float)a->lighting_result.width; tp.psy = 1.0f / (float)a->lighting_result.height;
    if (centre && half) hipLaunchKernelGGL((k_taa_resolve<true, true>), grid, block, 0, st, tp);
    else if (centre) hipLaunchKernelGGL((k_taa_resolve<true, false>), grid, block, 0, st, tp);
    else if (half) hipLaunchKernelGGL((k_taa_resolve<false, true>), grid, block, 0, st, tp);
    else hipLaunchKernelGGL((k_taa_resolve<false, false>), grid, block, 0, st, tp);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" int pbrk_final_post_process(const PbrkFinalArgs* a, void* stream) {
    if (!a || !a->out || a->width < 1 || a->height < 1 || a->y0 < 0 || a->y0 >= a->y1 || a->y1 > a->height) return PBRK_E_ARG;
    if (!tex_ok(a->src, PBRK_FMT_RGBA16F)) return PBRK_E_FORMAT;
    if (a->out == a->src.data) return PBRK_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool stream_map = a->src.width == a->width && a->src.height == a->height && a->width <= 8192 && a->height <= 8192 && (a->width & 3) == 0;
    dim3 block(256);
    dim3 grid = stream_map ? dim3((unsigned)((((size_t)(a->width >> 2) * (a->y1 - a->y0)) + 255) / 256)) : dim3((a->width + 63) / 64, (a->y1 - a->y0 + 3) / 4);
#define PBRK_FINAL_LAUNCH(F) do { if (stream_map) hipLaunchKernelGGL((k_final_post_process_stream<F>), grid, block, 0, st, *a); \
                                  else hipLaunchKernelGGL((k_final_post_process<F>), grid, block, 0, st, *a); } while (0)
    switch (a->out_format) {
    case PBRK_FMT_RGBA8UN: PBRK_FINAL_LAUNCH(PBRK_FMT_RGBA8UN); break;
    case PBRK_FMT_BGRA8UN: PBRK_FINAL_LAUNCH(PBRK_FMT_BGRA8UN); break;
    case PBRK_FMT_RGBA16F: PBRK_FINAL_LAUNCH(PBRK_FMT_RGBA16F); break;
    case PBRK_FMT_RGBA32F: PBRK_FINAL_LAUNCH(PBRK_FMT_RGBA32F); break;
    default: return PBRK_E_FORMAT;
    }
#undef PBRK_FINAL_LAUNCH
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

// which instantiation a bloom pass takes (all bit-identical): < 0 = the environment variable or the default
static long long g_quad_min = -1, g_small_max = -1;
extern "C" void pbrk_bloom_set_thresholds(long long quad_min_pixels, long long small_max_pixels) { g_quad_min = quad_min_pixels; g_small_max = small_max_pixels; }

extern "C" int pbrk_bloom_pass(const PbrkBloomArgs* a, void* stream) {
    if (!a || !a->dst || a->dst_width < 1 || a->dst_height < 1 || a->y0 < 0 || a->y0 >= a->y1 || a->y1 > a->dst_height) return PBRK_E_ARG;
    if (!tex_ok(a->src, PBRK_FMT_RGBA16F)) return PBRK_E_FORMAT;
    if ((long long)a->src.width * a->src.height > (1ll << 27) || a->dst == a->src.data) return PBRK_E_ARG;
    BloomParams p;
    p.a = *a;
    p.x_step = (a->upsample ? 1.5f : 1.0f) / (float)a->src.width;                       // radius / size, 1 / size (the shaders' x, y)
    p.y_step = (a->upsample ? 1.5f : 1.0f) / (float)a->src.height;
    p.rcp_dw = 1.0f / (float)a->dst_width; p.rcp_dh = 1.0f / (float)a->dst_height;
    // closed-form taps need the exact ratio and the snap to absorb the fp32 coordinate error (extents <= 8192), see the kernel
    const bool small = a->src.width <= 8192 && a->src.height <= 8192 && a->dst_width <= 8192 && a->dst_height <= 8192;
    p.exact2to1 = small && (a->upsample ? (a->dst_width == 2 * a->src.width && a->dst_height == 2 * a->src.height)
                                        : (a->src.width == 2 * a->dst_width && a->src.height == 2 * a->dst_height));
    // two (downsample) / 2 x 2 (upsample) pixels per thread where the level is large enough to fill the chip with such threads (PBR_BLOOM_QUAD_MIN_PIXELS)
    if (g_quad_min < 0) { const char* e = getenv("PBR_BLOOM_QUAD_MIN_PIXELS"); g_quad_min = e ? atoll(e) : 100000; }
    const long long quad_min = g_quad_min;
    const bool wide = p.exact2to1 && ((uintptr_t)a->dst & 15) == 0 && ((uintptr_t)a->src.data & 15) == 0 && ((uintptr_t)a->blend_src & 15) == 0 &&
                      (long long)a->dst_width * (a->y1 - a->y0) >= quad_min;
    if (wide && !a->upsample && !a->blend_additive && !(a->dst_width & 1)) {
        hipLaunchKernelGGL(k_bloom_down_pair, dim3((a->dst_width / 2 + 63) / 64, (a->y1 - a->y0 + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    }
    if (wide && a->upsample && !((a->dst_width | a->dst_height | a->y0 | a->y1) & 1)) {
        dim3 qgrid((a->dst_width / 2 + 63) / 64, ((a->y1 - a->y0) / 2 + 3) / 4);
        hipLaunchKernelGGL(k_bloom_quad, qgrid, dim3(256), 0, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    }
    // four lanes per pixel where a level is too small to fill the chip and goes through the general sampler (PBR_BLOOM_SMALL_MAX_PIXELS)
    if (g_small_max < 0) { const char* e = getenv("PBR_BLOOM_SMALL_MAX_PIXELS"); g_small_max = e ? atoll(e) : 40000; }
    const long long small_max = g_small_max;
    if (!p.exact2to1 && (long long)a->dst_width * (a->y1 - a->y0) <= small_max) {
        dim3 sgrid((a->dst_width + 63) / 64, a->y1 - a->y0);
        if (a->upsample) hipLaunchKernelGGL((k_bloom_small<true>), sgrid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((k_bloom_small<false>), sgrid, dim3(256), 0, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    }
    dim3 grid((a->dst_width + 63) / 64, (a->y1 - a->y0 + 3) / 4), block(256);
    if (a->upsample) hipLaunchKernelGGL((k_bloom_pass<true>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_bloom_pass<false>), grid, block, 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" __attribute__((visibility("hidden"))) int pbrk_debug_sample_post(const void* texture, int w, int h, const void* coords, int count, void* out, void* stream) {
    PbrkTex2D t; t.data = texture; t.format = PBRK_FMT_RGBA16F; t.width = w; t.height = h;
    hipLaunchKernelGGL(k_debug_sample_post, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (const float*)coords, count, (float4*)out);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
