// k_shade.hip -- K5: deferred Cook-Torrance shade pass (gfx950).
//
// Replaces the fragment stage of shaders/lighting_pass.glsl (main :432-716) as drawn by
// render.cpp:1119-1127 (full-screen triangle => one invocation per pixel, point fetch of the
// G-buffer at pixel centres).  In-scope sub-blocks: G-buffer decode :433-442,
// position reconstruction :444-451, interleaved-gradient noise :456-459, view vector :612-613,
// F0/kS/kD :657-661, sun term (GGX D :21-31, Mikkelsen G :72-74, Schlick F :76-79) :664-679,
// split-sum LUT fetch :681, reflection vector :693-697, specular compose :702, emissive :706,
// sky :708-710, clamp :712-713.  IBL mode adds the commented lines :690 and :699.  The blocks fed by
// raster passes are optional (flags): sun shadows :594-608 and light shafts :622-651 (PBRK_SHADE_SHADOWS /
// _SHAFTS, sun depth map), the voxel-GI ambient / specular traces :273-424, :546-577, :685, :701
// (PBRK_SHADE_GI, instantiation k_shade<true>: light grid, previous-frame pyramid, depth buffer).  With all
// three the kernel is the reference's complete live shader; without them GI == 0 and shadow == 1.
//
// Streaming kernel: 20 B read + 8 B written per pixel; irradiance / prefiltered cubes, LUT and
// Globals are cache resident.  Discontinuous inputs (noise, sky test, N.L > 0) are evaluated in the
// shader's operation order so that branch decisions match a CPU evaluation bit for bit.
#include "pbr_device.h"
#include "pbr_kernels.h"
#include <hip/hip_fp16.h>
#include <string.h>
#include <stdlib.h>
#include "k_shade_internal.h"


// sampler2DShadow + SAMPLER_PERCENTAGE_CLOSER (render.cpp:664-673: linear, clamp, compare Less): each bilinear tap contributes
// (ref < texel ? 1 : 0); coordinates snapped to 1/256 texel (the 2-D sampler convention of k_post.hip / the oracle).  EXACT.
__device__ __forceinline__ float shadow_sample(const float* __restrict__ d, int w, int h, float u, float v, float ref) {
    float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    fx = floorf(fx * 256.0f + 0.5f) * (1.0f / 256.0f);
    fy = floorf(fy * 256.0f + 0.5f) * (1.0f / 256.0f);
    float flx = floorf(fx), fly = floorf(fy);
    float a = fx - flx, b = fy - fly;
    int i0 = (int)fminf(fmaxf(flx, -1.0f), (float)w), j0 = (int)fminf(fmaxf(fly, -1.0f), (float)h);   // float-domain clamp: see snap_split
    int i1 = min(max(i0 + 1, 0), w - 1), j1 = min(max(j0 + 1, 0), h - 1);
    i0 = min(max(i0, 0), w - 1); j0 = min(max(j0, 0), h - 1);
    float c00 = ref < d[j0 * w + i0] ? 1.0f : 0.0f, c10 = ref < d[j0 * w + i1] ? 1.0f : 0.0f;
    float c01 = ref < d[j1 * w + i0] ? 1.0f : 0.0f, c11 = ref < d[j1 * w + i1] ? 1.0f : 0.0f;
    float top = c00 + a * (c10 - c00), bot = c01 + a * (c11 - c01);
    return top + b * (bot - top);
}

__device__ __forceinline__ float fract_(float x) { return x - floorf(x); }
// EXACT: InterleavedGradientNoise, lighting_pass.glsl:119-121
__device__ __forceinline__ float ign(float px, float py) {
    return fract_(52.9829189f * fract_(0.06711056f * px + 0.00583715f * py));
}
__device__ __forceinline__ float ggx_d(float NdotH, float roughness) {      // :21-31
    float a = roughness * roughness;
    float a2 = a * a;
    float n2 = NdotH * NdotH;
    float denom = (n2 * (a2 - 1.0f) + 1.0f);
    denom = PBR_PI * denom * denom;
    return a2 * __builtin_amdgcn_rcpf(denom);        // continuous term: 1-ulp reciprocal is inside the tolerance
}
__device__ __forceinline__ f3 fresnel_schlick(float c, f3 F0) {             // :76-79
    float p = pow5(1.0f - c);          // pow(x, 5.) of the shader; within 2 ulp of powf, not a discontinuity
    return mk3(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
}

__device__ __forceinline__ int bordered_level_off(int W, int level) {
    int off = 0;
    for (int l = 0; l < level; ++l) { int n = max(W >> l, 1) + 2; off += 6 * n * n; }
    return off;
}
// per-block table of level offsets (texels): [l] bordered pyramid, [16 + l] cells twin; filled once by 16 lanes
__device__ __forceinline__ void fill_level_table(int* tab, int W, int levels, int cells_first) {
    if (threadIdx.x < 16) {
        int l = min((int)threadIdx.x, levels - 1);
        tab[threadIdx.x] = bordered_level_off(W, l);
        tab[16 + threadIdx.x] = cells_level_off(W, cells_first, l);
    }
    __syncthreads();
}
__device__ __forceinline__ f3 level_fetch(const float4* __restrict__ pyr, const float4* __restrict__ cells, int cells_first,
                                          int W, int l, int face, float s, float t, const int* tab, bool snap = false) {
    int n = max(W >> l, 1);
    CubeTap tp = cube_tap_from_st(face, s, t, n, snap);
    if (cells && l >= cells_first) return fetch_rgb_cells_tap(cells + tab[16 + l], n, tp);
    return fetch_rgb_tap(pyr + tab[l], n, tp);
}
// trilinear fetch from a bordered pyramid (sampler: linear mip filter, LOD clamped to the chain)
template <bool kExactDiv>
__device__ __forceinline__ f3 pyramid_fetch(const float4* __restrict__ pyr, const float4* __restrict__ cells, int cells_first,
                                            int W, int levels, f3 d, float lod, const int* tab, bool snap = false) {
    CubeST cs = cube_select(d);
    float s, t;
    if (kExactDiv) cube_st_exact(cs, &s, &t); else cube_st_shared(cs, &s, &t);
    float maxl = (float)(levels - 1);
    lod = fminf(fmaxf(lod, 0.0f), maxl);
    if (snap) lod = snap256(lod);                                  // diagnostic sampler convention: 8 bits of LOD fraction
    float fl = floorf(lod);
    int l0 = (int)fl;
    float w = lod - fl;
    f3 c0 = level_fetch(pyr, cells, cells_first, W, l0, cs.face, s, t, tab, snap);
    if (w > 0.0f) {
        int l1 = min(l0 + 1, levels - 1);
        f3 c1 = level_fetch(pyr, cells, cells_first, W, l1, cs.face, s, t, tab, snap);
        c0.x = lerp_fma(c0.x, c1.x, w); c0.y = lerp_fma(c0.y, c1.y, w); c0.z = lerp_fma(c0.z, c1.z, w);
    }
    return c0;
}

__device__ __forceinline__ float2 lut_fetch(const __half2* __restrict__ lut, const uint4* __restrict__ cells, int S, float u, float v) {
    float fx = u * (float)S - 0.5f, fy = v * (float)S - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float a = fx - flx, b = fy - fly;
    int i0 = (int)flx, j0 = (int)fly;
    if (cells) {
        int ci = min(max(i0 + 1, 0), S), cj = min(max(j0 + 1, 0), S);
        uint4 c = cells[cj * (S + 1) + ci];
        float2 t00 = __half22float2(*reinterpret_cast<__half2*>(&c.x)), t10 = __half22float2(*reinterpret_cast<__half2*>(&c.y));
        float2 t01 = __half22float2(*reinterpret_cast<__half2*>(&c.z)), t11 = __half22float2(*reinterpret_cast<__half2*>(&c.w));
        float2 r;
        r.x = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
        r.y = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
        return r;
    }
    int i1 = min(max(i0 + 1, 0), S - 1), j1 = min(max(j0 + 1, 0), S - 1);
    i0 = min(max(i0, 0), S - 1); j0 = min(max(j0, 0), S - 1);
    float2 t00 = __half22float2(lut[j0 * S + i0]), t10 = __half22float2(lut[j0 * S + i1]);
    float2 t01 = __half22float2(lut[j1 * S + i0]), t11 = __half22float2(lut[j1 * S + i1]);
    float2 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
    return r;
}

// ---- N4: voxel-GI sampling (lighting_pass.glsl:273-424, :546-577).  Branch decisions along the rays (alpha thresholds, ray
//      behind the visible surface, leaving the screen) must agree with a CPU evaluation, so everything here is the shader's
//      operation order in correctly rounded fp32, and sin / cos / acos are the fixed polynomials the oracle defines.  EXACT.
__device__ __forceinline__ void sincos_det(float x, float* sn, float* cs) {
    float kf = floorf(x * 0.63661977236758134f + 0.5f);
    float r = fmaf(-kf, 1.5707962512969971f, x);
    r = fmaf(-kf, 7.5497894158615964e-08f, r);
    float z = r * r;
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    int k = (int)kf & 3;
    *sn = (k == 0) ? ps : (k == 1) ? pc : (k == 2) ? -ps : -pc;
    *cs = (k == 0) ? pc : (k == 1) ? -ps : (k == 2) ? -pc : ps;
}
__device__ __forceinline__ float asin_poly_det(float x) {
    float z = x * x;
    float pp = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    return fmaf(pp * z, x, x);
}
__device__ __forceinline__ float acos_det(float x) {
    if (x > 0.5f) return 2.0f * asin_poly_det(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - asin_poly_det(x);
}
__device__ __forceinline__ void snap_split(float coord01, int extent, int& i0, int& i1, float& a) {
    float f = coord01 * (float)extent - 0.5f;
    f = floorf(f * 256.0f + 0.5f) * (1.0f / 256.0f);
    float fl = floorf(f);
    a = f - fl;
    // clamp in the float domain first: a ray that has marched to 1e30 must not reach the float->int conversion (saturation +
    // `i + 1` would be signed overflow, i.e. an arbitrary index); the clamped indices are the same as clamping the true index
    int i = (int)fminf(fmaxf(fl, -1.0f), (float)extent);
    i0 = min(max(i, 0), extent - 1); i1 = min(max(i + 1, 0), extent - 1);
}
__device__ __forceinline__ float lerp_x(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ float4 unpack_h4(uint2 v) {
    __half2 lo = *reinterpret_cast<__half2*>(&v.x), hi = *reinterpret_cast<__half2*>(&v.y);
    float2 a = __half22float2(lo), b = __half22float2(hi);
    return make_float4(a.x, a.y, b.x, b.y);
}
// texture(sampler3D(LIGHTGRID, SAMPLER_LINEAR_CLAMP), p): trilinear, clamp, snapped coordinates
__device__ __forceinline__ float4 grid_sample(const ShadeParams& p, float px, float py, float pz) {
    int i0, i1, j0, j1, k0, k1; float a, b, c;
    const int n = p.grid_n;
    snap_split(px, n, i0, i1, a); snap_split(py, n, j0, j1, b); snap_split(pz, n, k0, k1, c);
    float4 t000 = unpack_h4(p.grid[(k0 * n + j0) * n + i0]), t001 = unpack_h4(p.grid[(k0 * n + j0) * n + i1]);
    float4 t010 = unpack_h4(p.grid[(k0 * n + j1) * n + i0]), t011 = unpack_h4(p.grid[(k0 * n + j1) * n + i1]);
    float4 t100 = unpack_h4(p.grid[(k1 * n + j0) * n + i0]), t101 = unpack_h4(p.grid[(k1 * n + j0) * n + i1]);
    float4 t110 = unpack_h4(p.grid[(k1 * n + j1) * n + i0]), t111 = unpack_h4(p.grid[(k1 * n + j1) * n + i1]);
    float4 r;
    r.x = lerp_x(lerp_x(lerp_x(t000.x, t001.x, a), lerp_x(t010.x, t011.x, a), b), lerp_x(lerp_x(t100.x, t101.x, a), lerp_x(t110.x, t111.x, a), b), c);
    r.y = lerp_x(lerp_x(lerp_x(t000.y, t001.y, a), lerp_x(t010.y, t011.y, a), b), lerp_x(lerp_x(t100.y, t101.y, a), lerp_x(t110.y, t111.y, a), b), c);
    r.z = lerp_x(lerp_x(lerp_x(t000.z, t001.z, a), lerp_x(t010.z, t011.z, a), b), lerp_x(lerp_x(t100.z, t101.z, a), lerp_x(t110.z, t111.z, a), b), c);
    r.w = lerp_x(lerp_x(lerp_x(t000.w, t001.w, a), lerp_x(t010.w, t011.w, a), b), lerp_x(lerp_x(t100.w, t101.w, a), lerp_x(t110.w, t111.w, a), b), c);
    return r;
}
__device__ __forceinline__ float4 grid_at(const ShadeParams& p, f3 ro) { return grid_sample(p, ro.x * 0.5f + 0.5f, ro.y * 0.5f + 0.5f, ro.z * 0.5f + 0.5f); }
// one level of PREV_FRAME_RESULT, bilinear clamp with snapped coordinates (xyz only)
__device__ __forceinline__ f3 prev_level_sample(const ShadeParams& p, int l, float u, float v) {
    const int w = max(p.prev_w >> l, 1), h = max(p.prev_h >> l, 1);
    int i0, i1, j0, j1; float a, b;
    snap_split(u, w, i0, i1, a); snap_split(v, h, j0, j1, b);
    const uint2* t = p.prev[l];
    float4 t00 = unpack_h4(t[j0 * w + i0]), t10 = unpack_h4(t[j0 * w + i1]), t01 = unpack_h4(t[j1 * w + i0]), t11 = unpack_h4(t[j1 * w + i1]);
    return mk3(lerp_x(lerp_x(t00.x, t10.x, a), lerp_x(t01.x, t11.x, a), b), lerp_x(lerp_x(t00.y, t10.y, a), lerp_x(t01.y, t11.y, a), b),
               lerp_x(lerp_x(t00.z, t10.z, a), lerp_x(t01.z, t11.z, a), b));
}
__device__ __forceinline__ f3 luminance_normalise(float x, float y, float z) {            // :267-269, :314-316, :420-422
    float luminance = 0.299f * x + 0.587f * y + 0.114f * z;
    float k = sqrtf(luminance) / fmaxf(luminance, 0.0001f);
    return mk3(x * k, y * k, z * k);
}
#ifndef GI_AHEAD
#define GI_AHEAD 4
#endif
// lighting_pass.glsl:273-424 SampleRadianceWithScreenSpaceTrace
__device__ __forceinline__ f3 sample_radiance_ss(const ShadeParams& p, f3 V, const float* p0_vs, f3 ray_origin, f3 ray_direction,
                                              int num_steps, float step_scale, float noise_01, float foggyness, float ss_intensity) {
    const float voxel_scale = 2.0f / 128.0f;                                              // :274
    const float ls = p.lightgrid_scale;
    f3 rd = scale3(ray_direction, voxel_scale);
    f3 ro = scale3(ray_origin, ls);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0001f;
    for (int i = 0; i < 4; ++i) {                                                         // :281-288
        ro = add3(ro, rd);
        float4 rad = grid_at(p, ro);
        if (rad.w < 0.3f) { s0 += rad.x; s1 += rad.y; s2 += rad.z; s3 += 1.0f; break; }
    }
    float open_vs[4];
    mat_mul(p.vfw, ro.x / ls, ro.y / ls, ro.z / ls, 1.0f, open_vs);                       // :290
    const float d0 = open_vs[0] - p0_vs[0], d1 = open_vs[1] - p0_vs[1], d2 = open_vs[2] - p0_vs[2];   // :298
    float step_length = fmaxf(p0_vs[2], 1.0f) * (1.0f + noise_01) / 100.0f;               // :300
    const float lxy = sqrtf(d0 * d0 + d1 * d1);
    f3 ssray_step = scale3(mk3(d0 / lxy, d1 / lxy, d2 / lxy), step_length);               // :301-302
    f3 pos = mk3(p0_vs[0], p0_vs[1], p0_vs[2]);
    const float dist_to_travel = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);                      // :308
    float dist_travelled = 0.0f;
    for (int guard = 0; guard < 512; ++guard) {                                           // :315 for (;;): every wave reaches an exit (steps grow >= 1.2x)
        pos = add3(pos, ssray_step);
        dist_travelled += step_length;
        float ndc[4];
        mat_mul(p.cfv, pos.x, pos.y, pos.z, 1.0f, ndc);                                   // :319-320
        const float nw = ndc[3];
        ndc[0] = ndc[0] / nw; ndc[1] = ndc[1] / nw;
        if (fminf(fmaxf(ndc[0], -1.0f), 1.0f) != ndc[0] || fminf(fmaxf(ndc[1], -1.0f), 1.0f) != ndc[1]) {   // :322-330
            float4 s4 = grid_at(p, mk3(ray_origin.x * ls + 2.5f * V.x * voxel_scale, ray_origin.y * ls + 2.5f * V.y * voxel_scale,
                                       ray_origin.z * ls + 2.5f * V.z * voxel_scale));
            return luminance_normalise(s4.x, s4.y, s4.z);
        }
        ssray_step = scale3(ssray_step, 1.2f); step_length *= 1.2f;                       // :332-333
        const float su = ndc[0] * 0.5f + 0.5f, sv = ndc[1] * 0.5f + 0.5f;
        const int di = min(max((int)fminf(fmaxf(floorf(su * (float)p.width), 0.0f), (float)p.width), 0), p.width - 1);
        const int dj = min(max((int)fminf(fmaxf(floorf(sv * (float)p.height), 0.0f), (float)p.height), 0), p.height - 1);
        const float depth_ndc = p.depth[(size_t)dj * p.width + di];                       // :335 SAMPLER_NEAREST_CLAMP
        float surf[4];
        mat_mul(p.vfc, ndc[0], ndc[1], depth_ndc, 1.0f, surf);                            // :338-339
        const float sw = surf[3];
        surf[0] = surf[0] / sw; surf[1] = surf[1] / sw; surf[2] = surf[2] / sw; surf[3] = surf[3] / sw;
        const float ls_surf = sqrtf(surf[0] * surf[0] + surf[1] * surf[1] + surf[2] * surf[2]);
        const float ls_pos = sqrtf(pos.x * pos.x + pos.y * pos.y + pos.z * pos.z);
        if (ls_surf < ls_pos) {                                                           // :343
            float ts[4], te[4];
            mat_mul(p.wfv, surf[0], surf[1], surf[2], surf[3], ts);                       // :348-349
            mat_mul(p.wfv, pos.x, pos.y, pos.z, 1.0f, te);
#pragma unroll
            for (int k = 0; k < 3; ++k) { ts[k] = ts[k] * ls * 0.5f + 0.5f; te[k] = te[k] * ls * 0.5f + 0.5f; }
            const float noise_offset = noise_01 * 0.2f;                                   // :351
            float alpha = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {                                                 // :352-355
                const float t = noise_offset + (k == 0 ? 0.2f : (k == 1 ? 0.4f : 0.6f));
                float4 r4 = grid_sample(p, mix_(ts[0], te[0], t), mix_(ts[1], te[1], t), mix_(ts[2], te[2], t));
                alpha = (k == 0) ? r4.w : alpha + r4.w;
            }
            if (alpha < 1.5f) {                                                           // :357-361
                const float f = 2.0f + noise_01;
                ssray_step = scale3(ssray_step, f); step_length *= f;
                continue;
            }
            float lod = fminf(fmaxf(fminf(step_length * 5.0f, 5.0f), 0.0f), (float)(p.prev_levels - 1));   // :380, clamped to the chain
            const float fl = floorf(lod);
            const int l0 = (int)fl;
            const float w = lod - fl;
            f3 c = prev_level_sample(p, l0, su, sv);
            if (w > 0.0f) {
                f3 c1 = prev_level_sample(p, min(l0 + 1, p.prev_levels - 1), su, sv);
                c = mk3(lerp_x(c.x, c1.x, w), lerp_x(c.y, c1.y, w), lerp_x(c.z, c1.z, w));
            }
            return mk3(c.x * ss_intensity, c.y * ss_intensity, c.z * ss_intensity);       // :382
        }
        if (dist_travelled > dist_to_travel) break;                                       // :393
    }
    if (s3 < 0.5f) return mk3(0.0f, 0.0f, 0.0f);                                          // :400-403
    rd = scale3(rd, step_scale);                                                          // :407-408
    ro = mk3(ro.x + rd.x * noise_01, ro.y + rd.y * noise_01, ro.z + rd.z * noise_01);
    // :411-419.  The march positions do not depend on what is read (only the exit does), so the samples of GI_AHEAD steps are
    // fetched together and consumed in order: the same values in the same order, GI_AHEAD times fewer dependent round trips
    // (num_steps is 12 or 16).
    for (int i = 0; i < num_steps; i += GI_AHEAD) {
        f3 rk[GI_AHEAD]; float4 radk[GI_AHEAD];
#pragma unroll
        for (int k = 0; k < GI_AHEAD; ++k) {
            ro = mk3(ro.x + 0.5f * rd.x, ro.y + 0.5f * rd.y, ro.z + 0.5f * rd.z);
            rk[k] = ro;
            radk[k] = grid_at(p, ro);
        }
        bool done = false;
#pragma unroll
        for (int k = 0; k < GI_AHEAD; ++k) {
            if (done || i + k >= num_steps) break;
            if (radk[k].w > 0.3f) { done = true; break; }
            s0 = s0 * foggyness + radk[k].x; s1 = s1 * foggyness + radk[k].y; s2 = s2 * foggyness + radk[k].z; s3 = s3 * foggyness + 1.0f;
        }
        if (done) break;
    }
    return luminance_normalise(s0 / s3, s1 / s3, s2 / s3);                                // :421-423
}

// kGI: the voxel-GI traces need ~250 VGPRs; the common modes keep their own instantiation (54 VGPRs)
template <bool kGI>
__global__ __launch_bounds__(256) void k_shade(const ShadeParams p) {
    __shared__ int level_tab[32];
    fill_level_table(level_tab, p.pre_size, min(p.pre_levels, 16), p.pre_cells_first);
    {   // grid: x = 64-pixel column blocks, y = 4-row blocks (no integer division per pixel).  The GI instantiation takes 8 x 8 pixels
        // per wave (a 32 x 8 block): its traces diverge, and a square footprint has fewer waves that straddle a surface edge and
        // rays that stay closer together than a 64 x 1 strip (lane utilisation 0.53 -> see DESIGN.md).
        int lx, ly;
        if (kGI) { lx = blockIdx.x * 32 + (threadIdx.x >> 6) * 8 + (threadIdx.x & 7); ly = blockIdx.y * 8 + ((threadIdx.x >> 3) & 7); }
        else { lx = blockIdx.x * 64 + (threadIdx.x & 63); ly = blockIdx.y * 4 + (threadIdx.x >> 6); }
        if (lx >= p.w || ly >= p.h) return;
        int px = p.x0 + lx, py = p.y0 + ly;
        size_t pi = (size_t)py * p.width + px;
        uchar4 bb = p.base[pi], nn = p.normal[pi], oo = p.orm[pi], ee = p.emissive[pi];
        float depth = p.depth[pi];

        // :433-442
        f3 base = mk3(unorm8(bb.x), unorm8(bb.y), unorm8(bb.z));
        f3 N = mk3(unorm8(nn.x) * 2.0f - 1.0f, unorm8(nn.y) * 2.0f - 1.0f, unorm8(nn.z) * 2.0f - 1.0f);
        float roughness = unorm8(oo.y), metallic = unorm8(oo.z);
        f3 emissive = mk3(unorm8(ee.x) * 10.0f, unorm8(ee.y) * 10.0f, unorm8(ee.z) * 10.0f);

        // :444-451
        SharedRcp rw, rh;
        rw.d = (float)p.width; rw.r = p.rcp_width; rh.d = (float)p.height; rh.r = p.rcp_height;
        float fs_u = div_by((float)px + 0.5f, rw), fs_v = div_by((float)py + 0.5f, rh);
        float pw[4];
        mat_mul(p.wfc, fs_u * 2.0f - 1.0f, fs_v * 2.0f - 1.0f, depth, 1.0f, pw);   // world_space_from_clip
        // kGI: the traces branch on quantities derived from P, V, R: true divisions there, so that the kernel follows the CPU
        // evaluation always; the other modes (1e-4 tolerance, no data-dependent control flow) share reciprocals
        f3 P;
        if (kGI) P = mk3(pw[0] / pw[3], pw[1] / pw[3], pw[2] / pw[3]);
        else { SharedRcp rpw = shared_rcp(pw[3]); P = mk3(div_by(pw[0], rpw), div_by(pw[1], rpw), div_by(pw[2], rpw)); }

        // :456-459
        float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
        float noise_offset = (1000 * 1.61803398875f) * p.frame_idx_mod_59;                       // frame_idx_mod_59
        float noise_1 = fract_(ign(fcx, fcy) + noise_offset);
        float noise_2 = fract_(ign(fcx + 90.0f, fcy + 20.0f) + noise_offset);
        float noise_3 = fract_(ign(fcx + 522.0f, fcy + 55.0f) + noise_offset);

        f3 cam = mk3(p.cam[0], p.cam[1], p.cam[2]);
        f3 V = kGI ? normalize3(sub3(cam, P)) : normalize3_shared(sub3(cam, P));       // :612
        float VdotN = fmaxf(dot3(V, N), 0.0f);                                         // :613
        f3 sun_emission = mk3(25.0f * 1.0f, 25.0f * 0.9f, 25.0f * 0.7f);               // :616
        f3 outl = mk3(0.0f, 0.0f, 0.0f);

        bool sky = (fminf(fmaxf(P.x, -99.0f), 99.0f) != P.x) || (fminf(fmaxf(P.y, -99.0f), 99.0f) != P.y) ||
                   (fminf(fmaxf(P.z, -99.0f), 99.0f) != P.z);                          // :708

        // :708-710 the sky branch replaces everything else: take it first (most waves of a frame are all-sky or all-surface)
        if (sky) {
            outl = pyramid_fetch<kGI>(p.pre, p.pre_cells, p.pre_cells_first, p.pre_size, p.pre_levels, mk3(-V.x, -V.y, -V.z), 1.0f, level_tab, p.snap != 0);
        } else {
            // :594-608 sun shadow (1 without PBRK_SHADE_SHADOWS); p0_sun_space also feeds the light shafts.  Inside the surface branch: a
            // sky pixel uses neither (four PCF taps saved per sky pixel: the shadows-only 1080p frame 38 -> 33 us)
            float shadow = 1.0f;
            float sp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (p.flags & (PBRK_SHADE_SHADOWS | PBRK_SHADE_SHAFTS))
                mat_mul(p.ssw, P.x + N.x * 0.1f, P.y + N.y * 0.1f, P.z + N.z * 0.1f, 1.0f, sp);   // :596-597 sun_space_from_world
            if (p.flags & PBRK_SHADE_SHADOWS) {
                const float px_size = 1.0f / 2048.0f;                                      // :594 (a constant of the shader, not the map's size)
                float sx = sp[0] * 0.5f + 0.5f, sy = sp[1] * 0.5f + 0.5f, sz = sp[2];
                sx = sx + (2.0f * (noise_2 - 0.5f)) * px_size;                             // :600
                sy = sy + (2.0f * (noise_1 - 0.5f)) * px_size;
                float acc = 0.0f;
                acc = acc + shadow_sample(p.sun_depth, p.sun_w, p.sun_h, sx + 0.75f * px_size, sy + 0.25f * px_size, sz);
                acc = acc + shadow_sample(p.sun_depth, p.sun_w, p.sun_h, sx + -0.25f * px_size, sy + 0.75f * px_size, sz);
                acc = acc + shadow_sample(p.sun_depth, p.sun_w, p.sun_h, sx + 0.25f * px_size, sy + -0.75f * px_size, sz);
                acc = acc + shadow_sample(p.sun_depth, p.sun_w, p.sun_h, sx + -0.75f * px_size, sy + -0.25f * px_size, sz);
                shadow = acc * 0.25f;
            }
            if (p.flags & PBRK_SHADE_SHAFTS) {                                   // :622-651
                float cp4[4];
                mat_mul(p.ssw, cam.x, cam.y, cam.z, 1.0f, cp4);                          // :627
                f3 delta = mk3(sp[0] - cp4[0], sp[1] - cp4[1], sp[2] - cp4[2]);
                float dist = sqrtf(dot3(delta, delta));
                const float step = 1.0f / 16.0f;
                float travelled = 0.0f;
                travelled += step * noise_1;                                               // :638
                if (p.flags & PBRK_SHADE_SHADOWS) {                                      // visibility from the sun depth map (:644-646)
                    f3 stepv = mk3(step * (delta.x / dist), step * (delta.y / dist), step * (delta.z / dist));   // :635
                    f3 pos = mk3(cp4[0] + stepv.x * noise_1, cp4[1] + stepv.y * noise_1, cp4[2] + stepv.z * noise_1);   // :637
                    for (int it = 0; it < 4096; ++it) {                                  // bounded like below
                        pos = add3(pos, stepv);
                        travelled += step;
                        if (travelled > dist) break;
                        float vis = shadow_sample(p.sun_depth, p.sun_w, p.sun_h, pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z);
                        outl.x += 0.001f * vis * sun_emission.x;
                        outl.y += 0.001f * vis * sun_emission.y;
                        outl.z += 0.001f * vis * sun_emission.z;
                    }
                } else {
                    // bounded: non-sky pixels lie within +-99 world units => dist < 16 in sun space
                    for (int it = 0; it < 4096; ++it) {
                        travelled += step;
                        if (travelled > dist) break;
                        outl.x += 0.001f * 1.0f * sun_emission.x;
                        outl.y += 0.001f * 1.0f * sun_emission.y;
                        outl.z += 0.001f * 1.0f * sun_emission.z;
                    }
                }
            }

            // :657-661
            f3 F0 = mk3(mix_(0.04f, base.x, metallic), mix_(0.04f, base.y, metallic), mix_(0.04f, base.z, metallic));
            f3 kS = fresnel_schlick(fmaxf(dot3(N, V), 0.0f), F0);
            f3 kD = mk3((1.0f - kS.x) * (1.0f - metallic), (1.0f - kS.y) * (1.0f - metallic), (1.0f - kS.z) * (1.0f - metallic));

            // :664-679
            {
                f3 L = mk3(-p.sun[0], -p.sun[1], -p.sun[2]);
                f3 Hs = add3(L, V);
                f3 H = kGI ? normalize3(Hs) : normalize3_shared(Hs);   // no rsq here: the GGX lobe amplifies an error in N.H by 1/a^2 (1300x at roughness 1/6)
                float NdotL = fmaxf(dot3(N, L), 0.0f);
                if (NdotL > 0.0f) {
                    float VdotH = fmaxf(dot3(V, H), 0.0f);
                    float NdotH = fmaxf(dot3(N, H), 0.0f);
                    float D = ggx_d(NdotH, roughness);
                    float rvh = __builtin_amdgcn_rcpf(VdotH);
                    float G = fminf(1.0f, fminf(2.0f * NdotH * VdotN * rvh, 2.0f * NdotH * NdotL * rvh));
                    f3 F = fresnel_schlick(VdotH, F0);
                    float rden = __builtin_amdgcn_rcpf(fmaxf(4.0f * NdotL * VdotN, 0.0001f));
                    const float rpi = 1.0f / PBR_PI;
                    outl.x += shadow * (kD.x * base.x * rpi + F.x * G * D * rden) * sun_emission.x * NdotL;
                    outl.y += shadow * (kD.y * base.y * rpi + F.y * G * D * rden) * sun_emission.y * NdotL;
                    outl.z += shadow * (kD.z * base.z * rpi + F.z * G * D * rden) * sun_emission.z * NdotL;
                }
            }

            const bool gi = kGI;
            if ((p.flags & PBRK_SHADE_IBL) || gi) {
                float2 sb = lut_fetch(p.lut, p.lut_cells, p.lut_size, VdotN, fmaxf(roughness, 0.05f));  // :681
                f3 ambient = mk3(0.0f, 0.0f, 0.0f);
                if (p.flags & PBRK_SHADE_IBL) {                                            // :690 (commented line)
                    CubeST cs = cube_select(N);
                    float si, ti;
                    if (kGI) cube_st_exact(cs, &si, &ti); else cube_st_shared(cs, &si, &ti);
                    CubeTap tp = cube_tap_from_st(cs.face, si, ti, p.irr_size, p.snap != 0);
                    ambient = p.irr_cells ? fetch_rgb_cells_tap(p.irr_cells, p.irr_size, tp) : fetch_rgb_tap(p.irr, p.irr_size, tp);
                }
                float p0_view[4] = {0.0f, 0.0f, 0.0f, 1.0f};
                if (gi) {
                    float pv[4];
                    mat_mul(p.vfc, fs_u * 2.0f - 1.0f, fs_v * 2.0f - 1.0f, depth, 1.0f, pv);   // :446-447
                    p0_view[0] = pv[0] / pv[3]; p0_view[1] = pv[1] / pv[3]; p0_view[2] = pv[2] / pv[3]; p0_view[3] = pv[3] / pv[3];
                    // :546-577 random direction in the hemisphere around N
                    f3 some_vector = normalize3(mk3(0.7128864983f, 0.8217892113f, 0.948912748f));
                    f3 tangent = normalize3(cross3(some_vector, N));
                    f3 bitangent = cross3(N, tangent);
                    float pitch = acos_det(sqrtf(1.0f - noise_1));
                    float yaw = (2.0f * PBR_PI) * noise_3;
                    float sp_, cp_, sy_, cy_;
                    sincos_det(pitch, &sp_, &cp_); sincos_det(yaw, &sy_, &cy_);
                    float lx = sp_ * cy_, ly = sp_ * sy_, lz = cp_;
                    f3 bent = mk3((tangent.x * lx + bitangent.x * ly) + N.x * lz, (tangent.y * lx + bitangent.y * ly) + N.y * lz,
                                  (tangent.z * lx + bitangent.z * ly) + N.z * lz);
                    ambient = sample_radiance_ss(p, V, p0_view, P, bent, 12, 1.0f, noise_3, 0.5f, 0.75f);   // :685
                }
                outl.x += kD.x * ambient.x * base.x;                                       // :687
                outl.y += kD.y * ambient.y * base.y;
                outl.z += kD.z * ambient.z * base.z;
                // :693-697
                f3 I = mk3(-V.x, -V.y, -V.z);
                float dNI = dot3(N, I);
                f3 R = mk3(I.x - 2.0f * dNI * N.x, I.y - 2.0f * dNI * N.y, I.z - 2.0f * dNI * N.z);
                float jr = 0.6f * roughness;
                f3 Rj = mk3(R.x + jr * (noise_1 - 0.5f), R.y + jr * (noise_2 - 0.5f), R.z + jr * (noise_3 - 0.5f));
                R = kGI ? normalize3(Rj) : normalize3_shared(Rj);
                float r2 = roughness * roughness;
                float r4 = r2 * r2;
                R = mk3(mix_(R.x, N.x, r4), mix_(R.y, N.y, r4), mix_(R.z, N.z, r4));
                f3 spec = mk3(0.0f, 0.0f, 0.0f);
                if (p.flags & PBRK_SHADE_IBL) spec = pyramid_fetch<kGI>(p.pre, p.pre_cells, p.pre_cells_first, p.pre_size, p.pre_levels, R, roughness * 4.0f, level_tab, p.snap != 0);   // :699
                if (gi) spec = sample_radiance_ss(p, V, p0_view, P, R, 16, 2.0f, noise_3, roughness, 0.9f);   // :701
                outl.x += spec.x * (F0.x * sb.x + sb.y);                                   // :702
                outl.y += spec.y * (F0.y * sb.x + sb.y);
                outl.z += spec.z * (F0.z * sb.x + sb.y);
            }

            outl = add3(outl, emissive);                                                   // :706
        }
        outl = mk3(fmaxf(outl.x, 0.0f), fmaxf(outl.y, 0.0f), fmaxf(outl.z, 0.0f));     // :712

        if (p.out_fmt == PBRK_FMT_RGBA16F) {
            __half2 lo = __halves2half2(__float2half_rn(outl.x), __float2half_rn(outl.y));
            __half2 hi = __halves2half2(__float2half_rn(outl.z), __float2half_rn(1.0f));
            uint2 packed;
            packed.x = *reinterpret_cast<unsigned*>(&lo);
            packed.y = *reinterpret_cast<unsigned*>(&hi);
            ((uint2*)p.out)[pi] = packed;
        } else {
            ((float4*)p.out)[pi] = make_float4(outl.x, outl.y, outl.z, 1.0f);
        }
    }
}

__global__ __launch_bounds__(256) void k_lut_cells(const unsigned* __restrict__ lut, uint4* __restrict__ cells, int S) {
    int total = (S + 1) * (S + 1);
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        int ci = id % (S + 1), cj = id / (S + 1);
        int i0 = ci - 1, j0 = cj - 1;                       // tap origin; clamp-to-edge addressing
        int ia = min(max(i0, 0), S - 1), ib = min(max(i0 + 1, 0), S - 1);
        int ja = min(max(j0, 0), S - 1), jb = min(max(j0 + 1, 0), S - 1);
        cells[id] = make_uint4(lut[ja * S + ia], lut[ja * S + ib], lut[jb * S + ia], lut[jb * S + ib]);
    }
}
extern "C" int pbrk_lut_cells_build(const void* lut_half2, int size, void* cells_out, void* stream) {
    if (!lut_half2 || !cells_out || size < 1) return PBRK_E_ARG;
    int total = (size + 1) * (size + 1);
    hipLaunchKernelGGL(k_lut_cells, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned*)lut_half2, (uint4*)cells_out, size);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

// diagnostics (pbrk_debug_sample): the samplers above at arbitrary coordinates
__global__ void k_debug_sample_shade(int which, ShadeParams p, const float* __restrict__ coords, int count, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float a = coords[i * 3], b = coords[i * 3 + 1], c = coords[i * 3 + 2];
    if (which == 0) out[i] = grid_sample(p, a, b, c);
    else out[i] = make_float4(shadow_sample(p.sun_depth, p.sun_w, p.sun_h, a, b, c), 0.0f, 0.0f, 0.0f);
}
extern "C" __attribute__((visibility("hidden"))) int pbrk_debug_sample_post(const void* texture, int w, int h, const void* coords, int count, void* out, void* stream);
extern "C" int pbrk_debug_sample(int which, const void* texture, int w, int h, int d, const void* coords, int count, void* out, void* stream) {
    if (!texture || !coords || !out || count < 1 || w < 1 || h < 1 || d < 1 || which < 0 || which > 2) return PBRK_E_ARG;
    if (which == 2) return pbrk_debug_sample_post(texture, w, h, coords, count, out, stream);
    if (which == 0 && !(w == h && h == d && w <= 1024)) return PBRK_E_ARG;
    ShadeParams p;
    memset(&p, 0, sizeof p);
    p.grid = (const uint2*)texture; p.grid_n = w;
    p.sun_depth = (const float*)texture; p.sun_w = w; p.sun_h = h;
    hipLaunchKernelGGL(k_debug_sample_shade, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, which, p, (const float*)coords, count, (float4*)out);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

// Per-column / per-row tables of a frame size for k_shade_tile (lighting_pass.glsl:444-447, 456-459 and InterleavedGradientNoise
// :119-121), in the shader's operation order with correctly rounded fp32 operations:
//   col[x] = { (x + .5) / W * 2 - 1, .06711056 (x + .5), .06711056 ((x + .5) + 90), .06711056 ((x + .5) + 522) }
//   row[y] = { (y + .5) / H * 2 - 1, .00583715 (y + .5), .00583715 ((y + .5) + 20), .00583715 ((y + .5) + 55) }
// Built once per frame size (synchronous upload: never inside a stream capture -- pbrk_shade_tables_ready tells the backend).
#include <map>
#include <mutex>
#include <vector>
struct ShadeTables { float4* col = nullptr; float4* row = nullptr; };
static std::map<std::pair<int, int>, ShadeTables> g_shade_tables;
static std::mutex g_shade_tables_mu;
extern "C" int pbrk_shade_tables_ready(int width, int height) {
    std::lock_guard<std::mutex> lk(g_shade_tables_mu);
    return g_shade_tables.count({width, height}) ? 1 : 0;
}
static long long g_tile_min_pixels = -1;        // -1: PBR_SHADE_TILE_MIN_PIXELS or the default
static int g_fast_mode = -1;                    // -1: PBR_SHADE_FAST or the default (1)
extern "C" void pbrk_shade_set_fast(int on) { g_fast_mode = on; }
extern "C" void pbrk_shade_set_tile_min_pixels(long long pixels) { g_tile_min_pixels = pixels; }
static long long tile_min_pixels() {
    if (g_tile_min_pixels < 0) { const char* e = getenv("PBR_SHADE_TILE_MIN_PIXELS"); g_tile_min_pixels = e ? atoll(e) : (1ll << 62); }
    return g_tile_min_pixels;
}
// 1 when the next launch for this frame size would still have to build (allocate + upload) its tables
extern "C" int pbrk_shade_needs_tables(int width, int height) {
    static int tab = -1;
    if (tab < 0) { const char* e = getenv("PBR_SHADE_TABLES"); tab = e ? atoi(e) : 0; }
    return (tab || (long long)width * height >= tile_min_pixels()) && !pbrk_shade_tables_ready(width, height);
}
static bool shade_tables(int width, int height, const float4** col, const float4** row) {
    std::lock_guard<std::mutex> lk(g_shade_tables_mu);
    auto it = g_shade_tables.find({width, height});
    if (it == g_shade_tables.end()) {
        std::vector<float4> hc((size_t)width), hr((size_t)height);
        const float wf = (float)width, hf = (float)height;
        for (int x = 0; x < width; ++x) {
            const float fc = (float)x + 0.5f;
            const float u = fc / wf;
            float4 v; v.x = u * 2.0f - 1.0f; v.y = 0.06711056f * fc; v.z = 0.06711056f * (fc + 90.0f); v.w = 0.06711056f * (fc + 522.0f);
            hc[(size_t)x] = v;
        }
        for (int y = 0; y < height; ++y) {
            const float fc = (float)y + 0.5f;
            const float u = fc / hf;
            float4 v; v.x = u * 2.0f - 1.0f; v.y = 0.00583715f * fc; v.z = 0.00583715f * (fc + 20.0f); v.w = 0.00583715f * (fc + 55.0f);
            hr[(size_t)y] = v;
        }
        ShadeTables t;
        if (hipMalloc((void**)&t.col, hc.size() * sizeof(float4)) != hipSuccess) return false;
        if (hipMalloc((void**)&t.row, hr.size() * sizeof(float4)) != hipSuccess) { (void)hipFree(t.col); return false; }
        if (hipMemcpy(t.col, hc.data(), hc.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(t.row, hr.data(), hr.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(t.col); (void)hipFree(t.row); return false; }
        it = g_shade_tables.emplace(std::make_pair(width, height), t).first;
    }
    *col = it->second.col; *row = it->second.row;
    return true;
}

extern "C" int pbrk_shade(const PbrkShadeArgs* a, void* stream) {
    if (!a || a->width < 1 || a->height < 1) return PBRK_E_ARG;
    if (a->x0 < 0 || a->y0 < 0 || a->x1 > a->width || a->y1 > a->height || a->x0 >= a->x1 || a->y0 >= a->y1) return PBRK_E_ARG;
    if (!a->base_color || !a->normal || !a->orm || !a->emissive || !a->depth || !a->out) return PBRK_E_ARG;
    if (a->out_format != PBRK_FMT_RGBA16F && a->out_format != PBRK_FMT_RGBA32F) return PBRK_E_FORMAT;
    if (!a->prefiltered_bordered || a->prefiltered_size < 1 || a->prefiltered_levels < 1 ||
        a->prefiltered_levels > pbrk_mip_count(a->prefiltered_size, a->prefiltered_size) || a->prefiltered_levels > 16) return PBRK_E_ARG;   // sky branch is live code; 16-entry level table
    if (a->flags & PBRK_SHADE_IBL) {
        if (!a->irradiance_bordered || a->irradiance_size < 1 || !a->lut || a->lut_size < 1) return PBRK_E_ARG;
    }
    if (a->flags & PBRK_SHADE_GI) {
        if (!a->lightgrid || a->lightgrid_size < 1 || a->lightgrid_size > 1024 || !a->lut || a->lut_size < 1) return PBRK_E_ARG;
        if (a->prev_frame_levels < 1 || a->prev_frame_levels > 8 || a->prev_frame_w < 1 || a->prev_frame_h < 1) return PBRK_E_ARG;
        for (int l = 0; l < a->prev_frame_levels; ++l) if (!a->prev_frame[l]) return PBRK_E_ARG;
    }
    if (a->flags & PBRK_SHADE_SHADOWS) {
        if (!a->sun_depth || a->sun_depth_w < 1 || a->sun_depth_h < 1 || (long long)a->sun_depth_w * a->sun_depth_h > (1ll << 30)) return PBRK_E_ARG;
    }
    ShadeParams p;
    p.width = a->width; p.height = a->height; p.x0 = a->x0; p.y0 = a->y0; p.w = a->x1 - a->x0; p.h = a->y1 - a->y0;
    p.base = (const uchar4*)a->base_color; p.normal = (const uchar4*)a->normal; p.orm = (const uchar4*)a->orm;
    p.emissive = (const uchar4*)a->emissive; p.depth = (const float*)a->depth;
    p.irr = (const float4*)a->irradiance_bordered; p.irr_size = a->irradiance_size;
    p.pre = (const float4*)a->prefiltered_bordered; p.pre_size = a->prefiltered_size; p.pre_levels = a->prefiltered_levels;
    p.lut = (const __half2*)a->lut; p.lut_size = a->lut_size;
    p.irr_cells = (const float4*)a->irradiance_cells; p.pre_cells = (const float4*)a->prefiltered_cells;
    p.pre_cells_first = a->prefiltered_cells_first; p.lut_cells = (const uint4*)a->lut_cells;
    p.out = a->out; p.out_fmt = a->out_format; p.flags = a->flags;
    for (int i = 0; i < 16; ++i) { p.wfc[i] = a->globals[32 + i]; p.ssw[i] = a->globals[96 + i]; }
    for (int i = 0; i < 3; ++i) { p.sun[i] = a->globals[128 + i]; p.cam[i] = a->globals[132 + i]; }
    p.frame_idx_mod_59 = a->globals[135];
    p.grid = (const uint2*)a->lightgrid; p.grid_n = a->lightgrid_size;
    for (int l = 0; l < 8; ++l) p.prev[l] = (const uint2*)a->prev_frame[l];
    p.prev_w = a->prev_frame_w; p.prev_h = a->prev_frame_h; p.prev_levels = a->prev_frame_levels;
    for (int i = 0; i < 16; ++i) { p.cfv[i] = a->globals[16 + i]; p.vfc[i] = a->globals[48 + i]; p.vfw[i] = a->globals[64 + i]; p.wfv[i] = a->globals[80 + i]; }
    p.lightgrid_scale = a->globals[136];
    p.sun_depth = (const float*)a->sun_depth; p.sun_w = a->sun_depth_w; p.sun_h = a->sun_depth_h;
    p.rcp_width = 1.0f / (float)a->width; p.rcp_height = 1.0f / (float)a->height;
    // fast instantiation: no sun shadows / GI, power-of-two prefiltered cube with a complete cells twin (and, IBL, the two other twins)
    if (g_fast_mode < 0) { const char* e = getenv("PBR_SHADE_FAST"); g_fast_mode = e ? atoi(e) : 1; }
    const int fast_mode = g_fast_mode;
    p.snap = pbrk_get_cube_sampler_snap();                          // diagnostic cube-sampler convention: general kernel only
    bool fast = fast_mode && !p.snap && !(a->flags & (PBRK_SHADE_GI | PBRK_SHADE_SHADOWS)) && p.pre_cells && p.pre_cells_first == 0 &&
                (a->prefiltered_size & (a->prefiltered_size - 1)) == 0 && a->prefiltered_size <= 512 &&
                (long long)a->width * a->height < (1ll << 29);
    if (fast && (a->flags & PBRK_SHADE_IBL)) fast = p.irr_cells && p.lut_cells && a->irradiance_size <= 512;
    if (fast) {
        size_t cb = 0;
        for (int l = 0; l < a->prefiltered_levels; ++l) { int n = a->prefiltered_size >> l; if (n < 1) n = 1; cb += pbrk_cells_bytes(n); }
        p.pre_cells_bytes = (int)cb;
        // tiled instantiation (LDS-staged windows of the prefiltered levels + per-column / per-row host tables): frames from
        // PBR_SHADE_TILE_MIN_PIXELS on (a tile must subtend a small solid angle for its windows to cover its taps); the choice
        // depends on the FRAME's size only, so a row-banded dispatch runs the same kernel as the whole frame
        if ((long long)a->width * a->height >= tile_min_pixels() && shade_tables(a->width, a->height, &p.col_tab, &p.row_tab)) {
            p.irr_nf = (float)p.irr_size; p.irr_off1 = 0.5f * p.irr_nf + 0.5f;
            p.noise_offset = (1000 * 1.61803398875f) * p.frame_idx_mod_59;
            p.pre_maxl = (float)(p.pre_levels - 1); p.pre_wf = (float)p.pre_size; p.lut_sf = (float)p.lut_size;
            p.dbg = 0; p.dbg_stats = nullptr;
#ifdef PBR_K5_DEBUG
            {   // diagnostics build: PBR_K5_DBG bits (k_shade_internal.h); counters printed at every 16th launch
                static int dbg = -1; static unsigned long long* stats = nullptr; static int launches = 0;
                if (dbg < 0) { const char* e = getenv("PBR_K5_DBG"); dbg = e ? atoi(e) : 0; }
                if ((dbg & 4) && !stats) { (void)hipMalloc((void**)&stats, 64); (void)hipMemset(stats, 0, 64); }
                p.dbg = dbg; p.dbg_stats = stats;
                if ((dbg & 4) && (++launches % 8) == 0) {
                    unsigned long long h[6]; (void)hipDeviceSynchronize(); (void)hipMemcpy(h, stats, 48, hipMemcpyDeviceToHost);
                    fprintf(stderr, "k5 tile: %llu fetching lanes, %.1f %% in window A, %.1f %% in window B; %llu waves, %.1f %% all-A, %.1f %% all-B\n",
                            h[0], 100.0 * h[1] / h[0], 100.0 * h[2] / h[0], h[3], 100.0 * h[4] / h[3], 100.0 * h[5] / h[3]);
                    (void)hipMemset(stats, 0, 64);
                }
            }
#endif
            return launch_shade_tile(p, (a->flags & PBRK_SHADE_IBL) != 0, (a->flags & PBRK_SHADE_SHAFTS) != 0, (hipStream_t)stream);
        }
        p.col_tab = nullptr; p.row_tab = nullptr;
        static int tab = -1;
        if (tab < 0) { const char* e = getenv("PBR_SHADE_TABLES"); tab = e ? atoi(e) : 0; }
        if (tab) (void)shade_tables(a->width, a->height, &p.col_tab, &p.row_tab);
        return launch_shade_fast(p, (a->flags & PBRK_SHADE_IBL) != 0, (a->flags & PBRK_SHADE_SHAFTS) != 0, (hipStream_t)stream);
    }
    if (a->flags & PBRK_SHADE_GI) hipLaunchKernelGGL(k_shade<true>, dim3((p.w + 31) / 32, (p.h + 7) / 8), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_shade<false>, dim3((p.w + 63) / 64, (p.h + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
