// pbr_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Compiled with -ffp-contract=off: every FMA in this code base is written explicitly (fmaf /
// __builtin_fmaf).  Functions marked EXACT keep the operation order of the reference shader text
// (separately rounded IEEE fp32 ops, correctly rounded / and sqrt) so that quantities feeding
// discontinuous or ill-conditioned expressions (texel directions, tangent frames, noise, N.H near 1)
// come out bit-identical to a scalar CPU evaluation; the hot inner loops use explicit FMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PBR_PI 3.14159265358979323846f

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// EXACT: GLSL dot / cross / normalize in source order
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// Several quotients by one divisor: r = RN(1/d) once (correctly rounded divide), then per numerator
//   q = RN(a*r);  rem = a - q*d (exact, FMA);  result = RN(q + rem*r) = RN(a/d)            [Markstein 1990]
// which is the correctly rounded quotient whenever d's significand is not all ones (and nothing under/overflows): 3
// instructions per quotient instead of ~12.  The exception cannot occur for integer divisors (image extents) and is
// a 2^-23 event of 1 ulp elsewhere.
struct SharedRcp { float d, r; };
__device__ __forceinline__ SharedRcp shared_rcp(float d) { SharedRcp e; e.d = d; e.r = __fdiv_rn(1.0f, d); return e; }
__device__ __forceinline__ float div_by(float a, const SharedRcp& e) {
    float q = a * e.r;
    float rem = fmaf(-q, e.d, a);
    return fmaf(rem, e.r, q);
}
// EXACT: a / length(a) with three true divisions (bit-identical to a scalar CPU evaluation, always)
__device__ __forceinline__ f3 normalize3(f3 a) {
    float len = sqrtf(dot3(a, a));   // correctly rounded expansion (unlike __fsqrt_rn, which is the 1-ulp v_sqrt_f32)
    return mk3(__fdiv_rn(a.x, len), __fdiv_rn(a.y, len), __fdiv_rn(a.z, len));
}
// the same through one shared reciprocal: ~half the instructions, identical except when len's significand is all ones
// (2^-23 of the inputs, then 1 ulp) -- for the shade pass at 1e-4, not for lookups that must match bit for bit
__device__ __forceinline__ f3 normalize3_shared(f3 a) {
    SharedRcp e = shared_rcp(sqrtf(dot3(a, a)));
    return mk3(div_by(a.x, e), div_by(a.y, e), div_by(a.z, e));
}

// Correctly rounded reciprocal / square root / normalize through v_rcp_f32 / v_rsq_f32 (1 ulp) and ONE Newton step in FMAs:
// the corrected value carries a relative error of ~1e-14 before its final rounding, so the result is RN(1/d) / RN(sqrt(x))
// unless the exact value lies that close to a rounding boundary (about 2e-7 of all operands; the result is then one ulp off).
// 3 and 5 instructions instead of the ~10 / ~12 of the IEEE expansions.  Used where bit-equality with a CPU evaluation matters
// for conditioning, not for a branch (k_shade_fast, k_brdf_lut); the bit-exact kernels keep true divisions.
__device__ __forceinline__ float rcp_nr(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float sqrt_nr(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y, h = 0.5f * y;
    return fmaf(fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ f3 normalize3_nr(f3 a) {
    SharedRcp e; e.d = sqrt_nr(dot3(a, a)); e.r = rcp_nr(e.d);
    return mk3(div_by(a.x, e), div_by(a.y, e), div_by(a.z, e));
}
// EXACT: CubemapSampleDirFromFaceUV (reference shaders/gen_prefiltered_env_map.glsl:11-66):
// texel (ix,iy) of an n*n face -> unit direction through the texel centre.
__device__ __forceinline__ f3 face_texel_dir(int face, int ix, int iy, int n) {
    // power-of-two sizes (every size the reference and BASELINE use): x / n == x * (1/n) exactly
    float u, v;
    if ((n & (n - 1)) == 0) { float rn = 1.0f / (float)n; u = ((float)ix + 0.5f) * rn; v = ((float)iy + 0.5f) * rn; }
    else { u = __fdiv_rn((float)ix + 0.5f, (float)n); v = __fdiv_rn((float)iy + 0.5f, (float)n); }
    float sc = 2 * (u - 0.5f);
    float tc = 2 * (v - 0.5f);
    f3 r;
    switch (face) {
    case 0: r = mk3(1.0f, -tc, -sc); break;
    case 1: r = mk3(-1.0f, -tc, sc); break;
    case 2: r = mk3(sc, 1.0f, tc); break;
    case 3: r = mk3(sc, -1.0f, -tc); break;
    case 4: r = mk3(sc, -tc, 1.0f); break;
    default: r = mk3(-sc, -tc, -1.0f); break;
    }
    return normalize3(r);
}

// EXACT: tangent = normalize(cross(R, some_vector)) (gen_prefiltered_env_map.glsl:108-109)
__device__ __forceinline__ f3 tangent_of(f3 R) {
    return normalize3(cross3(R, mk3(12.123825810901f, 6.11831989512f, -5.12039214121f)));
}

// ---- cube addressing on the BORDERED layout -----------------------------------------------
// A bordered level stores each face as (n+2)*(n+2) float4 texels: the interior is the face, the
// one-texel apron holds the adjacent faces' edge texels (corners: mean of the three existing
// texels).  Seamless bilinear filtering then needs no branches: taps are (i0,j0)..(i0+1,j0+1)
// with i0 = floor(s*n - 0.5) + 1 in [0, n].
//
// Face selection follows the Vulkan table quoted in gen_prefiltered_env_map.glsl:12-23
// (major axis = largest magnitude, ties z > y > x).
struct CubeTap {
    int base;      // texel index of tap (i0, j0) inside the bordered level (face included)
    int face, i0, j0;
    float a, b;    // bilinear weights along x and y
};

// EXACT = false: one v_rcp_f32 + FMAs (Monte-Carlo inner loops: coordinate rounding noise averages out over
//                thousands of samples).
// EXACT = true : the projection in separately rounded IEEE ops, s = (0.5*sc)/|rc| + 0.5, u = s*n - 0.5
//                (single-sample lookups: K4a copy, shade pass) so that tap selection and weights are
//                bit-identical to a scalar CPU evaluation even next to a 5e4:1 HDR sun texel.
__device__ __forceinline__ float lerp_fma(float p, float q, float t) { return fmaf(t, q - p, p); }

// Face selection uses v_cubeid/sc/tc/ma_f32: the hardware implements exactly the table above (z >= x,y first,
// then y >= x; results are sign-flipped copies of the inputs, cubema = 2 * major axis), branch-free.
struct CubeST { int face; float sc, tc, ma; };
__device__ __forceinline__ CubeST cube_select(f3 d) {
    CubeST r;
    r.face = (int)__builtin_amdgcn_cubeid(d.x, d.y, d.z);
    r.sc = __builtin_amdgcn_cubesc(d.x, d.y, d.z);
    r.tc = __builtin_amdgcn_cubetc(d.x, d.y, d.z);
    r.ma = 0.5f * fabsf(__builtin_amdgcn_cubema(d.x, d.y, d.z));      // exact: cubema = 2*major
    return r;
}
// EXACT sampler coordinates s,t in [0,1] (independent of the level): s = (0.5*sc)/|rc| + 0.5
__device__ __forceinline__ void cube_st_exact(const CubeST& c, float* s, float* t) {
    *s = 0.5f * c.sc / c.ma + 0.5f;
    *t = 0.5f * c.tc / c.ma + 0.5f;
}
__device__ __forceinline__ void cube_st_shared(const CubeST& c, float* s, float* t) {     // see normalize3_shared
    SharedRcp rma = shared_rcp(c.ma);
    *s = div_by(0.5f * c.sc, rma) + 0.5f;
    *t = div_by(0.5f * c.tc, rma) + 0.5f;
}
// Sampler-coordinate convention: exact fp32 (default) or -- pbrk_set_cube_sampler_snap(1), general kernels only --
// snapped to 1/256 texel, the sub-texel resolution of real texture units (and of this repo's 2-D / 3-D samplers).
__device__ __forceinline__ float snap256(float x) { return floorf(x * 256.0f + 0.5f) * (1.0f / 256.0f); }
__device__ __forceinline__ CubeTap cube_tap_from_st(int face, float s, float t, int n, bool snap = false) {
    float u = s * (float)n - 0.5f;                        // unbordered coordinate, as the sampler definition states it
    float v = t * (float)n - 0.5f;
    if (snap) { u = snap256(u); v = snap256(v); }
    float fu = floorf(u), fv = floorf(v);
    int i0 = min(max((int)fu + 1, 0), n), j0 = min(max((int)fv + 1, 0), n);   // +1: bordered layout
    CubeTap tp;
    tp.a = u - fu; tp.b = v - fv;
    int nb = n + 2;
    tp.base = (face * nb + j0) * nb + i0;
    tp.face = face; tp.i0 = i0; tp.j0 = j0;
    return tp;
}

// EXACT = false: one v_rcp_f32 + FMAs (Monte-Carlo inner loops: coordinate rounding noise averages out over
//                thousands of samples).
// EXACT = true : the projection in separately rounded IEEE ops, s = (0.5*sc)/|rc| + 0.5, u = s*n - 0.5
//                (single-sample lookups: K4a copy, shade pass) so that tap selection and weights are
//                bit-identical to a scalar CPU evaluation even next to a 5e4:1 HDR sun texel.
template <bool EXACT>
__device__ __forceinline__ CubeTap cube_tap(f3 d, int n, bool snap = false) {
    CubeST c = cube_select(d);
    if (EXACT) {
        float s, t;
        cube_st_exact(c, &s, &t);
        return cube_tap_from_st(c.face, s, t, n, snap);
    }
    float h = 0.5f * __builtin_amdgcn_rcpf(c.ma) * (float)n;     // (0.5 / |rc|) * n
    float off = 0.5f * (float)n + 0.5f;                          // s*n - 0.5 + 1 = sc*h + n/2 + 0.5 (bordered)
    float u = fmaf(c.sc, h, off);
    float v = fmaf(c.tc, h, off);
    float fu = floorf(u), fv = floorf(v);
    int i0 = min(max((int)fu, 0), n), j0 = min(max((int)fv, 0), n);
    CubeTap tp;
    tp.a = u - fu; tp.b = v - fv;
    int nb = n + 2;
    tp.base = (c.face * nb + j0) * nb + i0;
    tp.face = c.face; tp.i0 = i0; tp.j0 = j0;
    return tp;
}

// fetches given a precomputed tap (lets a trilinear lookup share one face selection / projection)
__device__ __forceinline__ f3 fetch_rgb_tap(const float4* __restrict__ lvl, int n, const CubeTap& t) {
    int nb = n + 2;
    float4 t00 = lvl[t.base], t10 = lvl[t.base + 1];
    float4 t01 = lvl[t.base + nb], t11 = lvl[t.base + nb + 1];
    f3 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, t.a), lerp_fma(t01.x, t11.x, t.a), t.b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, t.a), lerp_fma(t01.y, t11.y, t.a), t.b);
    r.z = lerp_fma(lerp_fma(t00.z, t10.z, t.a), lerp_fma(t01.z, t11.z, t.a), t.b);
    return r;
}
// Stride of a cell in float4 units: 3 = packed (48 B).  4 (one 64-byte sector per cell, the fourth float4 padding, so that a
// scattered cell fetch never straddles two sectors) was measured on K5's prefiltered taps and is slower: 567 vs 540 us on the 8K
// frame -- a third more bytes through the caches outweighs the saved straddles.
#ifndef PBR_CELL_F4
#define PBR_CELL_F4 3
#endif
#define PBR_CELL_BYTES (16 * PBR_CELL_F4)
// One cell of the "cells" twin = the 2x2 RGB footprint of a tap position in coefficient form, 48 bytes:
//   A = (t00.rgb, d0.r)   B = (d0.gb, t01.rg)   C = (t01.b, d1.rgb)      d0 = t10 - t00, d1 = t11 - t01 (rounded once, at build time)
// fma(a, d0, t00) is lerp_fma(t00, t10, a) bit for bit (lerp_fma forms the same rounded difference), so a bilinear fetch from a
// cell equals the four-tap form exactly and costs 12 instructions instead of 18.
__device__ __forceinline__ f3 cells_bilerp(float4 A, float4 Bq, float4 Cq, float a, float b) {
    float tx = fmaf(a, A.w, A.x), ty = fmaf(a, Bq.x, A.y), tz = fmaf(a, Bq.y, A.z);            // row j0
    float bx = fmaf(a, Cq.y, Bq.z), by = fmaf(a, Cq.z, Bq.w), bz = fmaf(a, Cq.w, Cq.x);        // row j0 + 1
    return mk3(fmaf(b, bx - tx, tx), fmaf(b, by - ty, ty), fmaf(b, bz - tz, tz));
}
__device__ __forceinline__ f3 fetch_rgb_cells_tap(const float4* __restrict__ cells, int n, const CubeTap& t) {
    int nc = n + 1;
    const float4* c = cells + (size_t)((t.face * nc + t.j0) * nc + t.i0) * PBR_CELL_F4;
    return cells_bilerp(c[0], c[1], c[2], t.a, t.b);
}


// bilinear RGB fetch from a bordered level (global memory or LDS pointer)
template <bool EXACT>
__device__ __forceinline__ f3 cube_fetch_rgb(const float4* __restrict__ lvl, int n, f3 d) {
    CubeTap t = cube_tap<EXACT>(d, n);
    int nb = n + 2;
    float4 t00 = lvl[t.base], t10 = lvl[t.base + 1];
    float4 t01 = lvl[t.base + nb], t11 = lvl[t.base + nb + 1];
    f3 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, t.a), lerp_fma(t01.x, t11.x, t.a), t.b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, t.a), lerp_fma(t01.y, t11.y, t.a), t.b);
    r.z = lerp_fma(lerp_fma(t00.z, t10.z, t.a), lerp_fma(t01.z, t11.z, t.a), t.b);
    return r;
}

template <bool EXACT>
__device__ __forceinline__ float4 cube_fetch_rgba(const float4* __restrict__ lvl, int n, f3 d, bool snap = false) {
    CubeTap t = cube_tap<EXACT>(d, n, snap);
    int nb = n + 2;
    float4 t00 = lvl[t.base], t10 = lvl[t.base + 1];
    float4 t01 = lvl[t.base + nb], t11 = lvl[t.base + nb + 1];
    float4 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, t.a), lerp_fma(t01.x, t11.x, t.a), t.b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, t.a), lerp_fma(t01.y, t11.y, t.a), t.b);
    r.z = lerp_fma(lerp_fma(t00.z, t10.z, t.a), lerp_fma(t01.z, t11.z, t.a), t.b);
    r.w = lerp_fma(lerp_fma(t00.w, t10.w, t.a), lerp_fma(t01.w, t11.w, t.a), t.b);
    return r;
}

// bilinear RGB fetch from the "cells" twin of a level: cell (face, j0, i0), i0/j0 in [0, n] (bordered tap
// coordinates), holds the 2x2 RGB footprint in 48 contiguous bytes (coefficient form, cells_bilerp) -> 3 loads.
template <bool EXACT>
__device__ __forceinline__ f3 cube_fetch_rgb_cells(const float4* __restrict__ cells, int n, f3 d) {
    CubeTap t = cube_tap<EXACT>(d, n);
    int nc = n + 1;
    const float4* c = cells + (size_t)((t.face * nc + t.j0) * nc + t.i0) * PBR_CELL_F4;
    return cells_bilerp(c[0], c[1], c[2], t.a, t.b);
}

// XCD-aware remap of a 1-D block index (workgroups b and b+8 share an XCD/L2): consecutive
// logical tiles are placed on the same XCD so that tiles that share source texels share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    unsigned q = nblocks >> 3, r = nblocks & 7u;
    unsigned xcd = bid & 7u, idx = bid >> 3;
    unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}
