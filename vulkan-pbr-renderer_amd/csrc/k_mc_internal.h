// k_mc_internal.h -- argument block shared by the Monte-Carlo filter kernels (k_mc.hip, k_mc_region.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "pbr_device.h"

struct McArgs {
    const float4* src; int n_src; unsigned src_bytes;
    const float4* cells; unsigned cells_bytes;          // optional 2x2-footprint layout: 3 loads per sample instead of 4
    const float4* tab; int n_tab;
    float divisor, alpha;
    float4* out; int size;
    int face0, y0, rows, tiles_x, tiles_per_face;
    int snap;                                           // host side: cube-sampler convention (pbrk_set_cube_sampler_snap) -> the SNAP instantiation of the direct kernel
};

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));


__device__ __forceinline__ f3 tap_rgb(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 0);
    return mk3(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
}

// "cells" layout (pbr_device.h, cells_bilerp): for every tap position (face, j0, i0), i0/j0 in [0, n], the 2x2 RGB footprint in
// coefficient form {t00, t10 - t00, t01, t11 - t01}, 48 contiguous bytes -> three 16-byte loads and 12 instructions per fetch.
typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f3 fetch_cells(__amdgpu_buffer_rsrc_t rc, int voff, float a, float b) {
    u32x4c A = __builtin_amdgcn_raw_buffer_load_b128(rc, voff, 0, 0);
    u32x4c Bq = __builtin_amdgcn_raw_buffer_load_b128(rc, voff + 16, 0, 0);
    u32x4c Cq = __builtin_amdgcn_raw_buffer_load_b128(rc, voff + 32, 0, 0);
    return cells_bilerp(make_float4(__uint_as_float(A.x), __uint_as_float(A.y), __uint_as_float(A.z), __uint_as_float(A.w)),
                        make_float4(__uint_as_float(Bq.x), __uint_as_float(Bq.y), __uint_as_float(Bq.z), __uint_as_float(Bq.w)),
                        make_float4(__uint_as_float(Cq.x), __uint_as_float(Cq.y), __uint_as_float(Cq.z), __uint_as_float(Cq.w)), a, b);
}

// one sample: direction L -> bilinear RGB of the bordered level behind `rs`
template <bool CELLS>
__device__ __forceinline__ f3 sample_bordered(__amdgpu_buffer_rsrc_t rs, f3 L, float nf, float off, int nb, int row_bytes, bool snap = false) {
    float fid = __builtin_amdgcn_cubeid(L.x, L.y, L.z);
    float sc = __builtin_amdgcn_cubesc(L.x, L.y, L.z);
    float tc = __builtin_amdgcn_cubetc(L.x, L.y, L.z);
    float ma2 = __builtin_amdgcn_cubema(L.x, L.y, L.z);          // 2 * major axis
    float h = __builtin_amdgcn_rcpf(fabsf(ma2)) * nf;            // n / (2 |rc|)
    float u = fmaf(sc, h, off);                                  // s*n - 0.5 + 1 (bordered), in [0.5, n + 0.5]
    float v = fmaf(tc, h, off);
    if (snap) { u = snap256(u); v = snap256(v); }                // bordered = unbordered + 1: the snap commutes with the offset
    float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
    if (CELLS) {
        // Cell byte offset in fp32: cells exist for n <= 512 only, so face*nc + j0, the cell index (< 6*513^2 < 2^24) and
        // 48*cell (= 16 * an integer < 2^24) are all exact -- six FMA-rate instructions instead of three conversions and three
        // integer multiplies (v_mul_lo_u32 / v_mad_u64_u32 issue at 1.45x / 1.7x the cost of an FMA here, tools/ubench_valu.hip).
        float ncf = (float)(nb - 1);                             // n + 1 tap positions per edge
        float cellf = fmaf(fmaf(fid, ncf, floorf(v)), ncf, floorf(u));
        return fetch_cells(rs, (int)(cellf * (float)PBR_CELL_BYTES), a, b);
    }
    int i0 = (int)u, j0 = (int)v, face = (int)fid;
    int voff = ((face * nb + j0) * nb + i0) << 4;
    f3 t00 = tap_rgb(rs, voff, 0), t10 = tap_rgb(rs, voff + 16, 0);
    f3 t01 = tap_rgb(rs, voff, row_bytes), t11 = tap_rgb(rs, voff + 16, row_bytes);
    f3 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
    r.z = lerp_fma(lerp_fma(t00.z, t10.z, a), lerp_fma(t01.z, t11.z, a), b);
    return r;
}

// k_mc_region.hip: taps from LDS-staged regions of the source level.  Returns false when the kernel does not apply
// (the caller then takes the direct kernel); the decision depends on the level's shape only, never on the dispatched range.
bool launch_mc_region(McArgs a, int nfaces, hipStream_t st);
extern int g_mc_region_mode, g_mc_lds_mode;         // pbrk_mc_set_kernels (k_mc_region.hip)
