// k_shade_internal.h -- parameter block and small helpers shared by the shade kernels (k_shade.hip: the general / live
// instantiations; k_shade_fast.hip: the fast instantiation, its own translation unit because it is compiled without SLP packing).
#pragma once
#include "pbr_device.h"
#include "pbr_kernels.h"
#include <hip/hip_fp16.h>

struct ShadeParams {
    int width, height, x0, y0, w, h;
    const uchar4* base; const uchar4* normal; const uchar4* orm; const uchar4* emissive; const float* depth;
    const float4* irr; int irr_size;
    const float4* pre; int pre_size, pre_levels;
    const __half2* lut; int lut_size;
    const float4* irr_cells; const float4* pre_cells; int pre_cells_first; const uint4* lut_cells;
    int pre_cells_bytes;       // size of the prefiltered cells twin (fast instantiation: range-checked buffer loads)
    void* out; int out_fmt; int flags;
    float wfc[16];       // world_space_from_clip
    float ssw[16];       // sun_space_from_world (light shafts only)
    float sun[3], cam[3], frame_idx_mod_59;
    float rcp_width, rcp_height;   // RN(1/width), RN(1/height), computed on the host
    // tiled instantiation (k_shade_tile.hip): per-column / per-row tables of the frame and wave-uniform constants, all computed on the host
    const float4* col_tab; const float4* row_tab;
    float irr_nf, irr_off1, noise_offset, pre_maxl, pre_wf, lut_sf;
    int snap;                      // cube-sampler convention (pbrk_set_cube_sampler_snap): general kernel only
    int dbg; void* dbg_stats;      // -DPBR_K5_DEBUG builds only (tools/k5_tile_probe.sh): 1 = no staging, 2 = every lane reads LDS, 4 = window hit counters
    const float* sun_depth; int sun_w, sun_h;
    // voxel GI (PBRK_SHADE_GI)
    const uint2* grid; int grid_n;
    const uint2* prev[8]; int prev_w, prev_h, prev_levels;
    float vfw[16], cfv[16], vfc[16], wfv[16];   // view_space_from_world, clip_space_from_view, view_space_from_clip, world_space_from_view
    float lightgrid_scale;
};

// EXACT b / 255.0f for b in 0..255 in 3 instructions: one Newton correction of b * fl(1/255) is the correctly
// rounded quotient for all 256 inputs (checked exhaustively: tests/test_host_cpu.py::test_unorm8_decode_trick)
__device__ __forceinline__ float unorm8(unsigned b) {
    const float rc = 1.0f / 255.0f;
    float x = (float)b;
    float q = x * rc;
    float r = fmaf(-255.0f, q, x);
    return fmaf(r, rc, q);
}
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ void mat_mul(const float* m, float x, float y, float z, float w, float* o) {
    for (int r = 0; r < 4; ++r) o[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
__device__ __forceinline__ int cells_level_off(int W, int first, int level) {
    int off = 0;
    for (int l = first; l < level; ++l) { int n = max(W >> l, 1) + 1; off += 6 * n * n * PBR_CELL_F4; }
    return off;
}

// k_shade_fast.hip
int launch_shade_fast(const ShadeParams& p, bool ibl, bool shafts, hipStream_t stream);
// k_shade_tile.hip
int launch_shade_tile(const ShadeParams& p, bool ibl, bool shafts, hipStream_t stream);
