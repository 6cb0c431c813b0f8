/*
 * pbr_kernels.h -- low-level C ABI of the gfx950 kernels (device pointers + stream in, status out).
 *
 * This is the layer *under* the GPU_* boundary of include/gpu_hip.h: every entry point takes
 * plain device pointers and sizes, launches asynchronously on `stream` (a hipStream_t passed as
 * void*) and returns 0 or a negative PBRK_E_* code after validating shapes on the host (no launch
 * happens when validation fails).  Each entry names the reference code it replaces.
 *
 * Memory layouts (all tightly packed, little endian):
 *   cube level        float4 [6][n][n]                (RGBA32F, face order +X,-X,+Y,-Y,+Z,-Z)
 *   cube pyramid      levels 0..L-1 back to back, level l has n_l = max(1, W >> l)
 *   bordered level    float4 [6][n+2][n+2]            (apron = adjacent faces' edge texels)
 *   bordered pyramid  bordered levels back to back
 *   G-buffer planes   uchar4 [H][W] x4, float [H][W]  (base colour, normal, ORM, emissive, depth)
 *   lit frame         half4 [H][W] or float4 [H][W]
 */
#ifndef PBR_KERNELS_H
#define PBR_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    PBRK_OK = 0,
    PBRK_E_ARG = -1,        /* null pointer / non-positive size / bad range */
    PBRK_E_FORMAT = -2,     /* unsupported pixel format */
    PBRK_E_LAUNCH = -3      /* hipGetLastError() != hipSuccess after the launch */
};

enum {                      /* output formats understood by the kernels */
    PBRK_FMT_RG16F = 1,
    PBRK_FMT_RG32F = 2,
    PBRK_FMT_RGBA16F = 3,
    PBRK_FMT_RGBA32F = 4,
    PBRK_FMT_R32F = 5,      /* depth planes as sampled by the post-process passes */
    PBRK_FMT_RGBA8UN = 6,
    PBRK_FMT_BGRA8UN = 7
};

enum {                      /* shade flags (reference lighting_pass.glsl sub-blocks) */
    PBRK_SHADE_IBL = 1 << 0,     /* ambient = irradiance(N); spec = prefiltered(R, rough*4)  (:690, :699) */
    PBRK_SHADE_SHAFTS = 1 << 1,  /* light-shaft loop (:622-651); visibility == 1 unless PBRK_SHADE_SHADOWS */
    PBRK_SHADE_SHADOWS = 1 << 2, /* sun shadow: 4 PCF taps of the sun depth map (:594-608) + shaft visibility (:646) */
    PBRK_SHADE_GI = 1 << 3       /* the live ambient / specular terms: SampleRadianceWithScreenSpaceTrace (:273-424, :685, :701) */
};

/* sizes / offsets of the pyramid layouts, in float4 texels */
size_t pbrk_level_offset(int W, int level);
size_t pbrk_pyramid_texels(int W, int levels);
size_t pbrk_bordered_level_offset(int W, int level);
size_t pbrk_bordered_pyramid_texels(int W, int levels);
int    pbrk_mip_count(int w, int h);                    /* src/gpu/gpu_vulkan.c:1344-1351 */

/* ---- host-side tables (libm on the host, so CPU and GPU agree bit-for-bit on directions/weights) -----
 * angles[i] = (cos pitch_i, sin pitch_i, cos yaw_i, sin yaw_i), the Fibonacci hemisphere of
 * gen_prefiltered_env_map.glsl:125-128 / gen_irradiance_map.glsl:85-88 / gen_brdf_integration_map.glsl:171-174 */
void   pbrk_host_sample_angles(int nsamples, float* angles4);
/* Prefilter table (gen_prefiltered_env_map.glsl:122-143): entries (lx, ly, lz, w) for the samples
 * whose weight w = D_i*cos(pitch_i)*dw is non-zero, in index order.  Returns the entry count;
 * *alpha receives the texel-independent alpha channel sum_i(w_i)/PI evaluated in shader order. */
int    pbrk_host_prefilter_table(int nsamples, float roughness, float* table4, float* alpha);
/* Irradiance table (gen_irradiance_map.glsl:84-96): entries (lx, ly, lz, cos pitch_i). */
int    pbrk_host_irradiance_table(int nsamples, float* table4);

/* ---- K2: mip chain, replaces GPU_OpGenerateMipmaps' per-(level,face) linear blits
 *      (src/gpu/gpu_vulkan.c:1458-1483, :2786-2826): level l = 2x2 box mean of level l-1, per face. */
int pbrk_mip_chain(void* pyramid, int W, int levels, void* stream);
/* Cube-sampler convention (the reference leaves it to the driver, gpu_vulkan.c:613-634).  0 (default): exact fp32
 * tap weights.  1: texel coordinates and the LOD fraction snapped to 1/256, the sub-texel / mip-fraction resolution of real
 * texture units and of this repo's 2-D / 3-D samplers.  With 1, K3 / K4a / K4b / K5 run their general kernels (the fast ones
 * implement the default only): a switch to measure how far the outputs move between the two, not a production mode. */
/* {min, max} over the RGB values of `texels` float4 texels as fp32 bit patterns in out2_device (two unsigned, initialised by the
 * caller to {0x7F800000, 0}); a negative or NaN input reports min = 0.  Used by the tolerance-budgeted sample cut. */
int pbrk_level_minmax(const void* level, size_t texels, void* out2_device, void* stream);
void pbrk_set_cube_sampler_snap(int on);
int pbrk_get_cube_sampler_snap(void);
/* one exact 2:1 linear blit (GPU_OpBlit, gpu_vulkan.c:2786-2826) of `nlayers` square RGBA32F layers of size ns */
int pbrk_box_downsample(const void* src, int ns, void* dst, int nlayers, void* stream);
/* linear blit between whole RGBA32F subresources of any two sizes (all `nlayers` layers): the resample an odd mip level and a
 * non-2:1 GPU_OpBlit need (vkCmdBlitImage, unnormalised linear filter, clamp to edge; rule stated in oracle/pbr_oracle.c A2) */
int pbrk_blit_linear(const void* src, int ns_w, int ns_h, void* dst, int nd_w, int nd_h, int nlayers, void* stream);
/* GPU_OpClearColorF / GPU_OpClearColorI [gpu.h]: `bytes` at dst filled with a texel pattern of 1, 2, 4, 8 or 16 bytes (dst and bytes
 * multiples of the pattern size), one launch. */
int pbrk_fill_pattern(void* dst, unsigned long long bytes, const void* pattern, int pattern_bytes, void* stream);

/* ---- border build: pyramid -> bordered pyramid (seamless-cube apron; sampler state of
 *      src/gpu/gpu_vulkan.c:613-634 applied to a cube view). */
int pbrk_border_build(const void* pyramid, void* bordered, int W, int levels, void* stream);
/* the same for levels [level0, level1) of a `levels`-deep chain only (the other levels of `bordered` are left alone) */
int pbrk_border_build_range(const void* pyramid, void* bordered, int W, int levels, int level0, int level1, void* stream);

/* ---- K6 (extension): equirectangular RGBA32F panorama [h][w] -> cube level 0 [6][size][size].
 * Z-up: u = atan2(y,x)/2pi + .5 (wraps), v = acos(z/|d|)/pi (clamps); bilinear; angles in fp64. */
int pbrk_equirect_to_cube(const void* equirect_rgba32f, int w, int h, void* cube_level0, int size, void* stream);

/* ---- K7: light-grid sweep, shaders/lightgrid_sweep.glsl:9-75 (dispatch render.cpp:1064-1072).
 * image: RGBA16F [d][h][w], updated in place.  Invocations (iy, iz) in [y0,y1) x [z0,z1) each own one line of
 * PBRK_SWEEP_LEN voxels: direction 0 -> (x, iy, iz), 1 -> (iz, x, iy), 2 -> (iy, iz, x).  The line axis must be at
 * least PBRK_SWEEP_LEN long and the two ranges must lie inside the other two axes (checked; PBRK_E_ARG). */
#define PBRK_SWEEP_LEN 128
int pbrk_lightgrid_sweep(void* image_rgba16f, int w, int h, int d, int direction, int y0, int y1, int z0, int z1, void* stream);

/* ---- K1: split-sum BRDF LUT (shaders/gen_brdf_integration_map.glsl:142-210).
 * angles4: device copy of pbrk_host_sample_angles(nsamples); view_cs: device float2[size] with
 * (cos, sin) of acos((x+.5)/size) computed on the host.  Rows [y0,y1) are written. */
int pbrk_brdf_lut(void* out, int out_format, int size, int nsamples, const void* angles4,
                  const void* view_cs, int y0, int y1, void* stream);

/* ---- K4a: prefilter mip 0 = bilinear copy of one env level (gen_prefiltered_env_map.glsl:112-114).
 * src_bordered_level: bordered level of size n_src.  out: cube level [6][out_size][out_size]. */
int pbrk_prefilter_copy(const void* src_bordered_level, int n_src, void* out, int out_size,
                        int face0, int face1, int y0, int y1, void* stream);

/* ---- K4b / K3: Monte-Carlo hemisphere filter of one output level.
 * out.rgb = (sum_i w_i * bilinear(src, frame(texel) * l_i)) / divisor ; out.a = alpha.
 * Prefilter (gen_prefiltered_env_map.glsl:115-146): table from pbrk_host_prefilter_table, divisor PI.
 * Irradiance (gen_irradiance_map.glsl:81-97): table from pbrk_host_irradiance_table, divisor N, alpha 0. */
/* src_cells (optional, may be NULL): the same level in the 2x2-footprint "cells" layout built by pbrk_cells_build
 * (48 B per tap position, 3 loads per sample instead of 4; the vector-memory instruction rate bounds this kernel). */
size_t pbrk_cells_bytes(int n_src);
int pbrk_cells_build(const void* bordered_level, int n_src, void* cells, void* stream);
int pbrk_mc_filter(const void* src_bordered_level, const void* src_cells, int n_src, const void* table4, int n_entries,
                   float divisor, float alpha, void* out, int out_size,
                   int face0, int face1, int y0, int y1, void* stream);

/* self-check of the region kernel (PBR_MC_STATS=1): {wave-slices whose sample count came up short and were recomputed with
 * direct loads, all wave-slices}; the first must stay 0.  reset != 0 clears the counters after reading. */
int pbrk_mc_region_stats(unsigned long long* out2, int reset);
/* K4b / K3 kernel choice for tests and A-B runs: region = 0 skips the region kernel, lds = 0 the level-in-LDS kernel (the direct kernel
 * then serves every level); -1 = the environment (PBR_MC_REGION, PBR_MC_LDS) or the default 1.  Results agree to the order of the fp32 sums. */
void pbrk_mc_set_kernels(int region, int lds);
/* the binning's yield with the same switch: {(region, sample) flags set, samples x tiles, regions visited} summed over all tiles since the last reset */
int pbrk_mc_region_flag_stats(unsigned long long* out3);
/* samples that binning proved to tap one region from every texel of their tile (they run the body without tests), summed over tiles */
int pbrk_mc_region_window_stats(unsigned long long* out1);

/* ---- K5: deferred shade pass (shaders/lighting_pass.glsl:432-716, in-scope sub-blocks). */
typedef struct PbrkShadeArgs {
    int width, height;
    int x0, x1, y0, y1;                 /* pixel rectangle to shade */
    const void* base_color;             /* uchar4 [H][W] */
    const void* normal;
    const void* orm;
    const void* emissive;
    const void* depth;                  /* float [H][W] */
    const void* irradiance_bordered;    /* bordered level, size irradiance_size (IBL mode) */
    int irradiance_size;
    const void* prefiltered_bordered;   /* bordered pyramid of the prefiltered cube */
    int prefiltered_size, prefiltered_levels;
    const void* lut;                    /* half2 [S][S] */
    int lut_size;
    /* optional gather-saving twins (NULL = not available): cells of the irradiance level; cells of the prefiltered
     * levels >= prefiltered_cells_first (back to back, pbrk_cells_bytes each); 2x2-footprint cells of the LUT */
    const void* irradiance_cells;
    const void* prefiltered_cells;
    int prefiltered_cells_first;
    const void* lut_cells;              /* uint4 [(S+1)][(S+1)]: {t00,t10,t01,t11} half2, tap origin (-1,-1), clamp-to-edge */
    const void* sun_depth;              /* float [sun_depth_h][sun_depth_w] (SUN_DEPTH_MAP, render.cpp:676); PBRK_SHADE_SHADOWS */
    int sun_depth_w, sun_depth_h;
    const void* lightgrid;              /* half4 [n][n][n] (LIGHTGRID, render.cpp:678, after the sweeps); PBRK_SHADE_GI */
    int lightgrid_size;
    const void* prev_frame[8];          /* PREV_FRAME_RESULT mip chain, half4 per level (the reference binds bloom_downscale_rt, render.cpp:862) */
    int prev_frame_w, prev_frame_h;     /* level 0 extent; level l is max(1, w >> l) x max(1, h >> l) */
    int prev_frame_levels;              /* 1..8 */
    void* out;                          /* half4 or float4 [H][W] */
    int out_format;                     /* PBRK_FMT_RGBA16F / PBRK_FMT_RGBA32F */
    int flags;                          /* PBRK_SHADE_* */
    float globals[138];                 /* RendererGlobalsBuffer, render.h:122-136 (552 bytes) */
} PbrkShadeArgs;
int pbrk_shade(const PbrkShadeArgs* args, void* stream);
/* The modes without sun shadows / voxel GI have two instantiations with bit-identical results: k_shade_fast (64 x 4 pixel
 * workgroups, every tap through the texture path; the default) and k_shade_tile (64 x 4 pixel tiles whose prefiltered taps come
 * from LDS-staged windows placed by the tile's centre pixel; measured slower on MI355X -- DESIGN.md K5 -- and therefore opt-in:
 * frames of at least `pixels` pixels take it; default: never, env PBR_SHADE_TILE_MIN_PIXELS; < 0 restores the default).  Both
 * read per-column / per-row tables of the frame size ((x + .5) / W * 2 - 1 and the interleaved-gradient-noise products, computed
 * on the host in the shader's operation order); pbrk_shade_tables_ready: they exist (the first launch of a frame size builds
 * them with a synchronous upload, so that launch cannot be part of a stream capture: pbrk_shade_needs_tables). */
void pbrk_shade_set_tile_min_pixels(long long pixels);
void pbrk_shade_set_fast(int on);                        /* 0: every mode through the general kernel k_shade (env PBR_SHADE_FAST); -1 = default */
int pbrk_shade_tables_ready(int width, int height);
int pbrk_shade_needs_tables(int width, int height);     /* the next launch of this frame size would build them (not capturable) */
/* LUT twin for K5: one 16-byte load per bilinear LUT fetch */
int pbrk_lut_cells_build(const void* lut_half2, int size, void* cells_out, void* stream);

/* ---- K8 / K9: post-process tail.  2-D sampler = linear clamp with coordinates snapped to 1/256
 *      texel (Vulkan subTexelPrecisionBits = 8), exact fp32 lerps. ---- */
typedef struct PbrkTex2D { const void* data; int format, width, height; } PbrkTex2D;     /* device pointer, PBRK_FMT_* */

/* K8: shaders/taa_resolve.glsl:180-287 (3x3 Mitchell-Netravali resolve, variance clamp of the Catmull-Rom-filtered
 * history, velocity / off-screen rejection).  All inputs and the target have the frame's size. */
typedef struct PbrkTaaArgs {
    PbrkTex2D lighting_result;          /* RGBA16F */
    PbrkTex2D gbuffer_depth;            /* R32F */
    PbrkTex2D gbuffer_velocity;         /* RG16F */
    PbrkTex2D gbuffer_velocity_prev;    /* RG16F */
    PbrkTex2D prev_frame_result;        /* RGBA16F */
    void* out;                          /* half4 or float4 [height][width] */
    int out_format;                     /* PBRK_FMT_RGBA16F / PBRK_FMT_RGBA32F */
    int width, height, y0, y1;          /* rows [y0,y1) */
} PbrkTaaArgs;
int pbrk_taa_resolve(const PbrkTaaArgs* args, void* stream);

/* K9: shaders/final_post_process.glsl:2-10,31-34: pow(aces_approx(2 * src), 1/2.2), alpha 1.  src may have another
 * size than the target (bilinear); 8-bit targets round to nearest even. */
typedef struct PbrkFinalArgs {
    PbrkTex2D src;                      /* RGBA16F (bloom / TAA result) */
    void* out;
    int out_format;                     /* PBRK_FMT_RGBA8UN / BGRA8UN / RGBA16F / RGBA32F */
    int width, height, y0, y1;
} PbrkFinalArgs;
int pbrk_final_post_process(const PbrkFinalArgs* args, void* stream);

/* K10 / K11: one pass of the bloom chain (render.cpp:1139-1176): bloom_downsample.glsl:38-98 (13 bilinear taps; firefly
 * clamp when dst_mip_level == 1) or bloom_upsample.glsl:23-58 (3x3 tent at radius 1.5 source texels; x0.06 when
 * dst_mip_level == 0) from `src` into an RGBA16F target of dst_width x dst_height.  blend_additive: target = fp16(colour +
 * target) with alpha = 1 (VK_BLEND_FACTOR_ONE / ONE, alpha ONE / ZERO; gpu_vulkan.c:1828-1842). */
typedef struct PbrkBloomArgs {
    PbrkTex2D src;                      /* RGBA16F (a mip of a texture: data points at the level, width/height are the level's) */
    void* dst;                          /* half4 [dst_height][dst_width] */
    int dst_width, dst_height;
    int dst_mip_level;                  /* the shader's push constant */
    int upsample;                       /* 0: bloom_downsample.glsl, 1: bloom_upsample.glsl */
    int blend_additive;
    int y0, y1;
    const void* blend_src;              /* blend_additive: the operand added to the pass's result, same layout as dst; NULL = dst itself */
} PbrkBloomArgs;
int pbrk_bloom_pass(const PbrkBloomArgs* args, void* stream);
/* The three instantiations of a bloom pass give the same bits: one pixel per thread; two (downsample) or 2 x 2 (upsample) pixels per thread
 * (exact 2 : 1 passes with even sizes and at least quad_min_pixels target pixels; default 100000, env PBR_BLOOM_QUAD_MIN_PIXELS); four lanes per pixel (general-sampler
 * passes of at most small_max_pixels target pixels; default 40000, env PBR_BLOOM_SMALL_MAX_PIXELS).  Negative = default. */
void pbrk_bloom_set_thresholds(long long quad_min_pixels, long long small_max_pixels);

/* ---- diagnostics: the device samplers of the widened passes evaluated at caller-supplied coordinates, so that tests can feed them
 *      NaN / inf / 1e30 / boundary values directly (a ray that has marched far away must never become an out-of-bounds read).
 *      which: 0 = LIGHTGRID (RGBA16F n^3, coords xyz), 1 = sampler2DShadow (R32F w x h, coords u, v, ref; result in out[0]),
 *      2 = the post-process 2-D sampler (RGBA16F w x h, coords u, v).  coords: device float[count][3]; out: device float[count][4]. */
int pbrk_debug_sample(int which, const void* texture, int w, int h, int d, const void* coords, int count, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
