/*
 * pbr_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A scalar C restatement of the reference renderer's IBL-precompute and shade-pass arithmetic,
 * written from scratch by reading the reference's GLSL/C sources (cited per function in
 * pbr_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library; the product (libgpu_hip.so) never links or loads it.
 *
 * Pinning: all shader arithmetic -- the Monte-Carlo maths, the lighting pass including its sun-shadow, light-shaft and
 * voxel-GI blocks, the light-grid sweep, TAA resolve, bloom and tone-map passes -- is pinned by known-answer vectors obtained
 * by executing the reference's own shader text on the CPU (tests/golden/oracle_a_*.json|npy|npz, generator
 * oracle/gen_oracle_a.py, values also listed in SURVEY.md 8c); this library reproduces every one of them bit for bit.
 * What the reference delegates to the Vulkan driver and never tests is "parity unpinned" and fixed by the definitions in
 * pbr_oracle.c: texture filtering (seamless-cube bilinear, 2:1 blit, the 2-D / 3-D / shadow samplers with their 1/256
 * coordinate snap), render-target rounding (fp16 RTE, unorm8 RTE), additive blending at fp16, and the precision of
 * sin / cos / acos (fixed fp32 polynomials for the per-pixel trig of the GI block; libm for the host-side sample tables).
 */
#ifndef PBR_ORACLE_H
#define PBR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors RendererGlobalsBuffer (reference src/demo_pbr_renderer/render.h:122-136); 552 bytes.
 * Matrices are column-major float[16] (HMM_Mat4: Elements[col][row]). */
typedef struct OrcGlobals {
    float clip_space_from_world[16];
    float clip_space_from_view[16];
    float world_space_from_clip[16];
    float view_space_from_clip[16];
    float view_space_from_world[16];
    float world_space_from_view[16];
    float sun_space_from_world[16];
    float old_clip_space_from_world[16];
    float sun_direction[4];
    float camera_pos[3];
    float frame_idx_mod_59;
    float lightgrid_scale;
    uint32_t visualize_lightgrid;
} OrcGlobals;

enum {
    ORC_SHADE_IBL       = 1 << 0, /* ambient = irradiance(N), spec = prefiltered(R, rough*4) (lighting_pass.glsl:690,699) */
    ORC_SHADE_SHAFTS    = 1 << 1, /* light-shaft loop (lighting_pass.glsl:622-651); visibility == 1 unless ORC_SHADE_SHADOWS */
    ORC_SHADE_ANALYTIC  = 1 << 2, /* analytic stand-ins for the env/irradiance/prefiltered/LUT textures */
    ORC_SHADE_SHADOWS   = 1 << 3, /* sun shadow PCF (:594-608) and light-shaft visibility (:646) from sun_depth_map */
    ORC_SHADE_GI        = 1 << 4  /* the live ambient / specular terms: SampleRadianceWithScreenSpaceTrace (:273-424, :685, :701) */
};

void   orc_set_threads(int n);
int    orc_get_threads(void);

/* ---- A1: Radiance RGBE decode (stb_image.h:7130-7286 semantics) ---- */
/* Returns 0 on success. If out_rgba==NULL only w/h are reported. out_rgba holds w*h*4 floats. */
int    orc_rgbe_decode(const uint8_t* bytes, size_t n, int* w, int* h, float* out_rgba);

/* ---- A2: cube pyramid ---- */
int    orc_mip_count(int w, int h);                 /* gpu_vulkan.c:1344-1351 */
size_t orc_level_offset(int W, int level);          /* float offset of level in [level][face][y][x][4] */
size_t orc_pyramid_floats(int W);
/* cube sampler convention: 0 = exact fp32 tap weights (default), 1 = coordinates and LOD fraction snapped to 1/256 (see pbr_oracle.c) */
void   orc_set_cube_sampler_snap(int on);
int    orc_get_cube_sampler_snap(void);
void   orc_build_pyramid(float* pyr, int W);        /* level 0 present; fills levels 1.. (gpu_vulkan.c:1458-1483) */
/* one linear blit of whole subresources, all layers: the rule stated in pbr_oracle.c A2 (exact 2:1 blits of the chain use the box rule) */
void   orc_blit_linear(const float* src, int ns_w, int ns_h, float* dst, int nd_w, int nd_h, int layers);

/* ---- cube sampling: seamless bilinear, trilinear across integer levels ---- */
void   orc_face_dir(int face, float u, float v, float out[3]);   /* A3 */
void   orc_cube_sample(const float* pyr, int W, int levels, const float dir[3], float lod, float out[4]);
/* neighbour texel across a face edge (exactly one of i,j out of range): returns face', writes i',j' */
int    orc_cube_neighbor(int face, int n, int i, int j, int* ni, int* nj);
/* analytic environment of SURVEY 8c: (1+.5dx, 1+.5dy^2, 1+.5dz*dx, 1), d normalised */
void   orc_env_analytic(const float dir[3], float out[4]);

/* ---- extension (no reference counterpart): equirectangular RGBA32F [h][w][4] -> cube level 0 [6][size][size][4] ---- */
void   orc_equirect_to_cube(const float* equirect, int w, int h, int size, float* out);

/* ---- N2: voxel light-grid sweep, lightgrid_sweep.glsl:9-75 ----
 * img: RGBA16F [d][h][w][4] (fp16 bit patterns), updated in place.  One invocation per (iy, iz) with
 * iy < ny, iz < nz handles a 128-voxel line: direction 0 -> voxels (x, iy, iz); 1 -> (iz, x, iy)
 * (base_coord.zxy, :15-18); 2 -> (iy, iz, x) (base_coord.yzx, :19-22).  imageLoad widens fp16 -> fp32,
 * imageStore rounds to nearest-even (the reference leaves the rounding to the driver: unpinned). */
#define ORC_SWEEP_LEN 128
void   orc_lightgrid_sweep(uint16_t* img, int w, int h, int d, int direction, int ny, int nz);

/* ---- N3: post-process tail -- taa_resolve.glsl:180-287, final_post_process.glsl:2-10,31-34 ----
 * 2-D sampler of these passes (SAMPLER_LINEAR_CLAMP; the reference leaves filtering to the driver: unpinned).  Defined
 * here like the hardware the reference targets: texel coordinates are snapped to 1/256 texel (Vulkan
 * subTexelPrecisionBits = 8) before the bilinear split, so a tap aimed at a texel centre returns that texel exactly;
 * weights and lerps are then exact fp32 ( a + t*(b-a), x then y ), edges clamp. */
enum { ORC_TEX_RGBA16F = 0, ORC_TEX_RG16F = 1, ORC_TEX_R32F = 2, ORC_TEX_RGBA32F = 3 };
typedef struct OrcTex2D { const void* data; int format, width, height; } OrcTex2D;
void   orc_tex2d_sample(const OrcTex2D* t, float u, float v, float out[4]);   /* missing channels: 0,0,0,1 */

typedef struct OrcTaaInputs {
    OrcTex2D lighting_result;      /* RGBA16F (render.cpp:693) */
    OrcTex2D gbuffer_depth;        /* D32F as R32F (render.cpp:686-687) */
    OrcTex2D gbuffer_velocity;     /* RG16F (render.cpp:690-691) */
    OrcTex2D gbuffer_velocity_prev;
    OrcTex2D prev_frame_result;    /* RGBA16F (render.cpp:696-697) */
} OrcTaaInputs;
/* out: float RGBA [y1-y0 rows are written in place][width][4] of a full-size array; gl_FragCoord = (x+.5, y+.5) */
void   orc_taa_resolve(const OrcTaaInputs* in, int width, int height, int y0, int y1, float* out_rgba);
/* fs_uv = ((x+.5)/width, (y+.5)/height) (full-screen triangle, final_post_process.glsl:16-19); out float RGBA */
void   orc_final_post_process(const OrcTex2D* bloom_result, int width, int height, int y0, int y1, float* out_rgba);
/* bloom_downsample.glsl:38-98 (13 bilinear taps, firefly clamp when dst_mip_level == 1) and bloom_upsample.glsl:23-58
 * (3x3 tent, radius 1.5 source texels, factor 0.06 when dst_mip_level == 0): the fragment colour for every pixel of a
 * dw x dh target, fs_uv = ((x+.5)/dw, (y+.5)/dh).  Blending and the RGBA16F store are the caller's (render.cpp:1139-1176). */
void   orc_bloom_downsample(const OrcTex2D* src, int dw, int dh, int dst_mip_level, float* out_rgba);
void   orc_bloom_upsample(const OrcTex2D* src, int dw, int dh, int dst_mip_level, float* out_rgba);
/* render-target conversion of a float colour to 8-bit unorm (round to nearest even of clamp(v,0,1)*255) */
uint8_t orc_unorm8(float v);

/* ---- A4/A5: per-sample tables (Fibonacci hemisphere; Beckmann weights) ---- */
/* cs[i*4+0..3] = cos(pitch_i), sin(pitch_i), cos(yaw_i), sin(yaw_i)  (gen_prefiltered_env_map.glsl:125-128) */
void   orc_sample_angles(int nsamples, float* cs);
/* D_i = DistributionBeckmann(cos(pitch_i*0.5), roughness) (gen_prefiltered_env_map.glsl:86-91,141) */
void   orc_prefilter_D(int nsamples, float roughness, float* D);

/* ---- A5: specular prefilter, one mip ---- */
/* pyr==NULL -> analytic env.  mip==0 -> copy of env LOD `src_lod` (reference: 1.0).
 * Writes faces [face0,face1) rows [y0,y1) into out (full [6][out_size][out_size][4] array). */
void   orc_prefilter_mip(const float* pyr, int W, int levels, int out_size, int mip, float roughness,
                         float src_lod, int nsamples, int literal,
                         int face0, int face1, int y0, int y1, float* out);

/* ---- A6: irradiance ---- */
void   orc_irradiance(const float* pyr, int W, int levels, int out_size, float src_lod, int nsamples,
                      int literal, int face0, int face1, int y0, int y1, float* out);

/* ---- A7: split-sum LUT; out holds size*size*2 floats (scale,bias), rows [y0,y1) ---- */
void   orc_brdf_lut(int size, int nsamples, int y0, int y1, float* out_rg);

/* ---- fp16 helpers (RTE) ---- */
uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);

/* ---- A8: shade pass ---- */
typedef struct OrcShadeInputs {
    int width, height;
    const uint8_t* base_color;   /* [H][W][4] unorm8 */
    const uint8_t* normal;       /* [H][W][4] unorm8 */
    const uint8_t* orm;          /* [H][W][4] unorm8 */
    const uint8_t* emissive;     /* [H][W][4] unorm8 */
    const float*   depth;        /* [H][W] */
    const float*   irradiance;   /* cube, 1 level, RGBA32F; may be NULL with ORC_SHADE_ANALYTIC or without IBL */
    int            irradiance_size;
    const float*   prefiltered;  /* cube pyramid RGBA32F */
    int            prefiltered_size, prefiltered_levels;
    const uint16_t* lut;         /* [S][S][2] fp16 */
    int            lut_size;
    /* inputs of the live shader's raster-fed blocks; each is read only under its flag */
    OrcTex2D       sun_depth_map;   /* R32F (render.cpp:676 D32F 2048^2): ORC_SHADE_SHADOWS, lighting_pass.glsl:594-608 and :646 */
    const uint16_t* lightgrid;      /* RGBA16F [n][n][n][4] (render.cpp:678, after the sweeps): ORC_SHADE_GI */
    int            lightgrid_size;
    const OrcTex2D* prev_frame;     /* PREV_FRAME_RESULT: RGBA16F mip chain (the reference binds bloom_downscale_rt, render.cpp:862) */
    int            prev_frame_levels;
} OrcShadeInputs;

/* ---- N4: voxel-GI sampling, lighting_pass.glsl:273-424 + :480-562 (bent normal), :685, :701 ----
 * sin / cos / acos of the per-pixel noise (bent normal, :566-577) decide where rays go and therefore which side of the
 * alpha > 0.3 / depth thresholds they land on; libm's last-bit differences would flip those branches, so this block is
 * defined on the fixed fp32 polynomials below (Cephes-style; |err| < 2e-7), shared by oracle, shim (Oracle-A) and kernel. */
float  orc_sinf_det(float x);       /* x in [0, 2 pi] */
float  orc_cosf_det(float x);
float  orc_acosf_det(float x);      /* x in [0, 1] */
/* texture(sampler3D(LIGHTGRID, SAMPLER_LINEAR_CLAMP), p): trilinear, clamp, coordinates snapped to 1/256 texel; x, y, z order */
void   orc_tex3d_sample(const uint16_t* grid, int n, const float p[3], float out[4]);
/* texture(sampler2D(GBUFFER_DEPTH, SAMPLER_NEAREST_CLAMP), uv).r */
float  orc_tex2d_nearest_r32f(const float* d, int w, int h, float u, float v);
/* textureLod(sampler2D(PREV_FRAME_RESULT, SAMPLER_LINEAR_CLAMP), uv, lod): bilinear per level, linear between levels, lod clamped */
void   orc_tex2d_sample_lod(const OrcTex2D* levels, int nlevels, float u, float v, float lod, float out[4]);
/* how often the trace left through each exit since the last reset: 0 off-screen fallback (:322-330), 1 screen-space hit (:372-382),
 * 2 no open point (:400-403), 3 voxel march (:411-422) */
void   orc_gi_exit_counts(uint64_t out[4], int reset);

/* sampler2DShadow with SAMPLER_PERCENTAGE_CLOSER (render.cpp:664-673: linear, clamp, compare Less): each of the four
 * bilinear taps contributes (ref < texel ? 1 : 0); coordinates snapped to 1/256 texel like orc_tex2d_sample.  Unpinned. */
float  orc_shadow_sample(const OrcTex2D* depth_map, float u, float v, float ref);

/* bilinear, clamp-to-edge fetch of an RG16F texture (lighting_pass.glsl:681 BRDF_INTEGRATION_MAP) */
void   orc_lut_sample(const uint16_t* lut, int size, float u, float v, float out[2]);

/* out_rgba: [H][W][4] fp32 (pre-quantisation); rows [y0,y1), columns [x0,x1) are written. */
void   orc_shade(const OrcGlobals* g, const OrcShadeInputs* in, int flags,
                 int x0, int x1, int y0, int y1, float* out_rgba);

#ifdef __cplusplus
}
#endif
#endif
