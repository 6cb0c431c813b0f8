/*
 * pbr_lighting.c -- host side of the deferred shade pass (C11): camera matrices, the Globals block,
 * and the lighting-pass objects / draw sequence of the reference renderer.
 *
 *   utils/camera.h:95-120          world/view/clip matrices (Z-up world, +Y-down view, LH zero-to-one clip)
 *   render.cpp:959-991             sun matrices + RendererGlobalsBuffer fill
 *   render.cpp:680-693             G-buffer + lighting result textures
 *   render.cpp:716-723, :829-871   lighting render pass, pipeline layout, descriptor set
 *   render.cpp:1119-1127           PrepareRenderPass .. Draw(3,1,0,0) .. EndRenderPass
 * Matrix code is written from the standard formulas (column-major, m[col*4+row]), all in fp32, in the operation order of the
 * reference's math library (third_party/HandmadeMath.h v2.0: linear-combination products, cross-product inverse, axis scaled by
 * 1/sqrt): the block equals the reference's bit for bit on the golden poses (tests/test_host_cpu.py), which matters for
 * world_space_from_clip -- it feeds the discontinuous sky test of lighting_pass.glsl:708.
 */
#include "pbr_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI32 3.14159265359f

typedef struct M4 { float m[16]; } M4;   /* column-major */

static M4 m4_identity(void) { M4 r; memset(&r, 0, sizeof r); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

static M4 m4_mul(const M4* a, const M4* b) {        /* a * b */
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row)
            r.m[c * 4 + row] = ((a->m[0 * 4 + row] * b->m[c * 4 + 0] + a->m[1 * 4 + row] * b->m[c * 4 + 1]) +
                                a->m[2 * 4 + row] * b->m[c * 4 + 2]) + a->m[3 * 4 + row] * b->m[c * 4 + 3];
    return r;
}

static M4 m4_translate(float x, float y, float z) { M4 r = m4_identity(); r.m[12] = x; r.m[13] = y; r.m[14] = z; return r; }

/* 4-component dot product as the reference's math library forms it on SSE hardware (its build target): two pairwise sums,
 * (x x' + y y') + (w w' + z z').  The scalar fallback of that library pairs differently; the renderer never takes it. */
static float dot4(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + (a[3] * b[3] + a[2] * b[2]); }

static void q_normalize(float q[4]) {
    float len = sqrtf(dot4(q, q));
    float inv = 1.0f / len;
    for (int i = 0; i < 4; ++i) q[i] *= inv;
}

static M4 m4_from_quat(const float qin[4]) {
    float q[4] = {qin[0], qin[1], qin[2], qin[3]};
    q_normalize(q);
    float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2];
    float xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    float wx = q[3] * q[0], wy = q[3] * q[1], wz = q[3] * q[2];
    M4 r = m4_identity();
    r.m[0] = 1.0f - 2.0f * (yy + zz); r.m[1] = 2.0f * (xy + wz);        r.m[2] = 2.0f * (xz - wy);
    r.m[4] = 2.0f * (xy - wz);        r.m[5] = 1.0f - 2.0f * (xx + zz); r.m[6] = 2.0f * (yz + wx);
    r.m[8] = 2.0f * (xz + wy);        r.m[9] = 2.0f * (yz - wx);        r.m[10] = 1.0f - 2.0f * (xx + yy);
    return r;
}

/* "true right-handed" zero-to-one perspective the reference selects at utils/camera.h:110-112 */
static M4 m4_perspective_lh_zo(float fov_rad, float aspect, float z_near, float z_far) {
    M4 r; memset(&r, 0, sizeof r);
    float cot = 1.0f / tanf(fov_rad / 2.0f);
    r.m[0] = cot / aspect;
    r.m[5] = cot;
    r.m[11] = 1.0f;
    r.m[10] = -(z_far / (z_near - z_far));
    r.m[14] = (z_near * z_far) / (z_near - z_far);
    return r;
}

static M4 m4_ortho_rh_zo(float l, float r_, float b, float t, float n, float f) {
    M4 r; memset(&r, 0, sizeof r);
    r.m[0] = 2.0f / (r_ - l);
    r.m[5] = 2.0f / (t - b);
    r.m[10] = 1.0f / (n - f);
    r.m[15] = 1.0f;
    r.m[12] = (l + r_) / (l - r_);
    r.m[13] = (b + t) / (b - t);
    r.m[14] = n / (n - f);
    return r;
}

static void v3_cross(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static float v3_dot(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* general 4x4 inverse through the 3-vector (cross-product) decomposition of the columns */
static M4 m4_inverse(const M4* M) {
    const float* c0 = &M->m[0]; const float* c1 = &M->m[4]; const float* c2 = &M->m[8]; const float* c3 = &M->m[12];
    float c01[3], c23[3], b10[3], b32[3];
    v3_cross(c0, c1, c01);
    v3_cross(c2, c3, c23);
    for (int i = 0; i < 3; ++i) { b10[i] = c0[i] * c1[3] - c1[i] * c0[3]; b32[i] = c2[i] * c3[3] - c3[i] * c2[3]; }
    float inv_det = 1.0f / (v3_dot(c01, b32) + v3_dot(c23, b10));
    for (int i = 0; i < 3; ++i) { c01[i] *= inv_det; c23[i] *= inv_det; b10[i] *= inv_det; b32[i] *= inv_det; }
    float t[3];
    M4 r;   /* rows of the inverse, written as columns and transposed at the end */
    v3_cross(c1, b32, t); for (int i = 0; i < 3; ++i) r.m[0 + i] = t[i] + c23[i] * c1[3];  r.m[3] = -v3_dot(c1, c23);
    v3_cross(b32, c0, t); for (int i = 0; i < 3; ++i) r.m[4 + i] = t[i] - c23[i] * c0[3];  r.m[7] = +v3_dot(c0, c23);
    v3_cross(c3, b10, t); for (int i = 0; i < 3; ++i) r.m[8 + i] = t[i] + c01[i] * c3[3];  r.m[11] = -v3_dot(c3, c01);
    v3_cross(b10, c2, t); for (int i = 0; i < 3; ++i) r.m[12 + i] = t[i] - c01[i] * c2[3]; r.m[15] = +v3_dot(c2, c01);
    M4 o;
    for (int c = 0; c < 4; ++c) for (int row = 0; row < 4; ++row) o.m[c * 4 + row] = r.m[row * 4 + c];
    return o;
}

static M4 m4_rotate_rh(float angle, float ax, float ay, float az) {
    float inv = 1.0f / sqrtf((ax * ax + ay * ay) + az * az);          /* the axis is scaled by the reciprocal of its length, not divided */
    ax *= inv; ay *= inv; az *= inv;
    float s = sinf(angle), c = cosf(angle), k = 1.0f - c;
    M4 r = m4_identity();
    r.m[0] = (ax * ax * k) + c;        r.m[1] = (ax * ay * k) + (az * s); r.m[2] = (ax * az * k) - (ay * s);
    r.m[4] = (ay * ax * k) - (az * s); r.m[5] = (ay * ay * k) + c;        r.m[6] = (ay * az * k) + (ax * s);
    r.m[8] = (az * ax * k) + (ay * s); r.m[9] = (az * ay * k) - (ax * s); r.m[10] = (az * az * k) + c;
    return r;
}

static const float kDegToRad = PI32 / 180.0f;

void PBR_FillGlobals(PBR_Globals* g, const float pos[3], const float ori_xyzw[4], float fov_degrees, float aspect,
                     float z_near, float z_far, float sun_angle_x_deg, float sun_angle_y_deg, uint32_t frame_idx) {
    float ori[4];
    if (ori_xyzw) memcpy(ori, ori_xyzw, sizeof ori);
    else {                                                    /* utils/camera.h:45: rotate to face +Y */
        float half = (-PI32 / 2.0f) / 2.0f;
        ori[0] = sinf(half); ori[1] = 0.0f; ori[2] = 0.0f; ori[3] = cosf(half);
    }
    /* utils/camera.h:103-120 (lazy pos/ori taken as converged) */
    M4 rot = m4_from_quat(ori);
    M4 tr = m4_translate(pos[0], pos[1], pos[2]);
    M4 world_from_view = m4_mul(&tr, &rot);
    float dq = dot4(ori, ori);
    float inv_ori[4] = {-ori[0] / dq, -ori[1] / dq, -ori[2] / dq, ori[3] / dq};
    M4 inv_rot = m4_from_quat(inv_ori);
    M4 inv_tr = m4_translate(pos[0] * -1.0f, pos[1] * -1.0f, pos[2] * -1.0f);
    M4 view_from_world = m4_mul(&inv_rot, &inv_tr);
    M4 clip_from_view = m4_perspective_lh_zo(fov_degrees * kDegToRad, aspect, z_near, z_far);
    M4 view_from_clip = m4_inverse(&clip_from_view);
    M4 clip_from_world = m4_mul(&clip_from_view, &view_from_world);
    M4 world_from_clip = m4_inverse(&clip_from_world);

    /* render.cpp:959-971 */
    const float sun_half_size = 40.0f, lightgrid_extent = 40.0f;
    M4 sun_ori = m4_rotate_rh(sun_angle_x_deg * kDegToRad, cosf(sun_angle_y_deg * kDegToRad), sinf(sun_angle_y_deg * kDegToRad), 0.0f);
    M4 sun_inv = m4_inverse(&sun_ori);
    M4 ortho = m4_ortho_rh_zo(-sun_half_size, sun_half_size, -sun_half_size, sun_half_size, -sun_half_size, sun_half_size);
    M4 sun_space_from_world = m4_mul(&ortho, &sun_inv);

    memset(g, 0, sizeof *g);
    memcpy(g->clip_space_from_world, clip_from_world.m, 64);
    memcpy(g->clip_space_from_view, clip_from_view.m, 64);
    memcpy(g->world_space_from_clip, world_from_clip.m, 64);
    memcpy(g->view_space_from_clip, view_from_clip.m, 64);
    memcpy(g->view_space_from_world, view_from_world.m, 64);
    memcpy(g->world_space_from_view, world_from_view.m, 64);
    memcpy(g->sun_space_from_world, sun_space_from_world.m, 64);
    memcpy(g->old_clip_space_from_world, clip_from_world.m, 64);              /* frame 0 (render.cpp:985) */
    /* sun_dir = sun_ori * (0,0,-1,0) (render.cpp:970), as the matrix-vector product forms it: ((c0*0 + c1*0) + c2*(-1)) + c3*0,
     * which turns a -0 component into +0 */
    for (int i = 0; i < 3; ++i)
        g->sun_direction[i] = ((sun_ori.m[i] * 0.0f + sun_ori.m[4 + i] * 0.0f) + sun_ori.m[8 + i] * -1.0f) + sun_ori.m[12 + i] * 0.0f;
    g->sun_direction[3] = 0.0f;
    g->camera_pos[0] = pos[0]; g->camera_pos[1] = pos[1]; g->camera_pos[2] = pos[2];
    g->frame_idx_mod_59 = (float)(frame_idx % 59);
    g->lightgrid_scale = 1.0f / lightgrid_extent;
    g->visualize_lightgrid = 0;
}

/* ---- G-buffer (render.cpp:680-693) ---- */
void PBR_MakeGBuffer(PBR_GBuffer* gb, uint32_t w, uint32_t h, GPU_Format result_format) {
    gb->base_color = GPU_MakeTexture(GPU_Format_RGBA8UN, w, h, 1, GPU_TextureFlag_RenderTarget, NULL);
    gb->normal = GPU_MakeTexture(GPU_Format_RGBA8UN, w, h, 1, GPU_TextureFlag_RenderTarget, NULL);
    gb->orm = GPU_MakeTexture(GPU_Format_RGBA8UN, w, h, 1, GPU_TextureFlag_RenderTarget, NULL);
    gb->emissive = GPU_MakeTexture(GPU_Format_RGBA8UN, w, h, 1, GPU_TextureFlag_RenderTarget, NULL);
    gb->depth = GPU_MakeTexture(GPU_Format_D32F_Or_X8D24UN, w, h, 1, GPU_TextureFlag_RenderTarget, NULL);
    gb->lighting_result = GPU_MakeTexture(result_format, w, h, 1, GPU_TextureFlag_RenderTarget, NULL);
}
void PBR_DestroyGBuffer(PBR_GBuffer* gb) {
    GPU_DestroyTexture(gb->base_color); GPU_DestroyTexture(gb->normal); GPU_DestroyTexture(gb->orm);
    GPU_DestroyTexture(gb->emissive); GPU_DestroyTexture(gb->depth); GPU_DestroyTexture(gb->lighting_result);
    memset(gb, 0, sizeof *gb);
}

/* ---- lighting pass objects ---- */
struct PBR_LightingPass {
    GPU_PipelineLayout* layout;
    GPU_RenderPass* render_pass;
    GPU_GraphicsPipeline* pipeline;
    GPU_DescriptorSet* desc_set;
    GPU_Buffer* globals_buffer;
    GPU_Texture* dummy2d; GPU_Texture* dummy3d; GPU_Texture* dummy_depth;
    GPU_Sampler* sampler_pcf;
};

PBR_LightingPass* PBR_MakeLightingPass(const PBR_GBuffer* gb, const PBR_IBLMaps* maps, uint32_t width, uint32_t height) {
    return PBR_MakeLightingPassEx(gb, maps, width, height, NULL);
}

PBR_LightingPass* PBR_MakeLightingPassEx(const PBR_GBuffer* gb, const PBR_IBLMaps* maps, uint32_t width, uint32_t height, GPU_Texture* sun_depth_map) {
    return PBR_MakeLightingPassLive(gb, maps, width, height, sun_depth_map, NULL, NULL);
}

PBR_LightingPass* PBR_MakeLightingPassLive(const PBR_GBuffer* gb, const PBR_IBLMaps* maps, uint32_t width, uint32_t height,
                                           GPU_Texture* sun_depth_map, GPU_Texture* lightgrid, GPU_Texture* prev_frame_result) {
    PBR_LightingPass* lp = (PBR_LightingPass*)calloc(1, sizeof *lp);
    /* render.cpp:664-675 */
    GPU_SamplerDesc pcf; memset(&pcf, 0, sizeof pcf);
    pcf.min_filter = pcf.mag_filter = pcf.mipmap_mode = GPU_Filter_Linear;
    pcf.address_modes[0] = pcf.address_modes[1] = pcf.address_modes[2] = GPU_AddressMode_Clamp;
    pcf.max_lod = 1000.f; pcf.compare_op = GPU_CompareOp_Less;
    lp->sampler_pcf = GPU_MakeSampler(&pcf);
    lp->globals_buffer = GPU_MakeBuffer((uint32_t)sizeof(PBR_Globals) + 8, GPU_BufferFlag_CPU | GPU_BufferFlag_GPU | GPU_BufferFlag_StorageBuffer, NULL);
    /* stand-ins for the out-of-scope inputs (light grid, previous frame, sun depth map): bound, never read */
    lp->dummy2d = GPU_MakeTexture(GPU_Format_RGBA16F, 1, 1, 1, GPU_TextureFlag_RenderTarget, NULL);
    lp->dummy3d = GPU_MakeTexture(GPU_Format_RGBA16F, 1, 1, 1, GPU_TextureFlag_StorageImage, NULL);
    lp->dummy_depth = GPU_MakeTexture(GPU_Format_D32F_Or_X8D24UN, 1, 1, 1, GPU_TextureFlag_RenderTarget, NULL);

    /* render.cpp:716-723 */
    GPU_TextureView lighting_color_targets[] = {{gb->lighting_result, 0}};
    GPU_RenderPassDesc pass_desc; memset(&pass_desc, 0, sizeof pass_desc);
    pass_desc.width = width; pass_desc.height = height;
    pass_desc.color_targets = lighting_color_targets; pass_desc.color_targets_count = 1;
    lp->render_pass = GPU_MakeRenderPass(&pass_desc);

    /* render.cpp:829-848 */
    GPU_PipelineLayout* lo = lp->layout = GPU_InitPipelineLayout();
    uint32_t globals_b = GPU_BufferBinding(lo, "GLOBALS");
    uint32_t base_b = GPU_TextureBinding(lo, "GBUFFER_BASE_COLOR");
    uint32_t normal_b = GPU_TextureBinding(lo, "GBUFFER_NORMAL");
    uint32_t orm_b = GPU_TextureBinding(lo, "GBUFFER_ORM");
    uint32_t emissive_b = GPU_TextureBinding(lo, "GBUFFER_EMISSIVE");
    uint32_t depth_b = GPU_TextureBinding(lo, "GBUFFER_DEPTH");
    uint32_t irr_b = GPU_TextureBinding(lo, "TEX_IRRADIANCE_MAP");
    uint32_t pre_b = GPU_TextureBinding(lo, "PREFILTERED_ENV_MAP");
    uint32_t lut_b = GPU_TextureBinding(lo, "BRDF_INTEGRATION_MAP");
    uint32_t grid_b = GPU_TextureBinding(lo, "LIGHTGRID");
    uint32_t prev_b = GPU_TextureBinding(lo, "PREV_FRAME_RESULT");
    uint32_t sun_b = GPU_TextureBinding(lo, "SUN_DEPTH_MAP");
    uint32_t s_lc = GPU_SamplerBinding(lo, "SAMPLER_LINEAR_CLAMP");
    uint32_t s_lw = GPU_SamplerBinding(lo, "SAMPLER_LINEAR_WRAP");
    uint32_t s_nc = GPU_SamplerBinding(lo, "SAMPLER_NEAREST_CLAMP");
    uint32_t s_pcf = GPU_SamplerBinding(lo, "SAMPLER_PERCENTAGE_CLOSER");
    GPU_FinalizePipelineLayout(lo);

    /* render.cpp:238-278: lighting pipeline from lighting_pass.glsl (vertex + fragment stage of one file) */
    static const char path[] = "../src/demo_pbr_renderer/shaders/lighting_pass.glsl";
    GPU_GraphicsPipelineDesc desc; memset(&desc, 0, sizeof desc);
    desc.layout = lo; desc.render_pass = lp->render_pass;
    desc.vs.glsl_debug_filepath.data = path; desc.vs.glsl_debug_filepath.length = sizeof path - 1;
    desc.fs.glsl_debug_filepath = desc.vs.glsl_debug_filepath;
    GPU_GLSLErrorArray errors = {0};
    desc.vs.spirv = GPU_SPIRVFromGLSL(NULL, GPU_ShaderStage_Vertex, lo, &desc.vs, &errors);
    desc.fs.spirv = GPU_SPIRVFromGLSL(NULL, GPU_ShaderStage_Fragment, lo, &desc.fs, &errors);
    lp->pipeline = GPU_MakeGraphicsPipeline(&desc);

    /* render.cpp:850-870 */
    GPU_DescriptorSet* s = lp->desc_set = GPU_InitDescriptorSet(NULL, lo);
    GPU_SetBufferBinding(s, globals_b, lp->globals_buffer);
    GPU_SetTextureBinding(s, base_b, gb->base_color);
    GPU_SetTextureBinding(s, normal_b, gb->normal);
    GPU_SetTextureBinding(s, orm_b, gb->orm);
    GPU_SetTextureBinding(s, emissive_b, gb->emissive);
    GPU_SetTextureBinding(s, depth_b, gb->depth);
    GPU_SetTextureBinding(s, irr_b, maps->irradiance_map);
    GPU_SetTextureBinding(s, pre_b, maps->tex_specular_env_map);
    GPU_SetTextureBinding(s, lut_b, maps->brdf_lut);
    GPU_SetTextureBinding(s, grid_b, lightgrid ? lightgrid : lp->dummy3d);                      /* render.cpp:861 r->lightgrid */
    GPU_SetTextureBinding(s, prev_b, prev_frame_result ? prev_frame_result : lp->dummy2d);      /* render.cpp:862 bloom_downscale_rt */
    GPU_SetTextureBinding(s, sun_b, sun_depth_map ? sun_depth_map : lp->dummy_depth);            /* render.cpp:676 sun_depth_rt */
    GPU_SetSamplerBinding(s, s_lc, GPU_SamplerLinearClamp());
    GPU_SetSamplerBinding(s, s_lw, GPU_SamplerLinearWrap());
    GPU_SetSamplerBinding(s, s_nc, GPU_SamplerNearestClamp());
    GPU_SetSamplerBinding(s, s_pcf, lp->sampler_pcf);
    GPU_FinalizeDescriptorSet(s);
    return lp;
}

void PBR_DestroyLightingPass(PBR_LightingPass* lp) {
    if (!lp) return;
    GPU_DestroyDescriptorSet(lp->desc_set);
    GPU_DestroyGraphicsPipeline(lp->pipeline);
    GPU_DestroyPipelineLayout(lp->layout);
    GPU_DestroyRenderPass(lp->render_pass);
    GPU_DestroyBuffer(lp->globals_buffer);
    GPU_DestroyTexture(lp->dummy2d); GPU_DestroyTexture(lp->dummy3d); GPU_DestroyTexture(lp->dummy_depth);
    GPU_DestroySampler(lp->sampler_pcf);
    free(lp);
}

GPU_Buffer* PBR_LightingGlobalsBuffer(PBR_LightingPass* lp) { return lp->globals_buffer; }
GPU_GraphicsPipeline* PBR_LightingPipeline(PBR_LightingPass* lp) { return lp->pipeline; }

void PBR_RecordLightingPass(PBR_LightingPass* lp, GPU_Graph* graph, const PBR_Globals* globals, uint32_t row0, uint32_t row1) {
    if (globals) memcpy(lp->globals_buffer->data, globals, sizeof *globals);           /* render.cpp:991 */
    GPU_OpPrepareRenderPass(graph, lp->render_pass);                                    /* render.cpp:1119-1127 */
    uint32_t draw_params = GPU_OpPrepareDrawParams(graph, lp->pipeline, lp->desc_set);
    GPU_OpBeginRenderPass(graph);
    GPU_OpBindDrawParams(graph, draw_params);
    if (row1 == 0) GPU_OpDraw(graph, 3, 1, 0, 0);                                       /* fullscreen triangle */
    else GPUX_OpDrawRows(graph, row0, row1);
    GPU_OpEndRenderPass(graph);
}
