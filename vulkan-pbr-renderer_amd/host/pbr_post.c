/*
 * pbr_post.c -- host side of the post-process tail, the GPU_* call sequences of
 *   render.cpp:690-697      gbuffer_velocity[2] (RG16F), taa_output_rt[2] (RGBA16F ping-pong)
 *   render.cpp:732-739      the two TAA resolve render passes
 *   render.cpp:281-337      TAA pipelines + descriptor sets (set i reads taa_output_rt[1-i], velocity[i] / [1-i])
 *   render.cpp:456-501      final post-process pipeline + descriptor sets
 *   render.cpp:782-785      final pass onto the swapchain  -> here: onto an 8-bit "backbuffer" texture
 *   render.cpp:1131-1137, 1181-1187   the two full-screen draws per frame
 *   render.cpp:741-777, 340-454, 1139-1176   bloom: render targets with mips, 6 + 6 passes, clear + blit + additive upsamples
 * The geometry raster pass that writes depth/velocity is outside this backend: callers upload velocity.  The final pass
 * exists in two flavours: reading bloom_upscale_rt like the reference (PBR_RecordFinalPostProcessBloom, after
 * PBR_RecordBloom) or reading the TAA result directly (its TEX0 slot is whatever texture is bound).
 */
#include "pbr_host.h"

#include <stdlib.h>
#include <string.h>

struct PBR_PostProcess {
    uint32_t width, height;
    GPU_Texture* gbuffer_velocity[2];
    GPU_Texture* taa_output_rt[2];
    GPU_Texture* backbuffer;
    GPU_PipelineLayout* layout;
    uint32_t lighting_result_b, depth_b, velocity_b, velocity_prev_b, prev_frame_b, tex0_b, sampler_b;
    GPU_RenderPass* taa_resolve_render_pass[2];
    GPU_GraphicsPipeline* taa_resolve_pipeline[2];
    GPU_DescriptorSet* taa_resolve_descriptor_set[2];
    GPU_RenderPass* final_post_process_render_pass;
    GPU_GraphicsPipeline* final_post_process_pipeline;
    GPU_DescriptorSet* final_post_process_desc_set[2];
    GPU_Texture* dummy;                  /* fills the slots a pass does not read (the reference's "unused descriptors") */
    /* bloom (render.h:2 BLOOM_PASS_COUNT 6; fewer when the frame is too small to have that many mips) */
    uint32_t bloom_pass_count;
    GPU_Texture* bloom_downscale_rt; GPU_Texture* bloom_upscale_rt;
    struct { GPU_RenderPass* render_pass[2]; GPU_GraphicsPipeline* pipeline[2]; GPU_DescriptorSet* desc_set[2]; } bloom_downsamples[6], bloom_upsamples[6];
    GPU_DescriptorSet* final_post_process_bloom_desc_set[2];
};

static GPU_GraphicsPipeline* make_fullscreen_pipeline_blend(const char* path, size_t path_len, GPU_PipelineLayout* lo, GPU_RenderPass* rp, bool additive);
static void fill_unused(PBR_PostProcess* pp, GPU_DescriptorSet* s) {
    GPU_SetSamplerBinding(s, pp->sampler_b, GPU_SamplerLinearClamp());
    GPU_SetTextureBinding(s, pp->prev_frame_b, pp->dummy);
    GPU_SetTextureBinding(s, pp->depth_b, pp->dummy);
    GPU_SetTextureBinding(s, pp->velocity_b, pp->dummy);
    GPU_SetTextureBinding(s, pp->velocity_prev_b, pp->dummy);
    GPU_SetTextureBinding(s, pp->lighting_result_b, pp->dummy);
}

static GPU_GraphicsPipeline* make_fullscreen_pipeline(const char* path, size_t path_len, GPU_PipelineLayout* lo, GPU_RenderPass* rp) {
    return make_fullscreen_pipeline_blend(path, path_len, lo, rp, false);
}
static GPU_GraphicsPipeline* make_fullscreen_pipeline_blend(const char* path, size_t path_len, GPU_PipelineLayout* lo, GPU_RenderPass* rp, bool additive) {
    GPU_GraphicsPipelineDesc desc; memset(&desc, 0, sizeof desc);
    desc.layout = lo; desc.render_pass = rp;
    desc.enable_blending = additive; desc.blending_mode_additive = additive;                  /* render.cpp:417-418 */
    desc.vs.glsl_debug_filepath.data = path; desc.vs.glsl_debug_filepath.length = path_len;
    desc.fs.glsl_debug_filepath = desc.vs.glsl_debug_filepath;
    GPU_GLSLErrorArray errors = {0};
    desc.vs.spirv = GPU_SPIRVFromGLSL(NULL, GPU_ShaderStage_Vertex, lo, &desc.vs, &errors);     /* render.cpp:32-60 LoadVertexAndFragmentShader */
    desc.fs.spirv = GPU_SPIRVFromGLSL(NULL, GPU_ShaderStage_Fragment, lo, &desc.fs, &errors);
    return GPU_MakeGraphicsPipeline(&desc);
}

PBR_PostProcess* PBR_MakePostProcess(const PBR_GBuffer* gb, uint32_t width, uint32_t height, GPU_Format backbuffer_format) {
    PBR_PostProcess* pp = (PBR_PostProcess*)calloc(1, sizeof *pp);
    pp->width = width; pp->height = height;
    for (int i = 0; i < 2; ++i) {
        pp->gbuffer_velocity[i] = GPU_MakeTexture(GPU_Format_RG16F, width, height, 1, GPU_TextureFlag_RenderTarget, NULL);    /* render.cpp:690-691 */
        pp->taa_output_rt[i] = GPU_MakeTexture(GPU_Format_RGBA16F, width, height, 1, GPU_TextureFlag_RenderTarget, NULL);     /* render.cpp:696-697 */
    }
    pp->backbuffer = GPU_MakeTexture(backbuffer_format, width, height, 1, GPU_TextureFlag_RenderTarget, NULL);
    pp->dummy = GPU_MakeTexture(GPU_Format_RGBA16F, 1, 1, 1, GPU_TextureFlag_RenderTarget, NULL);

    /* the slots of the main pass layout these two shaders name (render.cpp:800-826) */
    GPU_PipelineLayout* lo = pp->layout = GPU_InitPipelineLayout();
    pp->tex0_b = GPU_TextureBinding(lo, "TEX0");
    pp->sampler_b = GPU_SamplerBinding(lo, "SAMPLER_LINEAR_CLAMP");
    pp->prev_frame_b = GPU_TextureBinding(lo, "PREV_FRAME_RESULT");
    pp->depth_b = GPU_TextureBinding(lo, "GBUFFER_DEPTH");
    pp->velocity_b = GPU_TextureBinding(lo, "GBUFFER_VELOCITY");
    pp->velocity_prev_b = GPU_TextureBinding(lo, "GBUFFER_VELOCITY_PREV");
    pp->lighting_result_b = GPU_TextureBinding(lo, "LIGHTING_RESULT");
    GPU_FinalizePipelineLayout(lo);

    static const char taa_path[] = "../src/demo_pbr_renderer/shaders/taa_resolve.glsl";
    static const char final_path[] = "../src/demo_pbr_renderer/shaders/final_post_process.glsl";
    for (int i = 0; i < 2; ++i) {
        GPU_TextureView taa_resolve_color_targets[] = {{pp->taa_output_rt[i], 0}};                     /* render.cpp:732-739 */
        GPU_RenderPassDesc resolve_pass_desc; memset(&resolve_pass_desc, 0, sizeof resolve_pass_desc);
        resolve_pass_desc.width = width; resolve_pass_desc.height = height;
        resolve_pass_desc.color_targets = taa_resolve_color_targets; resolve_pass_desc.color_targets_count = 1;
        pp->taa_resolve_render_pass[i] = GPU_MakeRenderPass(&resolve_pass_desc);
        pp->taa_resolve_pipeline[i] = make_fullscreen_pipeline(taa_path, sizeof taa_path - 1, lo, pp->taa_resolve_render_pass[i]);   /* render.cpp:298-310 */

        GPU_DescriptorSet* s = pp->taa_resolve_descriptor_set[i] = GPU_InitDescriptorSet(NULL, lo);   /* render.cpp:312-336 */
        GPU_SetSamplerBinding(s, pp->sampler_b, GPU_SamplerLinearClamp());
        GPU_SetTextureBinding(s, pp->prev_frame_b, pp->taa_output_rt[1 - i]);
        GPU_SetTextureBinding(s, pp->depth_b, gb->depth);
        GPU_SetTextureBinding(s, pp->velocity_b, pp->gbuffer_velocity[i]);
        GPU_SetTextureBinding(s, pp->velocity_prev_b, pp->gbuffer_velocity[1 - i]);
        GPU_SetTextureBinding(s, pp->lighting_result_b, gb->lighting_result);
        GPU_SetTextureBinding(s, pp->tex0_b, pp->dummy);
        GPU_FinalizeDescriptorSet(s);
    }

    GPU_TextureView final_targets[] = {{pp->backbuffer, 0}};                                           /* render.cpp:782-785 (swapchain there) */
    GPU_RenderPassDesc final_pp_pass_desc; memset(&final_pp_pass_desc, 0, sizeof final_pp_pass_desc);
    final_pp_pass_desc.color_targets = final_targets; final_pp_pass_desc.color_targets_count = 1;
    pp->final_post_process_render_pass = GPU_MakeRenderPass(&final_pp_pass_desc);
    pp->final_post_process_pipeline = make_fullscreen_pipeline(final_path, sizeof final_path - 1, lo, pp->final_post_process_render_pass);   /* render.cpp:462-473 */
    for (int i = 0; i < 2; ++i) {
        GPU_DescriptorSet* s = pp->final_post_process_desc_set[i] = GPU_InitDescriptorSet(NULL, lo);  /* render.cpp:475-501 */
        GPU_SetTextureBinding(s, pp->tex0_b, pp->taa_output_rt[i]);                                   /* reference: bloom_upscale_rt */
        GPU_SetSamplerBinding(s, pp->sampler_b, GPU_SamplerLinearClamp());
        GPU_SetTextureBinding(s, pp->prev_frame_b, pp->dummy);
        GPU_SetTextureBinding(s, pp->depth_b, pp->dummy);
        GPU_SetTextureBinding(s, pp->velocity_b, pp->dummy);
        GPU_SetTextureBinding(s, pp->velocity_prev_b, pp->dummy);
        GPU_SetTextureBinding(s, pp->lighting_result_b, pp->dummy);
        GPU_FinalizeDescriptorSet(s);
    }
    /* ---- bloom: render.cpp:741-777 (targets, passes), :340-454 (pipelines, descriptor sets) ---- */
    static const char down_path[] = "../src/demo_pbr_renderer/shaders/bloom_downsample.glsl";
    static const char up_path[] = "../src/demo_pbr_renderer/shaders/bloom_upsample.glsl";
    GPU_TextureFlags bloom_flags = GPU_TextureFlag_RenderTarget | GPU_TextureFlag_HasMipmaps | GPU_TextureFlag_PerMipBinding;
    pp->bloom_downscale_rt = GPU_MakeTexture(GPU_Format_RGBA16F, width / 2 ? width / 2 : 1, height / 2 ? height / 2 : 1, 1, bloom_flags, NULL);
    pp->bloom_upscale_rt = GPU_MakeTexture(GPU_Format_RGBA16F, width, height, 1, bloom_flags, NULL);
    pp->bloom_pass_count = pp->bloom_downscale_rt->mip_level_count < 6 ? pp->bloom_downscale_rt->mip_level_count : 6;
    const uint32_t n = pp->bloom_pass_count;
    for (int i = 0; i < 2; ++i) {
        uint32_t w = width, h = height;
        for (uint32_t step = 0; step < n; ++step) {                                                   /* render.cpp:751-763 */
            GPU_TextureView targets[] = {{pp->bloom_downscale_rt, step}};
            w /= 2; h /= 2;
            GPU_RenderPassDesc pass_desc; memset(&pass_desc, 0, sizeof pass_desc);
            pass_desc.width = w ? w : 1; pass_desc.height = h ? h : 1;
            pass_desc.color_targets = targets; pass_desc.color_targets_count = 1;
            pp->bloom_downsamples[step].render_pass[i] = GPU_MakeRenderPass(&pass_desc);
            pp->bloom_downsamples[step].pipeline[i] = make_fullscreen_pipeline(down_path, sizeof down_path - 1, lo, pp->bloom_downsamples[step].render_pass[i]);
            GPU_DescriptorSet* s = pp->bloom_downsamples[step].desc_set[i] = GPU_InitDescriptorSet(NULL, lo);   /* render.cpp:364-393 */
            if (step == 0) GPU_SetTextureBinding(s, pp->tex0_b, pp->taa_output_rt[i]);
            else GPU_SetTextureMipBinding(s, pp->tex0_b, pp->bloom_downscale_rt, step - 1);
            fill_unused(pp, s);
            GPU_FinalizeDescriptorSet(s);
        }
        w = width; h = height;
        for (int step = (int)n - 1; step >= 0; --step) {                                              /* render.cpp:765-780 */
            uint32_t dst_level = n - 1 - (uint32_t)step;
            GPU_TextureView targets[] = {{pp->bloom_upscale_rt, dst_level}};
            GPU_RenderPassDesc pass_desc; memset(&pass_desc, 0, sizeof pass_desc);
            pass_desc.width = w ? w : 1; pass_desc.height = h ? h : 1;
            pass_desc.color_targets = targets; pass_desc.color_targets_count = 1;
            pp->bloom_upsamples[step].render_pass[i] = GPU_MakeRenderPass(&pass_desc);
            w /= 2; h /= 2;
            pp->bloom_upsamples[step].pipeline[i] = make_fullscreen_pipeline_blend(up_path, sizeof up_path - 1, lo, pp->bloom_upsamples[step].render_pass[i], true);
            GPU_DescriptorSet* s = pp->bloom_upsamples[step].desc_set[i] = GPU_InitDescriptorSet(NULL, lo);     /* render.cpp:422-451 */
            if (step == 0) GPU_SetTextureMipBinding(s, pp->tex0_b, pp->bloom_downscale_rt, n - 1);
            else GPU_SetTextureMipBinding(s, pp->tex0_b, pp->bloom_upscale_rt, n - (uint32_t)step);
            fill_unused(pp, s);
            GPU_FinalizeDescriptorSet(s);
        }
        GPU_DescriptorSet* s = pp->final_post_process_bloom_desc_set[i] = GPU_InitDescriptorSet(NULL, lo);      /* render.cpp:475-501 */
        GPU_SetTextureBinding(s, pp->tex0_b, pp->bloom_upscale_rt);
        fill_unused(pp, s);
        GPU_FinalizeDescriptorSet(s);
    }
    return pp;
}

void PBR_DestroyPostProcess(PBR_PostProcess* pp) {
    if (!pp) return;
    for (uint32_t step = 0; step < pp->bloom_pass_count; ++step)                                      /* render.cpp:890-902, 921-926 */
        for (int i = 0; i < 2; ++i) {
            GPU_DestroyGraphicsPipeline(pp->bloom_downsamples[step].pipeline[i]); GPU_DestroyDescriptorSet(pp->bloom_downsamples[step].desc_set[i]);
            GPU_DestroyGraphicsPipeline(pp->bloom_upsamples[step].pipeline[i]); GPU_DestroyDescriptorSet(pp->bloom_upsamples[step].desc_set[i]);
            GPU_DestroyRenderPass(pp->bloom_downsamples[step].render_pass[i]); GPU_DestroyRenderPass(pp->bloom_upsamples[step].render_pass[i]);
        }
    for (int i = 0; i < 2; ++i) GPU_DestroyDescriptorSet(pp->final_post_process_bloom_desc_set[i]);
    GPU_DestroyTexture(pp->bloom_downscale_rt); GPU_DestroyTexture(pp->bloom_upscale_rt);
    for (int i = 0; i < 2; ++i) {                                                        /* render.cpp:885-888, 904-906 */
        GPU_DestroyGraphicsPipeline(pp->taa_resolve_pipeline[i]);
        GPU_DestroyDescriptorSet(pp->taa_resolve_descriptor_set[i]);
        GPU_DestroyDescriptorSet(pp->final_post_process_desc_set[i]);
        GPU_DestroyRenderPass(pp->taa_resolve_render_pass[i]);
        GPU_DestroyTexture(pp->gbuffer_velocity[i]);
        GPU_DestroyTexture(pp->taa_output_rt[i]);
    }
    GPU_DestroyGraphicsPipeline(pp->final_post_process_pipeline);
    GPU_DestroyRenderPass(pp->final_post_process_render_pass);
    GPU_DestroyPipelineLayout(pp->layout);
    GPU_DestroyTexture(pp->backbuffer);
    GPU_DestroyTexture(pp->dummy);
    free(pp);
}

GPU_Texture* PBR_PostVelocity(PBR_PostProcess* pp, uint32_t frame_idx_mod2) { return pp->gbuffer_velocity[frame_idx_mod2 & 1]; }
GPU_Texture* PBR_PostTaaOutput(PBR_PostProcess* pp, uint32_t frame_idx_mod2) { return pp->taa_output_rt[frame_idx_mod2 & 1]; }
GPU_Texture* PBR_PostBackbuffer(PBR_PostProcess* pp) { return pp->backbuffer; }

void PBR_RecordTaaResolve(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx) {
    uint32_t frame_idx_mod2 = frame_idx % 2;
    GPU_OpPrepareRenderPass(graph, pp->taa_resolve_render_pass[frame_idx_mod2]);          /* render.cpp:1131-1137 */
    uint32_t taa_resolve_pass_draw_params = GPU_OpPrepareDrawParams(graph, pp->taa_resolve_pipeline[frame_idx_mod2], pp->taa_resolve_descriptor_set[frame_idx_mod2]);
    GPU_OpBeginRenderPass(graph);
    GPU_OpBindDrawParams(graph, taa_resolve_pass_draw_params);
    GPU_OpDraw(graph, 3, 1, 0, 0);
    GPU_OpEndRenderPass(graph);
}

void PBR_RecordTaaResolveRows(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx, uint32_t row0, uint32_t row1) {
    uint32_t frame_idx_mod2 = frame_idx % 2;
    GPU_OpPrepareRenderPass(graph, pp->taa_resolve_render_pass[frame_idx_mod2]);
    uint32_t draw_params = GPU_OpPrepareDrawParams(graph, pp->taa_resolve_pipeline[frame_idx_mod2], pp->taa_resolve_descriptor_set[frame_idx_mod2]);
    GPU_OpBeginRenderPass(graph);
    GPU_OpBindDrawParams(graph, draw_params);
    GPUX_OpDrawRows(graph, row0, row1);                                                   /* screen-band sharding */
    GPU_OpEndRenderPass(graph);
}

void PBR_RecordFinalPostProcess(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx) {
    uint32_t frame_idx_mod2 = frame_idx % 2;
    GPU_OpPrepareRenderPass(graph, pp->final_post_process_render_pass);                   /* render.cpp:1181-1187 */
    uint32_t final_pp_draw_params = GPU_OpPrepareDrawParams(graph, pp->final_post_process_pipeline, pp->final_post_process_desc_set[frame_idx_mod2]);
    GPU_OpBeginRenderPass(graph);
    GPU_OpBindDrawParams(graph, final_pp_draw_params);
    GPU_OpDraw(graph, 3, 1, 0, 0);
    GPU_OpEndRenderPass(graph);
}

GPU_Texture* PBR_PostBloomDownscale(PBR_PostProcess* pp) { return pp->bloom_downscale_rt; }
GPU_Texture* PBR_PostBloomUpscale(PBR_PostProcess* pp) { return pp->bloom_upscale_rt; }
uint32_t PBR_PostBloomPassCount(const PBR_PostProcess* pp) { return pp->bloom_pass_count; }

void PBR_RecordBloom(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx) {
    uint32_t frame_idx_mod2 = frame_idx % 2;
    const uint32_t n = pp->bloom_pass_count;
    for (uint32_t step = 0; step < n; step++) {                                           /* render.cpp:1141-1152 */
        GPU_OpPrepareRenderPass(graph, pp->bloom_downsamples[step].render_pass[frame_idx_mod2]);
        uint32_t draw_params = GPU_OpPrepareDrawParams(graph, pp->bloom_downsamples[step].pipeline[frame_idx_mod2], pp->bloom_downsamples[step].desc_set[frame_idx_mod2]);
        GPU_OpBeginRenderPass(graph);
        uint32_t dst_mip_level = step + 1;
        GPU_OpPushGraphicsConstants(graph, pp->layout, &dst_mip_level, sizeof(dst_mip_level));
        GPU_OpBindDrawParams(graph, draw_params);
        GPU_OpDraw(graph, 3, 1, 0, 0);
        GPU_OpEndRenderPass(graph);
    }
    GPU_OpClearColorF(graph, pp->bloom_upscale_rt, GPU_MIP_LEVEL_ALL, 0.f, 0.f, 0.f, 0.f); /* render.cpp:1156 */
    GPU_OpBlitInfo blit; memset(&blit, 0, sizeof blit);                                    /* render.cpp:1158-1163 */
    blit.src_area[1].x = (int)pp->width; blit.src_area[1].y = (int)pp->height; blit.src_area[1].z = 1;
    blit.dst_area[1] = blit.src_area[1];
    blit.src_texture = pp->taa_output_rt[frame_idx_mod2];
    blit.dst_texture = pp->bloom_upscale_rt;
    GPU_OpBlit(graph, &blit);
    for (uint32_t step = 0; step < n; step++) {                                           /* render.cpp:1165-1176 */
        GPU_OpPrepareRenderPass(graph, pp->bloom_upsamples[step].render_pass[frame_idx_mod2]);
        uint32_t draw_params = GPU_OpPrepareDrawParams(graph, pp->bloom_upsamples[step].pipeline[frame_idx_mod2], pp->bloom_upsamples[step].desc_set[frame_idx_mod2]);
        GPU_OpBeginRenderPass(graph);
        uint32_t dst_mip_level = n - step - 1;
        GPU_OpPushGraphicsConstants(graph, pp->layout, &dst_mip_level, sizeof(dst_mip_level));
        GPU_OpBindDrawParams(graph, draw_params);
        GPU_OpDraw(graph, 3, 1, 0, 0);
        GPU_OpEndRenderPass(graph);
    }
}

void PBR_RecordFinalPostProcessBloom(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx) {
    uint32_t frame_idx_mod2 = frame_idx % 2;
    GPU_OpPrepareRenderPass(graph, pp->final_post_process_render_pass);                   /* render.cpp:1181-1187, TEX0 = bloom_upscale_rt */
    uint32_t final_pp_draw_params = GPU_OpPrepareDrawParams(graph, pp->final_post_process_pipeline, pp->final_post_process_bloom_desc_set[frame_idx_mod2]);
    GPU_OpBeginRenderPass(graph);
    GPU_OpBindDrawParams(graph, final_pp_draw_params);
    GPU_OpDraw(graph, 3, 1, 0, 0);
    GPU_OpEndRenderPass(graph);
}
