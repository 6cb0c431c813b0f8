/*
 * pbr_lightgrid.c -- host side of the voxel light-grid sweep, the GPU_* call sequence of
 *   render.cpp:678          the 128^3 RGBA16F storage image
 *   render.cpp:816          its "IMG0" storage-image binding in the main pass layout
 *   render.cpp:151-187      the sweep compute pipeline + descriptor set (only IMG0 is read by the shader;
 *                           the reference fills the layout's other slots with dummies it calls "stupid")
 *   render.cpp:1028         frame-0 clear
 *   render.cpp:1061-1072    per frame: advance sweep_direction, push it, GPU_OpDispatch(1, 16, 16)
 * The voxelize raster pass that writes occupied voxels into the grid (render.cpp:1036-1056) is outside this
 * backend; callers upload grid contents with GPUX_OpCopyBufferToTextureMip.
 */
#include "pbr_host.h"

#include <stdlib.h>
#include <string.h>

struct PBR_Lightgrid {
    GPU_Texture* lightgrid;
    GPU_PipelineLayout* layout;
    uint32_t img0_binding;
    GPU_ComputePipeline* sweep_pipeline;
    GPU_DescriptorSet* sweep_desc_set;
    uint32_t sweep_direction;            /* render.h:204 (zero-initialised, incremented before use) */
};

PBR_Lightgrid* PBR_MakeLightgrid(uint32_t size) {
    PBR_Lightgrid* lg = (PBR_Lightgrid*)calloc(1, sizeof *lg);
    lg->lightgrid = GPU_MakeTexture(GPU_Format_RGBA16F, size, size, size, GPU_TextureFlag_StorageImage, NULL);   /* render.cpp:678 */
    lg->layout = GPU_InitPipelineLayout();
    lg->img0_binding = GPU_StorageImageBinding(lg->layout, "IMG0", lg->lightgrid->format);                       /* render.cpp:816 */
    GPU_FinalizePipelineLayout(lg->layout);

    /* render.cpp:156-162 */
    static const char path[] = "../src/demo_pbr_renderer/shaders/lightgrid_sweep.glsl";
    GPU_Access cs_accesses[] = { GPU_ReadWrite(lg->img0_binding) };
    GPU_ShaderDesc cs_desc; memset(&cs_desc, 0, sizeof cs_desc);
    cs_desc.accesses = cs_accesses; cs_desc.accesses_count = 1;
    cs_desc.glsl_debug_filepath.data = path; cs_desc.glsl_debug_filepath.length = sizeof path - 1;
    GPU_GLSLErrorArray errors = {0};
    cs_desc.spirv = GPU_SPIRVFromGLSL(NULL, GPU_ShaderStage_Compute, lg->layout, &cs_desc, &errors);
    lg->sweep_pipeline = GPU_MakeComputePipeline(lg->layout, &cs_desc);

    /* render.cpp:164-165, 186 */
    lg->sweep_desc_set = GPU_InitDescriptorSet(NULL, lg->layout);
    GPU_SetStorageImageBinding(lg->sweep_desc_set, lg->img0_binding, lg->lightgrid, 0);
    GPU_FinalizeDescriptorSet(lg->sweep_desc_set);
    return lg;
}

void PBR_DestroyLightgrid(PBR_Lightgrid* lg) {
    if (!lg) return;
    GPU_DestroyDescriptorSet(lg->sweep_desc_set);
    GPU_DestroyComputePipeline(lg->sweep_pipeline);                                      /* render.cpp:881 */
    GPU_DestroyPipelineLayout(lg->layout);
    GPU_DestroyTexture(lg->lightgrid);                                                   /* render.cpp:945 */
    free(lg);
}

GPU_Texture* PBR_LightgridTexture(PBR_Lightgrid* lg) { return lg->lightgrid; }
uint32_t PBR_LightgridSweepDirection(const PBR_Lightgrid* lg) { return lg->sweep_direction; }

void PBR_RecordLightgridClear(PBR_Lightgrid* lg, GPU_Graph* graph) {
    GPU_OpClearColorF(graph, lg->lightgrid, GPU_MIP_LEVEL_ALL, 0.f, 0.f, 0.f, 0.f);     /* render.cpp:1028 */
}

void PBR_RecordLightgridSweep(PBR_Lightgrid* lg, GPU_Graph* graph) {
    lg->sweep_direction++;                                                               /* render.cpp:1064-1065 */
    if (lg->sweep_direction == 3) lg->sweep_direction = 0;
    GPU_OpBindComputePipeline(graph, lg->sweep_pipeline);                                /* render.cpp:1067-1069 */
    GPU_OpBindComputeDescriptorSet(graph, lg->sweep_desc_set);
    GPU_OpPushComputeConstants(graph, lg->layout, &lg->sweep_direction, sizeof(lg->sweep_direction));
    /* render.cpp:1071-1072 asserts a 128^3 grid and dispatches (1,16,16) groups of 1x8x8; other cubic sizes scale the counts */
    uint32_t groups = lg->lightgrid->height / 8;
    GPU_OpDispatch(graph, 1, groups, groups);
}

void PBR_RecordLightgridSweepLines(PBR_Lightgrid* lg, GPU_Graph* graph, uint32_t direction, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1) {
    GPU_OpBindComputePipeline(graph, lg->sweep_pipeline);
    GPU_OpBindComputeDescriptorSet(graph, lg->sweep_desc_set);
    GPU_OpPushComputeConstants(graph, lg->layout, &direction, sizeof direction);
    GPUX_OpDispatchLines(graph, y0, y1, z0, z1);
}
