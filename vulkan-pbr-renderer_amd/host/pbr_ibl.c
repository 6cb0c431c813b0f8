/*
 * pbr_ibl.c -- host driver of the IBL precompute (C11), the GPU_* call sequences of the reference's
 * HotreloadShaders (src/demo_pbr_renderer/render.cpp:505-619) and InitRenderer (:794-796), plus the
 * cost-based work partitioner used to shard (mip, face, row-tile) units over the GPUs of a node.
 */
#include "pbr_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SHADER_DIR "../src/demo_pbr_renderer/shaders/"       /* render.h:4-16 ShaderAssetPaths */

static GPU_String str_of(const char* s) { GPU_String r = {s, strlen(s)}; return r; }

/* render.cpp:9-30 MakeComputePipelineFromShader, minus the file read: the HIP backend identifies the
 * built-in kernel from glsl_debug_filepath, so no GLSL text is needed. */
static GPU_ComputePipeline* make_compute_pipeline(const char* shader_path, GPU_PipelineLayout* layout, GPU_ShaderDesc* cs_desc) {
    cs_desc->glsl_debug_filepath = str_of(shader_path);
    GPU_GLSLErrorArray errors = {0};
    cs_desc->spirv = GPU_SPIRVFromGLSL(NULL, GPU_ShaderStage_Compute, layout, cs_desc, &errors);
    if (cs_desc->spirv.length == 0) {
        GPU_String msg = GPU_JoinGLSLErrorString(NULL, errors);
        fprintf(stderr, "GPU-ERROR: Error in \"%s\": %.*s\n", shader_path, (int)msg.length, msg.data);
        return NULL;
    }
    return GPU_MakeComputePipeline(layout, cs_desc);
}

void PBR_MakeIBLMaps(PBR_IBLMaps* m, uint32_t irradiance_size, uint32_t lut_size, uint32_t specular_size) {
    /* render.cpp:794-796 */
    m->irradiance_map = GPU_MakeTexture(GPU_Format_RGBA32F, irradiance_size, irradiance_size, 1, GPU_TextureFlag_Cubemap | GPU_TextureFlag_StorageImage, NULL);
    m->brdf_lut = GPU_MakeTexture(GPU_Format_RG16F, lut_size, lut_size, 1, GPU_TextureFlag_StorageImage, NULL);
    m->tex_specular_env_map = GPU_MakeTexture(GPU_Format_RGBA32F, specular_size, specular_size, 1,
                                              GPU_TextureFlag_Cubemap | GPU_TextureFlag_HasMipmaps | GPU_TextureFlag_StorageImage, NULL);
}
void PBR_DestroyIBLMaps(PBR_IBLMaps* m) {
    GPU_DestroyTexture(m->irradiance_map); GPU_DestroyTexture(m->brdf_lut); GPU_DestroyTexture(m->tex_specular_env_map);
    memset(m, 0, sizeof *m);
}

/* ---- pipelines + layouts shared by the three generators ---- */
struct PBR_IBLPipelines {
    GPU_PipelineLayout* cube_layout;     /* sampler, env cube, output: render.cpp:506-510 == :543-547 */
    uint32_t sampler_binding, tex_env_cube_binding, output_binding;
    GPU_PipelineLayout* lut_layout;      /* render.cpp:592-594 */
    uint32_t lut_output_binding;
    GPU_ComputePipeline* irradiance; GPU_ComputePipeline* prefilter; GPU_ComputePipeline* lut;
};

PBR_IBLPipelines* PBR_MakeIBLPipelines(void) {
    PBR_IBLPipelines* p = (PBR_IBLPipelines*)calloc(1, sizeof *p);
    p->cube_layout = GPU_InitPipelineLayout();
    p->sampler_binding = GPU_SamplerBinding(p->cube_layout, "SAMPLER_LINEAR_CLAMP");
    p->tex_env_cube_binding = GPU_TextureBinding(p->cube_layout, "TEX_ENV_CUBE");
    p->output_binding = GPU_StorageImageBinding(p->cube_layout, "OUTPUT", GPU_Format_RGBA32F);
    GPU_FinalizePipelineLayout(p->cube_layout);
    GPU_Access cube_accesses[] = { GPU_Read(p->sampler_binding), GPU_Read(p->tex_env_cube_binding), GPU_Write(p->output_binding) };
    GPU_ShaderDesc d1 = {0}; d1.accesses = cube_accesses; d1.accesses_count = 3;
    p->irradiance = make_compute_pipeline(SHADER_DIR "gen_irradiance_map.glsl", p->cube_layout, &d1);
    GPU_ShaderDesc d2 = {0}; d2.accesses = cube_accesses; d2.accesses_count = 3;
    p->prefilter = make_compute_pipeline(SHADER_DIR "gen_prefiltered_env_map.glsl", p->cube_layout, &d2);

    p->lut_layout = GPU_InitPipelineLayout();
    p->lut_output_binding = GPU_StorageImageBinding(p->lut_layout, "OUTPUT", GPU_Format_RG16F);
    GPU_FinalizePipelineLayout(p->lut_layout);
    GPU_Access lut_accesses[] = { GPU_Write(p->lut_output_binding) };
    GPU_ShaderDesc d3 = {0}; d3.accesses = lut_accesses; d3.accesses_count = 1;
    p->lut = make_compute_pipeline(SHADER_DIR "gen_brdf_integration_map.glsl", p->lut_layout, &d3);
    return p;
}
void PBR_DestroyIBLPipelines(PBR_IBLPipelines* p) {
    if (!p) return;
    GPU_DestroyComputePipeline(p->irradiance); GPU_DestroyComputePipeline(p->prefilter); GPU_DestroyComputePipeline(p->lut);
    GPU_DestroyPipelineLayout(p->cube_layout); GPU_DestroyPipelineLayout(p->lut_layout);
    free(p);
}

static uint32_t group_count(uint32_t size) { return size / 8 > 0 ? size / 8 : 1; }   /* reference: size / 8 (sizes >= 16) */

/* render.cpp:505-540 */
void PBR_GenIrradianceMap(GPU_Texture* tex_env_cube, GPU_Texture* irradiance_map) {
    PBR_IBLPipelines* p = PBR_MakeIBLPipelines();
    GPU_DescriptorSet* desc_set = GPU_InitDescriptorSet(NULL, p->cube_layout);
    GPU_SetSamplerBinding(desc_set, p->sampler_binding, GPU_SamplerLinearClamp());
    GPU_SetTextureBinding(desc_set, p->tex_env_cube_binding, tex_env_cube);
    GPU_SetStorageImageBinding(desc_set, p->output_binding, irradiance_map, 0);
    GPU_FinalizeDescriptorSet(desc_set);

    GPU_Graph* graph = GPU_MakeGraph();
    GPU_OpBindComputePipeline(graph, p->irradiance);
    GPU_OpBindComputeDescriptorSet(graph, desc_set);
    GPU_OpDispatch(graph, group_count(irradiance_map->width), group_count(irradiance_map->height), 1);
    GPU_GraphSubmit(graph);
    GPU_GraphWait(graph);

    GPU_DestroyGraph(graph);
    GPU_DestroyDescriptorSet(desc_set);
    PBR_DestroyIBLPipelines(p);
}

/* render.cpp:542-589 */
void PBR_GenPrefilteredEnvMap(GPU_Texture* tex_env_cube, GPU_Texture* spec, uint32_t min_size) {
    PBR_IBLPipelines* p = PBR_MakeIBLPipelines();
    GPU_Graph* graph = GPU_MakeGraph();
    GPU_OpBindComputePipeline(graph, p->prefilter);
    GPU_DescriptorArena* descriptor_arena = GPU_MakeDescriptorArena();

    uint32_t size = spec->width;
    for (uint32_t i = 0; i < spec->mip_level_count; i++) {
        if (size < min_size) break;                        /* reference: `if (size < 16) break;` */

        GPU_DescriptorSet* desc_set = GPU_InitDescriptorSet(descriptor_arena, p->cube_layout);
        GPU_SetSamplerBinding(desc_set, p->sampler_binding, GPU_SamplerLinearClamp());
        GPU_SetTextureBinding(desc_set, p->tex_env_cube_binding, tex_env_cube);
        GPU_SetStorageImageBinding(desc_set, p->output_binding, spec, i);
        GPU_FinalizeDescriptorSet(desc_set);

        GPU_OpBindComputePipeline(graph, p->prefilter);
        GPU_OpBindComputeDescriptorSet(graph, desc_set);
        GPU_OpPushComputeConstants(graph, p->cube_layout, &i, sizeof(i));
        GPU_OpDispatch(graph, group_count(size), group_count(size), 1);
        size /= 2;
    }
    GPU_GraphSubmit(graph);
    GPU_GraphWait(graph);

    GPU_DestroyDescriptorArena(descriptor_arena);
    GPU_DestroyGraph(graph);
    PBR_DestroyIBLPipelines(p);
}

/* render.cpp:591-619 */
void PBR_GenBRDFIntegrationMap(GPU_Texture* brdf_lut) {
    PBR_IBLPipelines* p = PBR_MakeIBLPipelines();
    GPU_Graph* graph = GPU_MakeGraph();
    GPU_DescriptorSet* desc_set = GPU_InitDescriptorSet(NULL, p->lut_layout);
    GPU_SetStorageImageBinding(desc_set, p->lut_output_binding, brdf_lut, 0);
    GPU_FinalizeDescriptorSet(desc_set);

    GPU_OpBindComputePipeline(graph, p->lut);
    GPU_OpBindComputeDescriptorSet(graph, desc_set);
    GPU_OpDispatch(graph, group_count(brdf_lut->width), group_count(brdf_lut->height), 1);   /* reference: 256/8 */
    GPU_GraphSubmit(graph);
    GPU_GraphWait(graph);

    GPU_DestroyDescriptorSet(desc_set);
    GPU_DestroyGraph(graph);
    PBR_DestroyIBLPipelines(p);
}

/* ---- sharded recording ---- */
void PBR_RecordUnits(PBR_IBLPipelines* p, GPU_Graph* graph, GPU_DescriptorArena* arena, GPU_Texture* env,
                     const PBR_IBLMaps* maps, const PBR_WorkUnit* units, uint32_t n) {
    for (uint32_t k = 0; k < n; ++k) {
        const PBR_WorkUnit* u = &units[k];
        if (u->kind == PBR_Unit_BrdfLut) {
            GPU_DescriptorSet* s = GPU_InitDescriptorSet(arena, p->lut_layout);
            GPU_SetStorageImageBinding(s, p->lut_output_binding, maps->brdf_lut, 0);
            GPU_FinalizeDescriptorSet(s);
            GPU_OpBindComputePipeline(graph, p->lut);
            GPU_OpBindComputeDescriptorSet(graph, s);
            GPUX_OpDispatchRows(graph, 0, 1, u->row0, u->row1);
            continue;
        }
        int irr = u->kind == PBR_Unit_Irradiance;
        GPU_DescriptorSet* s = GPU_InitDescriptorSet(arena, p->cube_layout);
        GPU_SetSamplerBinding(s, p->sampler_binding, GPU_SamplerLinearClamp());
        GPU_SetTextureBinding(s, p->tex_env_cube_binding, env);
        GPU_SetStorageImageBinding(s, p->output_binding, irr ? maps->irradiance_map : maps->tex_specular_env_map, irr ? 0 : u->mip);
        GPU_FinalizeDescriptorSet(s);
        GPU_OpBindComputePipeline(graph, irr ? p->irradiance : p->prefilter);
        GPU_OpBindComputeDescriptorSet(graph, s);
        uint32_t mip = u->mip;
        if (!irr) GPU_OpPushComputeConstants(graph, p->cube_layout, &mip, sizeof mip);
        else GPU_OpPushComputeConstants(graph, p->cube_layout, &mip, 0);
        GPUX_OpDispatchRows(graph, u->face0, u->face1, u->row0, u->row1);
    }
}

/* ---- partitioner: cost(unit) = texels * samples; units = (mip, face, row tile) ---- */
static double samples_per_texel(uint32_t mip) {
    /* non-zero Beckmann weights out of 8192 (mip 1 keeps 1389 samples, others all) */
    if (mip == 0) return 1.0;
    if (mip == 1) return 1389.0;
    return 8192.0;
}

/* Measured time per sample-evaluation relative to mip 2 (one MI355X, C4): the MC kernel is a little faster where
 * most weights are large (mip 1) and slower where levels are small; the irradiance pass reads a tiny level.  The copy mip moves
 * 16 B per texel at ~3.4 TB/s, i.e. as long as ~3.8 sample-evaluations per texel.  Used for balancing only; unit.cost stays
 * the plain count. */
static double time_weight(const PBR_WorkUnit* u) {
    if (u->kind == PBR_Unit_Irradiance) return 1.24;
    switch (u->mip) {
    case 0: return 3.8;
    case 1: return 0.965;
    case 2: return 1.0;
    case 3: return 1.04;
    case 4: return 1.13;
    case 5: return 1.07;
    default: return 1.1;
    }
}

static int cmp_cost_desc(const void* a, const void* b) {
    double ca = ((const PBR_WorkUnit*)a)->cost, cb = ((const PBR_WorkUnit*)b)->cost;
    if (ca != cb) return ca < cb ? 1 : -1;
    const PBR_WorkUnit* x = (const PBR_WorkUnit*)a; const PBR_WorkUnit* y = (const PBR_WorkUnit*)b;   /* deterministic tie-break */
    if (x->kind != y->kind) return x->kind < y->kind ? -1 : 1;
    if (x->mip != y->mip) return x->mip < y->mip ? -1 : 1;
    if (x->face0 != y->face0) return x->face0 < y->face0 ? -1 : 1;
    return x->row0 < y->row0 ? -1 : (x->row0 > y->row0);
}

uint32_t PBR_PartitionIBL(uint32_t specular_size, uint32_t min_size, uint32_t irradiance_size, uint32_t env_size,
                          int world, int rank, PBR_WorkUnit* out, uint32_t capacity) {
    (void)env_size;
    if (world < 1) world = 1;
    if (min_size < 1) min_size = 1;
    /* enumerate: each (mip, face) is cut into row tiles so that no unit exceeds ~1/(8*world) of the total cost */
    double total = 0.0;
    uint32_t mips = 0;
    for (uint32_t s = specular_size, m = 0; s >= min_size && s >= 1; s /= 2, ++m) {
        total += 6.0 * s * s * samples_per_texel(m);
        mips = m + 1;
        if (s == 1) break;
    }
    if (irradiance_size) total += 6.0 * irradiance_size * irradiance_size * 1024.0;
    double target = total / (8.0 * world);
    if (world == 1 && rank <= 0) {
        /* single GPU: one dispatch per output level (the reference's own granularity, render.cpp:564-580) */
        uint32_t n1 = 0;
        for (uint32_t m = 0; m <= mips; ++m) {
            int irr = m == mips;
            if (irr && !irradiance_size) break;
            uint32_t size = irr ? irradiance_size : (specular_size >> m ? specular_size >> m : 1);
            if (out && n1 < capacity) {
                PBR_WorkUnit* u = &out[n1];
                u->kind = irr ? PBR_Unit_Irradiance : PBR_Unit_Prefilter;
                u->mip = irr ? 0 : m; u->face0 = 0; u->face1 = 6; u->row0 = 0; u->row1 = size;
                u->cost = 6.0 * size * size * (irr ? 1024.0 : samples_per_texel(m));
            }
            ++n1;
        }
        return n1;
    }
    uint32_t cap_all = 0, n = 0;
    PBR_WorkUnit* all = NULL;
    for (int pass = 0; pass < 2; ++pass) {
        n = 0;
        for (uint32_t m = 0; m <= mips; ++m) {
            int irr = m == mips;
            if (irr && !irradiance_size) break;
            uint32_t size = irr ? irradiance_size : (specular_size >> m ? specular_size >> m : 1);
            double per_row = (double)size * (irr ? 1024.0 : samples_per_texel(m));
            uint32_t rows_per_tile = (uint32_t)(target / per_row);
            if (rows_per_tile < 1) rows_per_tile = 1;
            if (rows_per_tile > size) rows_per_tile = size;
            /* keep tiles a multiple of 16 rows where possible (kernel tile height) */
            if (rows_per_tile >= 16) rows_per_tile &= ~15u;
            for (uint32_t f = 0; f < 6; ++f)
                for (uint32_t r = 0; r < size; r += rows_per_tile) {
                    if (pass == 1) {
                        PBR_WorkUnit* u = &all[n];
                        u->kind = irr ? PBR_Unit_Irradiance : PBR_Unit_Prefilter;
                        u->mip = irr ? 0 : m; u->face0 = f; u->face1 = f + 1;
                        u->row0 = r; u->row1 = r + rows_per_tile < size ? r + rows_per_tile : size;
                        u->cost = per_row * (u->row1 - u->row0);
                    }
                    ++n;
                }
        }
        if (pass == 0) { cap_all = n; all = (PBR_WorkUnit*)malloc(sizeof(PBR_WorkUnit) * (cap_all ? cap_all : 1)); }
    }
    qsort(all, n, sizeof *all, cmp_cost_desc);
    /* Communication-aware greedy assignment.  Outputs are gathered on rank 0, so the copy mip (mip 0: 75 % of the
     * output bytes, 0.4 % of the compute) stays on rank 0 and never crosses xGMI; its cost is charged to rank 0's
     * load first.  Everything else: longest-first onto the least loaded rank (ties -> lowest rank). */
    double* load = (double*)calloc((size_t)world, sizeof(double));
    uint32_t written = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t k = 0; k < n; ++k) {
            int pinned = all[k].kind == PBR_Unit_Prefilter && all[k].mip == 0;
            if ((pass == 0) != pinned) continue;
            int best = 0;
            if (!pinned) for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
            load[best] += all[k].cost * time_weight(&all[k]);
            if (rank < 0 || best == rank) {
                if (written < capacity && out) out[written] = all[k];
                ++written;
            }
        }
    }
    free(load); free(all);
    return written;
}
