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

/* Measured time per sample-evaluation relative to mip 2 (one MI355X, C4, round 3 kernels; the values for mips >= 4 and the copy level are
 * fitted to the shares of an 8-way split, tools/rank_time.py, where those levels run as partial dispatches with short tails: the region kernel serves mips 1-3,
 * the level-in-LDS kernel mips >= 4; bench.py per-level times divided by 6 * size^2 * non-zero samples): mip 1 runs quarter-face
 * regions (more passes per tile), mip 3 has 33-cell faces (more samples on two faces), the small levels are launch / tail
 * limited.  The copy mip writes 16 B per texel at ~3.3 TB/s, i.e. as long as ~5.8 sample-evaluations per texel.
 * Used for balancing only; unit.cost stays the plain count. */
static double time_weight(const PBR_WorkUnit* u) {
    if (u->kind == PBR_Unit_Irradiance) return 1.24;
    switch (u->mip) {
    case 0: return 4.5;
    case 1: return 1.19;
    case 2: return 1.0;
    case 3: return 1.14;
    case 4: return 2.0;
    case 5: return 1.5;
    default: return 1.5;
    }
}

/* Faces 0 and 1 (+-X) contain the pole of the tangent frame (`some_vector`): neighbouring texels there take their samples at
 * different azimuths, so the lanes of a wave spread one sample over several regions of the source level and the tile's region
 * flags are wider.  Run alone, single-face dispatches of the region kernel take 1.06x, 1.06x, 1.25x as long there for mips 1-3
 * (tools/face_flags.sh, C4 shapes; tiles dealt round-robin over the XCDs and, on these two faces, outwards from the pole row so that
 * the long tiles start first -- in row order with an XCD-contiguous remap it was 1.23x / 1.23x / 1.9x); the weights below balance
 * the shares of a 4- and 8-way split (tools/rank_time.py). */
static double face_weight(const PBR_WorkUnit* u, uint32_t face) {
    static const double pole[6] = {1.0, 1.06, 1.06, 1.25, 1.1, 1.05};
    if (u->kind != PBR_Unit_Prefilter || face > 1 || u->mip > 5) return 1.0;
    return pole[u->mip];
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
    /* N ranks: contiguous shares of the row sequence.
     *
     * A launch needs thousands of workgroups in flight to hide the gather latency of the Monte-Carlo kernel, so a rank should
     * receive FEW, LARGE dispatches; balance, on the other hand, wants fine granularity.  Both are had by laying all rows of
     * all big levels end to end (level by level, face by face), weighting each row with the measured time per sample
     * evaluation of its level, and cutting that sequence into `world` shares of equal weighted cost: every rank gets at most
     * two partial faces plus runs of whole faces (merged into one dispatch per level), and the cut positions are exact to a
     * few rows.  Outputs are gathered on rank 0, so the copy level (75 % of the output bytes, 0.4 % of the compute) is pinned
     * to rank 0 and never crosses xGMI; levels cheaper than 1/16 of a share stay whole (their launches are latency-bound: six
     * single-face dispatches would take six times as long as one) and go to whichever rank has the most room left. */
    typedef struct { int irr; uint32_t mip, size; double raw_row, w_row[6], w_level; int whole; } Level;
    Level lv[40];
    uint32_t nl = 0;
    double total_w = 0.0;
    for (uint32_t m = 0; m <= mips && nl < 40; ++m) {
        int irr = m == mips;
        if (irr && !irradiance_size) break;
        Level* l = &lv[nl++];
        l->irr = irr; l->mip = irr ? 0 : m;
        l->size = irr ? irradiance_size : (specular_size >> m ? specular_size >> m : 1);
        PBR_WorkUnit probe; memset(&probe, 0, sizeof probe);
        probe.kind = irr ? PBR_Unit_Irradiance : PBR_Unit_Prefilter; probe.mip = l->mip;
        l->raw_row = (double)l->size * (irr ? 1024.0 : samples_per_texel(m));
        l->whole = 0; l->w_level = 0.0;
        for (uint32_t f = 0; f < 6; ++f) {
            l->w_row[f] = l->raw_row * time_weight(&probe) * face_weight(&probe, f);
            l->w_level += l->size * l->w_row[f];
        }
        total_w += l->w_level;
    }
    double budget = total_w / world;
    double* room = (double*)malloc(sizeof(double) * (size_t)world);
    for (int r = 0; r < world; ++r) room[r] = budget;
    uint32_t cap_all = 0, n = 0;
    for (uint32_t i = 0; i < nl; ++i) cap_all += 6 + 2 * (uint32_t)world;      /* per level: <= 6 faces, each cut splits one */
    PBR_WorkUnit* all = (PBR_WorkUnit*)malloc(sizeof(PBR_WorkUnit) * cap_all);
    int* owner = (int*)malloc(sizeof(int) * cap_all);
#define EMIT(L_, F0, F1, R0, R1, RANK) do { PBR_WorkUnit* u_ = &all[n]; \
        u_->kind = (L_)->irr ? PBR_Unit_Irradiance : PBR_Unit_Prefilter; u_->mip = (L_)->mip; \
        u_->face0 = (F0); u_->face1 = (F1); u_->row0 = (R0); u_->row1 = (R1); \
        u_->cost = (double)((F1) - (F0)) * ((R1) - (R0)) * (L_)->raw_row; owner[n++] = (RANK); } while (0)
    /* 1. whole levels: the pinned copy level, then the cheap ones (largest first; ties -> lowest rank) */
    for (uint32_t i = 0; i < nl; ++i) {
        Level* l = &lv[i];
        double cw = l->w_level;
        int pinned = !l->irr && l->mip == 0;
        if (!pinned && cw > budget / 16.0) continue;
        l->whole = 1;
        int best = 0;
        if (!pinned) for (int r = 1; r < world; ++r) if (room[r] > room[best]) best = r;
        EMIT(l, 0, 6, 0, l->size, best);
        room[best] -= cw;
    }
    const uint32_t n_whole = n;
    /* 2. the rows of the big levels.  Rank 0 first takes a contiguous share from the start of the sequence: the first levels
     *    have the most output bytes per unit of work (mip 1: 1389 samples per 16-byte texel; it is 3/4 of all bytes that could
     *    cross xGMI), and what rank 0 computes itself is already where the result is gathered.  The byte-heavy levels that
     *    remain (>= 5 % of the gathered bytes: mips 1 and 2) are striped over ranks 1..N-1 in proportion to their room, so
     *    every rank sends about the same number of bytes and the gather runs over all links at once; the compute-heavy rest
     *    (few bytes) is cut into contiguous shares again, which keeps its dispatches large.  Cuts sit on multiples of 16 rows
     *    (the kernel's tile height); what a rank leaves over or overdraws is carried to the next one. */
    double* quota = (double*)calloc((size_t)world, sizeof(double));
    double* frac = (double*)calloc((size_t)world, sizeof(double));
    double gathered_bytes = 0.0;
    for (uint32_t i = 0; i < nl; ++i) if (!lv[i].whole) gathered_bytes += 6.0 * lv[i].size * lv[i].size;
    int root_full = 0, cr = 1;                               /* cr: rank filling its contiguous share of the rest */
#define TAKE_ROWS(RANK, AVAIL) do { \
        left = l->size - row; take = left; cut = 0; \
        if ((RANK) < world - 1) { \
            double fit_ = *(AVAIL) / l->w_row[f]; \
            if (fit_ < (double)left) { \
                take = fit_ > 0.0 ? (uint32_t)(fit_ + 0.5) : 0; \
                if (left >= 32) take = (take + 8) & ~15u; \
                if (take >= left) take = left; else cut = 1; \
            } \
        } \
        if (take > 0) { \
            /* a whole face following whole faces of the same level on the same rank extends that dispatch */ \
            if (n > 0 && row == 0 && take == l->size && owner[n - 1] == (RANK) && \
                all[n - 1].kind == (uint32_t)(l->irr ? PBR_Unit_Irradiance : PBR_Unit_Prefilter) && all[n - 1].mip == l->mip && \
                all[n - 1].face1 == f && all[n - 1].row0 == 0 && all[n - 1].row1 == l->size) { \
                all[n - 1].face1 = f + 1; all[n - 1].cost += (double)take * l->raw_row; \
            } else { \
                EMIT(l, f, f + 1, row, row + take, (RANK)); \
            } \
            if ((AVAIL) != &room[(RANK)]) *(AVAIL) -= (double)take * l->w_row[f]; \
            room[(RANK)] -= (double)take * l->w_row[f]; \
            row += take; \
            if (row == l->size) { row = 0; ++f; } \
        } } while (0)
    for (uint32_t i = 0; i < nl; ++i) {
        Level* l = &lv[i];
        if (l->whole) continue;
        uint32_t f = 0, row = 0, left, take;
        int cut, was_cut_here = 0;
        if (!root_full) {
            while (f < 6) {
                TAKE_ROWS(0, &room[0]);
                if (cut) break;
            }
            if (f == 6) continue;                            /* rank 0 took the whole level and still has room */
            root_full = 1; was_cut_here = 1;
            room[1] += room[0];                              /* rank 0's rounding difference */
        }
        if (was_cut_here || 6.0 * l->size * l->size >= 0.05 * gathered_bytes) {
            /* striped over ranks 1..N-1 */
            double sum = 0.0;
            for (int r = 1; r < world; ++r) sum += room[r] > 0.0 ? room[r] : 0.0;
            for (int r = 1; r < world; ++r) frac[r] = sum > 0.0 ? (room[r] > 0.0 ? room[r] : 0.0) / sum : 1.0 / (world - 1);
            double rem = (double)(l->size - row) * l->w_row[f];
            for (uint32_t g = f + 1; g < 6; ++g) rem += (double)l->size * l->w_row[g];
            for (int r = 1; r < world; ++r) quota[r] = rem * frac[r];
            int r = 1;
            while (f < 6) {
                TAKE_ROWS(r, &quota[r]);
                if (cut && r < world - 1) { quota[r + 1] += quota[r]; ++r; }
            }
        } else {
            /* contiguous shares of what room the ranks have left */
            while (f < 6) {
                TAKE_ROWS(cr, &room[cr]);
                if (cut && cr < world - 1) { room[cr + 1] += room[cr]; ++cr; }
            }
        }
    }
#undef TAKE_ROWS
    free(quota); free(frac);
#undef EMIT
    /* A rank's list: its shares of the big levels, smallest level first, then the whole small levels.  The backend overlaps
     * these dispatches on side streams; the kernels of the smaller levels use 1024-thread workgroups with the source level in
     * LDS, which only find a CU with 16 free wave slots while the chip is not yet full of the 4-wave workgroups of the big
     * levels -- started last they would wait for those to drain and run alone at the end (measured: 19.2 instead of 17.8 ms
     * for a share holding two such dispatches).  The whole small levels are latency-bound and hide anywhere. */
    uint32_t written = 0;
    for (int m = 64; m >= -1; --m) {
        /* m = 64: irradiance shares, 63..0: prefilter shares by descending mip, -1: the whole levels of step 1 */
        for (uint32_t k = m < 0 ? 0 : n_whole; k < (m < 0 ? n_whole : n); ++k) {
            if (m >= 0 && (all[k].kind == PBR_Unit_Irradiance ? 64 : (int)all[k].mip) != m) continue;
            if (rank < 0 || owner[k] == rank) {
                if (written < capacity && out) out[written] = all[k];
                ++written;
            }
        }
    }
    free(room); free(all); free(owner);
    return written;
}
