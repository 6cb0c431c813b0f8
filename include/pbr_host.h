/*
 * pbr_host.h -- C host layer above the GPU_* boundary: the reference renderer's own call
 * sequences for the IBL precompute and the lighting pass, restated in C11 against gpu_hip.h so
 * that they run headless (no window, no assimp, no glslang).  Every function cites the reference
 * code whose GPU_* call sequence it reproduces; a reference maintainer could delete these and
 * keep calling GPU_* from render.cpp unchanged.
 */
#ifndef PBR_HOST_H
#define PBR_HOST_H

#include "gpux.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: Radiance .hdr (RGBE) decode, what stbi_loadf(..., 4) returns for asset_import.cpp:19
 * (third_party/stb_image.h:7130-7286): flat or new-RLE scanlines, rgb = mantissa * 2^(e-136),
 * alpha = 1, e == 0 -> 0.  Returns malloc'ed float RGBA [h][w][4] or NULL (reason in *err). */
float* PBR_DecodeHDR(const void* bytes, size_t size, int* w, int* h, const char** err);

/* asset_import.cpp:17-27: vertical strip of 6 square faces (height == 6*width) -> RGBA32F cubemap
 * with a full mip chain (GPU_MakeTexture uploads and generates the mips before returning). */
GPU_Texture* PBR_MakeTextureFromHDRIMemory(const void* bytes, size_t size);
GPU_Texture* PBR_MakeTextureFromHDRIFile(const char* filepath);

/* ---- extensions around the input/output files ----
 * Equirectangular .hdr (2:1) -> cubemap with mips; the strip loader above stays the reference path. */
GPU_Texture* PBR_MakeTextureFromEquirectHDRIMemory(const void* bytes, size_t size, uint32_t face_size);
GPU_Texture* PBR_MakeTextureFromEquirectHDRIFile(const char* filepath, uint32_t face_size);
/* Radiance RGBE writer (flat scanlines): returns malloc'ed file bytes, *out_size set. rgba: float [h][w][4] (alpha dropped). */
void* PBR_EncodeHDR(const float* rgba, int w, int h, size_t* out_size);
int   PBR_WriteHDRFile(const char* filepath, const float* rgba, int w, int h);
/* One mip of a cubemap as a vertical-strip .hdr (the layout MakeTextureFromHDRIFile reads back). Returns 0 on success. */
int   PBR_WriteCubeStripHDR(const char* filepath, GPU_Texture* cube, uint32_t mip_level);

/* ---- IBL precompute, render.cpp:505-619 ---- */
typedef struct PBR_IBLMaps {
    GPU_Texture* irradiance_map;        /* RGBA32F cube, render.cpp:794 (32x32) */
    GPU_Texture* brdf_lut;              /* RG16F 2D,    render.cpp:795 (256x256) */
    GPU_Texture* tex_specular_env_map;  /* RGBA32F cube + mips, render.cpp:796 (256x256) */
} PBR_IBLMaps;

/* render.cpp:794-796 with the sizes as parameters (reference: 32, 256, 256) */
void PBR_MakeIBLMaps(PBR_IBLMaps* maps, uint32_t irradiance_size, uint32_t lut_size, uint32_t specular_size);
void PBR_DestroyIBLMaps(PBR_IBLMaps* maps);

/* One work unit of the precompute = rows [row0,row1) of faces [face0,face1) of one output level. */
typedef enum PBR_UnitKind { PBR_Unit_Prefilter = 0, PBR_Unit_Irradiance = 1, PBR_Unit_BrdfLut = 2 } PBR_UnitKind;
typedef struct PBR_WorkUnit {
    uint32_t kind;                      /* PBR_UnitKind */
    uint32_t mip;                       /* prefilter output mip */
    uint32_t face0, face1, row0, row1;
    double   cost;                      /* texels * samples-per-texel (1 for the copy mip) */
} PBR_WorkUnit;

void PBR_GenIrradianceMap(GPU_Texture* tex_env_cube, GPU_Texture* irradiance_map);           /* render.cpp:505-540 */
/* render.cpp:542-589; min_size = 16 reproduces the reference's `if (size < 16) break;`, 1 runs the whole chain */
void PBR_GenPrefilteredEnvMap(GPU_Texture* tex_env_cube, GPU_Texture* tex_specular_env_map, uint32_t min_size);
void PBR_GenBRDFIntegrationMap(GPU_Texture* brdf_lut);                                       /* render.cpp:591-619 */
/* Records (does not submit) the dispatches of an explicit unit list into `graph`: the sharded form of the above.
 * `arena` receives the per-unit descriptor sets (caller resets it after GPU_GraphWait). */
typedef struct PBR_IBLPipelines PBR_IBLPipelines;
PBR_IBLPipelines* PBR_MakeIBLPipelines(void);
void PBR_DestroyIBLPipelines(PBR_IBLPipelines* p);
void PBR_RecordUnits(PBR_IBLPipelines* p, GPU_Graph* graph, GPU_DescriptorArena* arena, GPU_Texture* tex_env_cube,
                     const PBR_IBLMaps* maps, const PBR_WorkUnit* units, uint32_t unit_count);

/* Splits the precompute (prefilter mips down to min_size, optionally irradiance) over `world` ranks: the rows of
 * the big levels, laid end to end and weighted with the measured time per sample evaluation of their level, are cut into
 * `world` contiguous shares of equal cost, so a rank receives at most two partial faces plus runs of whole faces (one unit per
 * run); the copy level is pinned to rank 0 (where results are gathered), cheap levels stay whole.  world = 1: one unit per level.
 * Returns the number of units written to out (<= capacity) for `rank`; rank < 0 lists all units.  Deterministic. */
uint32_t PBR_PartitionIBL(uint32_t specular_size, uint32_t min_size, uint32_t irradiance_size, uint32_t env_size,
                          int world, int rank, PBR_WorkUnit* out, uint32_t capacity);

/* ---- the exchange step of the multi-GPU job: RCCL over xGMI, one grouped batch of point-to-point transfers.
 * The reference has no counterpart (single queue, gpu_vulkan.c:1040-1106).  `nccl_comm` is the caller's ncclComm_t (opaque:
 * the library does not own the bootstrap), `stream` the hipStream_t the transfers are enqueued on -- pass
 * GPUX_GraphStream(graph) after GPU_GraphSubmit(graph) and the exchange follows the graph's kernels without a host round
 * trip; GPU_GraphWait(graph) then also waits for the exchange.  Negative return values are PBR_E_*. ---- */
enum { PBR_OK = 0, PBR_E_BADARG = -1, PBR_E_COMM = -2 };
typedef struct PBR_XferRange { void* ptr; uint64_t bytes; int peer; } PBR_XferRange;    /* device pointer, byte count, peer rank */
/* RCCL is bound at first use (dlopen), never at link time: libgpu_hip.so has no librccl dependency, and the exchange calls the
 * copy of librccl the process already holds -- the one that made `nccl_comm`.  Search order: PBR_SetRcclLibrary(path) or
 * env PBR_RCCL_LIB; an already mapped "librccl.so.1"; the process's global symbols; a fresh dlopen of librccl.so.1.
 * PBR_SetRcclLibrary(NULL) forgets the binding (next use searches again).  PBR_RcclInfo: ncclGetVersion() code (e.g. 22707) and
 * the file the entry points were bound from.  PBR_CommInfo: ncclCommCount / ncclCommUserRank of the caller's communicator. */
int PBR_SetRcclLibrary(const char* path);
int PBR_RcclInfo(int* version, const char** path);
int PBR_CommInfo(void* nccl_comm, int* count, int* user_rank);
/* ncclGroupStart; ncclRecv x n_recvs; ncclSend x n_sends; ncclGroupEnd (a rank may send to itself).  If a call inside the group
 * fails, the group is still closed before PBR_E_COMM is returned (no dangling group on this thread); the communicator must then
 * be treated as dead. */
int PBR_ExchangeRanges(void* nccl_comm, void* stream, const PBR_XferRange* sends, uint32_t n_sends,
                       const PBR_XferRange* recvs, uint32_t n_recvs);
/* The bytes of one work unit inside its texture: contiguous in the [mip][face][y][x] layout (rows of one face, or whole faces). */
int PBR_UnitByteRange(const PBR_IBLMaps* maps, const PBR_WorkUnit* unit, GPU_Texture** texture, uint64_t* offset, uint64_t* bytes);
/* After every rank has run its share (PBR_PartitionIBL(..., world, rank) + PBR_RecordUnits): ranks != root send their units,
 * root receives all of them in place, so that `maps` on root holds the whole result.  Every rank must pass the same
 * min_size / env_size it partitioned with.  Returns the bytes this rank sent or received (0 when world == 1). */
int64_t PBR_GatherUnits(void* nccl_comm, void* stream, int root, int world, int rank, const PBR_IBLMaps* maps,
                        uint32_t min_size, uint32_t env_size);
/* The same for a subset of the levels: bit l of level_mask = prefilter mip l, PBR_LEVEL_IRRADIANCE = the irradiance map.
 * All ranks must call the phases in the same order (one communicator: RCCL keeps issue order). */
#define PBR_LEVEL_IRRADIANCE 0x80000000u
int64_t PBR_GatherUnitsMasked(void* nccl_comm, void* stream, int root, int world, int rank, const PBR_IBLMaps* maps,
                              uint32_t min_size, uint32_t env_size, uint32_t level_mask);
/* The transfers PBR_GatherUnitsMasked would enqueue for `rank` (root: receives from every peer; others: sends to root), without
 * a communicator: out may be NULL to count.  Returns the number of ranges. */
int64_t PBR_GatherPlan(int root, int world, int rank, const PBR_IBLMaps* maps, uint32_t min_size, uint32_t env_size,
                       uint32_t level_mask, PBR_XferRange* out, uint32_t capacity);
/* units[i] whose level is in level_mask, in order; out may alias nothing; returns the count */
uint32_t PBR_SelectUnits(const PBR_WorkUnit* units, uint32_t n, uint32_t level_mask, PBR_WorkUnit* out);
/* One rank's whole share with the exchange overlapped: the units of `early_mask` (the level that is most bytes on the wire: mip 1,
 * i.e. early_mask = 2) are recorded into g_early behind whatever the caller already recorded there (the source's
 * GPU_OpGenerateMipmaps), the rest into g_late; both are submitted, then the early units travel on g_early's stream while
 * g_late computes, the late ones follow g_late.  The caller waits: GPU_GraphWait(g_late); GPU_GraphWait(g_early); and resets
 * the arena.  world == 1 runs both graphs and moves nothing.  Returns bytes sent / received by this rank, or PBR_E_*.
 * After a NEGATIVE return both graphs are nevertheless submitted: the caller must still GPU_GraphWait both, and must treat the
 * communicator as dead (a failed early exchange means the late one was never issued; peers may be waiting in theirs). */
int64_t PBR_RunPartitionedIBL(PBR_IBLPipelines* p, GPU_Graph* g_early, GPU_Graph* g_late, GPU_DescriptorArena* arena,
                              GPU_Texture* tex_env_cube, const PBR_IBLMaps* maps, void* nccl_comm, int root, int world, int rank,
                              uint32_t min_size, uint32_t early_mask);
/* Screen-band split of the shade pass (C5): rank r owns rows [height*r/world, height*(r+1)/world) of the frame */
void PBR_BandRows(uint32_t height, int world, int rank, uint32_t* row0, uint32_t* row1);
int64_t PBR_GatherBands(void* nccl_comm, void* stream, int root, int world, int rank, GPU_Texture* frame);

/* ---- camera + Globals: utils/camera.h:95-120 and render.cpp:962-991 ---- */
typedef struct PBR_Globals {            /* RendererGlobalsBuffer, render.h:122-136; column-major mat4 */
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
} PBR_Globals;

/* ori_xyzw == NULL -> the reference's default orientation (faces +Y, utils/camera.h:45). */
void PBR_FillGlobals(PBR_Globals* out, const float pos[3], const float ori_xyzw[4], float fov_degrees, float aspect,
                     float z_near, float z_far, float sun_angle_x_deg, float sun_angle_y_deg, uint32_t frame_idx);

/* ---- lighting pass: layout/descriptor set of render.cpp:829-871, pass of :716-723, draw of :1119-1127 ---- */
typedef struct PBR_GBuffer {
    GPU_Texture* base_color; GPU_Texture* normal; GPU_Texture* orm; GPU_Texture* emissive; GPU_Texture* depth;   /* render.cpp:680-687 */
    GPU_Texture* lighting_result;                                                                                /* render.cpp:693 */
} PBR_GBuffer;
void PBR_MakeGBuffer(PBR_GBuffer* gb, uint32_t width, uint32_t height, GPU_Format result_format);
void PBR_DestroyGBuffer(PBR_GBuffer* gb);

typedef struct PBR_LightingPass PBR_LightingPass;
PBR_LightingPass* PBR_MakeLightingPass(const PBR_GBuffer* gb, const PBR_IBLMaps* maps, uint32_t width, uint32_t height);
/* the same with the sun depth map of the shadow pass bound to SUN_DEPTH_MAP (render.cpp:676, :863; D32F); NULL = 1x1 stand-in.
 * GPUX_SetShadeFlags(PBR_LightingPipeline(lp), ... | GPUX_Shade_SunShadows) makes the pass read it. */
PBR_LightingPass* PBR_MakeLightingPassEx(const PBR_GBuffer* gb, const PBR_IBLMaps* maps, uint32_t width, uint32_t height, GPU_Texture* sun_depth_map);
/* all raster-fed inputs of the live shader (render.cpp:861-863): the swept light grid (PBR_LightgridTexture), the previous frame
 * as the lighting pass sees it (the reference binds bloom_downscale_rt: PBR_PostBloomDownscale) and the sun depth map; NULLs = stand-ins.
 * GPUX_Shade_LightShafts | GPUX_Shade_SunShadows | GPUX_Shade_VoxelGI then runs the reference's complete live shader. */
PBR_LightingPass* PBR_MakeLightingPassLive(const PBR_GBuffer* gb, const PBR_IBLMaps* maps, uint32_t width, uint32_t height,
                                           GPU_Texture* sun_depth_map, GPU_Texture* lightgrid, GPU_Texture* prev_frame_result);
void PBR_DestroyLightingPass(PBR_LightingPass* lp);
GPU_Buffer* PBR_LightingGlobalsBuffer(PBR_LightingPass* lp);           /* persistently mapped (render.cpp:675) */
GPU_GraphicsPipeline* PBR_LightingPipeline(PBR_LightingPass* lp);
/* render.cpp:977-991 + 1119-1127: copies globals into the mapped buffer and records the pass; rows [row0,row1), row1 == 0 -> all */
void PBR_RecordLightingPass(PBR_LightingPass* lp, GPU_Graph* graph, const PBR_Globals* globals, uint32_t row0, uint32_t row1);

/* ---- voxel light grid + its sweep pass: render.cpp:678 (image), :816 (IMG0 binding), :151-187 (pipeline,
 *      descriptor set), :1028 (clear), :1061-1072 (per-frame sweep) ---- */
typedef struct PBR_Lightgrid PBR_Lightgrid;
PBR_Lightgrid* PBR_MakeLightgrid(uint32_t size);                 /* reference: LIGHTGRID_SIZE 128 (render.cpp:7); size >= 128, multiple of 8 */
void PBR_DestroyLightgrid(PBR_Lightgrid* lg);
GPU_Texture* PBR_LightgridTexture(PBR_Lightgrid* lg);
uint32_t PBR_LightgridSweepDirection(const PBR_Lightgrid* lg);   /* direction used by the last recorded sweep */
void PBR_RecordLightgridClear(PBR_Lightgrid* lg, GPU_Graph* graph);
/* advances the direction (1, 2, 0, 1, ... from a fresh grid, as render.cpp:1064-1065) and records the full dispatch */
void PBR_RecordLightgridSweep(PBR_Lightgrid* lg, GPU_Graph* graph);
/* sharded form: invocations (iy, iz) in [y0,y1) x [z0,z1) of an explicit direction; lines are independent */
void PBR_RecordLightgridSweepLines(PBR_Lightgrid* lg, GPU_Graph* graph, uint32_t direction, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1);

/* ---- post-process tail: TAA resolve (render.cpp:281-337, 690-697, 732-739, 1131-1137) and the final
 *      tone-map pass (render.cpp:456-501, 782-785, 1181-1187).  frame_idx selects the ping-pong half exactly as
 *      frame_idx_mod2 does in the reference: frame i resolves into taa_output_rt[i%2] reading taa_output_rt[1-i%2]. ---- */
typedef struct PBR_PostProcess PBR_PostProcess;
/* gb supplies GBUFFER_DEPTH and LIGHTING_RESULT (RGBA16F); backbuffer_format stands in for the swapchain (RGBA8UN / BGRA8UN, or a float format) */
PBR_PostProcess* PBR_MakePostProcess(const PBR_GBuffer* gb, uint32_t width, uint32_t height, GPU_Format backbuffer_format);
void PBR_DestroyPostProcess(PBR_PostProcess* pp);
GPU_Texture* PBR_PostVelocity(PBR_PostProcess* pp, uint32_t frame_idx_mod2);     /* gbuffer_velocity[i]: RG16F, uploaded by the caller */
GPU_Texture* PBR_PostTaaOutput(PBR_PostProcess* pp, uint32_t frame_idx_mod2);    /* taa_output_rt[i] */
GPU_Texture* PBR_PostBackbuffer(PBR_PostProcess* pp);
void PBR_RecordTaaResolve(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx);
void PBR_RecordTaaResolveRows(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx, uint32_t row0, uint32_t row1);   /* rows [row0,row1) only */
void PBR_RecordFinalPostProcess(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx);
/* bloom chain between the two (render.cpp:741-777, 340-454, 1139-1176): BLOOM_PASS_COUNT (6, or fewer on tiny frames) 13-tap
 * downsamples into the mips of a half-size target, clear + blit of the TAA result into the full-size target, as many additive
 * tent upsamples; then the final pass reading that target, as the reference binds it (render.cpp:478). */
void PBR_RecordBloom(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx);
void PBR_RecordFinalPostProcessBloom(PBR_PostProcess* pp, GPU_Graph* graph, uint32_t frame_idx);
GPU_Texture* PBR_PostBloomDownscale(PBR_PostProcess* pp);
GPU_Texture* PBR_PostBloomUpscale(PBR_PostProcess* pp);
uint32_t PBR_PostBloomPassCount(const PBR_PostProcess* pp);

#ifdef __cplusplus
}
#endif
#endif
