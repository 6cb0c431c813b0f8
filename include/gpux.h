/*
 * gpux.h -- extensions of the HIP backend beyond the reference's GPU_* API (new names only; nothing
 * in gpu_hip.h changes meaning).  They exist for what the reference cannot express:
 * multi-GPU sharding (dispatch of a (face,row) sub-range, device selection), reading back mips
 * other than 0 (the reference's GPU_OpCopyTextureToBuffer is mip-0 only, gpu_vulkan.c:2945-2951),
 * parameterised sizes/sample counts (literals inside the reference shaders), per-op HIP-event
 * timing, and handing device pointers to a communication library (RCCL).
 */
#ifndef GPUX_H
#define GPUX_H

#include "gpu_hip.h"

/* ---- device / errors ---- */
GPU_API void GPUX_SetDevice(int hip_device_index);        /* call before GPU_Init; default: $LOCAL_RANK or 0 */
GPU_API int  GPUX_GetDevice(void);
typedef void (*GPUX_ErrorHandler)(const char* message, void* user);
/* With a handler installed, a failing GPU_* call reports and returns (NULL / no-op) instead of aborting. */
GPU_API void GPUX_SetErrorHandler(GPUX_ErrorHandler handler, void* user);
GPU_API const char* GPUX_BackendName(void);               /* "hip-gfx950" */

/* ---- push-constant block understood by the IBL kernels when size == sizeof(GPUX_IBLConstants).
 * A 4-byte block keeps the reference meaning (int mip_level; gen_prefiltered_env_map.glsl:99-101):
 * roughness = {0,.03,.15,.4,.6}[mip] (mips >= 5: min(1, .6 + .08*(mip-4)), an extension),
 * source LOD = 1 for mip 0 else 3 + mip (clamped to the env chain), 8192 samples. ---- */
typedef struct GPUX_IBLConstants {
    int32_t mip_level;        /* prefilter: which branch (0 = copy, else Monte-Carlo) */
    float   roughness;        /* prefilter MC roughness */
    float   src_lod;          /* integer-valued source LOD */
    int32_t sample_count;     /* 0 = shader default (8192 prefilter / 1024 irradiance / 4096 LUT) */
} GPUX_IBLConstants;

/* ---- sub-range dispatch: like GPU_OpDispatch over the bound OUTPUT image, restricted to faces
 * [face0,face1) and rows [row0,row1) (rows of the LUT for gen_brdf_integration_map). ---- */
GPU_API void GPUX_OpDispatchRows(GPU_Graph* graph, uint32_t face0, uint32_t face1, uint32_t row0, uint32_t row1);
/* the same for the light-grid sweep pipeline (lightgrid_sweep.glsl): invocations (iy, iz) in [y0,y1) x [z0,z1),
 * where GPU_OpDispatch(1, gy, gz) covers [0, 8 gy) x [0, 8 gz); the direction comes from the push constant. */
GPU_API void GPUX_OpDispatchLines(GPU_Graph* graph, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1);

/* ---- shade pass controls ---- */
enum { GPUX_Shade_IBL = 1 << 0, GPUX_Shade_LightShafts = 1 << 1,
       GPUX_Shade_SunShadows = 1 << 2, /* SUN_DEPTH_MAP: 4-tap PCF sun shadow + shaft visibility (lighting_pass.glsl:594-608, 646) */
       GPUX_Shade_VoxelGI = 1 << 3     /* LIGHTGRID + PREV_FRAME_RESULT + GBUFFER_DEPTH: the live ambient / specular traces (:273-424, :685, :701);
                                          with LightShafts | SunShadows | VoxelGI the pass is the reference's complete live shader */ };
GPU_API void GPUX_SetShadeFlags(GPU_GraphicsPipeline* pipeline, int flags);    /* default GPUX_Shade_IBL */
/* full-screen draw restricted to rows [row0,row1) (screen-band sharding) */
GPU_API void GPUX_OpDrawRows(GPU_Graph* graph, uint32_t row0, uint32_t row1);

/* ---- transfers the reference API lacks ---- */
GPU_API void GPUX_OpCopyTextureMipToBuffer(GPU_Graph* graph, GPU_Texture* src, uint32_t mip_level, GPU_Buffer* dst, uint32_t dst_offset);
GPU_API void GPUX_OpCopyBufferToTextureMip(GPU_Graph* graph, GPU_Buffer* src, uint32_t src_offset, GPU_Texture* dst, uint32_t mip_level);
GPU_API uint64_t GPUX_TextureMipBytes(const GPU_Texture* texture, uint32_t mip_level);   /* all layers of one mip */
GPU_API void* GPUX_TextureDevicePtr(GPU_Texture* texture, uint32_t mip_level);            /* for RCCL / interop */
GPU_API void* GPUX_BufferDevicePtr(GPU_Buffer* buffer);
GPU_API void* GPUX_GraphStream(GPU_Graph* graph);                                         /* hipStream_t */

/* ---- textures over caller-owned HBM (e.g. a torch tensor that RCCL gathers into): same layout as GPU_MakeTexture,
 * [mip][layer][y][x] tight; the caller keeps the allocation alive and frees it after GPU_DestroyTexture. ---- */
GPU_API GPU_Texture* GPUX_MakeTextureExternal(GPU_Format format, uint32_t width, uint32_t height, uint32_t depth, GPU_TextureFlags flags,
                                              void* device_memory, uint64_t device_bytes);
/* Tell the backend that the texture's memory was written behind its back (RCCL receive, torch, any other library working on
 * GPUX_TextureDevicePtr / external memory): the lazily built sampler twins (apron, cells) are rebuilt before the next use. */
GPU_API void GPUX_InvalidateTexture(GPU_Texture* texture);
GPU_API uint64_t GPUX_TextureTotalBytes(const GPU_Texture* texture);
GPU_API uint64_t GPUX_TextureMipOffset(const GPU_Texture* texture, uint32_t mip_level);

/* ---- equirectangular input (extension; the reference only loads cube strips): uploads the RGBA32F panorama, converts it
 * to a face_size^2 cubemap (K6) and generates the mip chain before returning, like GPU_MakeTexture(data != NULL). ---- */
GPU_API GPU_Texture* GPUX_MakeCubemapFromEquirect(const void* rgba32f, uint32_t width, uint32_t height, uint32_t face_size, GPU_TextureFlags extra_flags);

/* ---- per-op timing with HIP events on the graph's own stream ---- */
GPU_API void GPUX_EnableOpTiming(int enable);
GPU_API uint32_t GPUX_GraphTimedOpCount(GPU_Graph* graph);          /* ops of the last waited submission */
GPU_API const char* GPUX_GraphTimedOpName(GPU_Graph* graph, uint32_t index);
GPU_API float GPUX_GraphTimedOpMs(GPU_Graph* graph, uint32_t index);
/* ---- tolerance-budgeted sample cut of the specular prefilter (opt-in, NEVER the default: the reference sums every sample) ----
 * The Monte-Carlo weights of a low-roughness level decay like exp(-i / 14.7): of mip 1's 1389 non-zero fp32 weights the last ~900
 * together carry less than 1e-13 of the sum.  With rel > 0 a prefilter dispatch keeps the first K samples, K the smallest count with
 *     (sum of dropped weights) * max(source level)  <=  rel * (sum of kept weights) * min(source level)
 * -- a rigorous bound on every texel's relative error (taps are convex combinations of the level's texels), evaluated on the host
 * from the weight table and the level's measured range (one reduction + read-back per source level and content change).  A level
 * with a zero or negative texel is never cut.  rel = 0 (default) restores the exact sum.  GPUX_PrefilterKeptSamples(mip): what the
 * last dispatch of that output mip kept (0: none recorded). */
GPU_API void GPUX_SetPrefilterTolerance(float rel);
GPU_API int GPUX_PrefilterKeptSamples(uint32_t mip_level);
/* busy span of the last waited submission on the graph's main stream: one event pair from before its first op to after the
 * last join of its side streams (overlapping dispatches are not counted twice, unlike the sum of the per-op times) */
GPU_API float GPUX_GraphSpanMs(GPU_Graph* graph);

/* ---- overlap of small precompute dispatches: consecutive row-ranged K3/K4 dispatches (GPUX_OpDispatchRows) of fewer than
 * 2M texels whose outputs are disjoint and that do not read each other's output are spread over `count` side streams
 * between a fork and a join event; anything else executes in recording order on the graph's stream.  0 or 1 turns it off,
 * a negative count restores the default (environment PBR_TILE_STREAMS, else 4).  Results do not depend on the setting. ---- */
GPU_API void GPUX_SetTileStreams(int count);

/* ---- hipGraph replay of the per-frame chain: with replay on, GPU_GraphSubmit captures the launches of a graph whose ops are all
 * launch-only (light-grid sweep, draws of the shade / TAA / bloom / tone-map pipelines, blits, memset clears, mip generation) and
 * whose sampler twins already exist into a hipGraph, updates the executable graph kept from this GPU_Graph's previous submission
 * in place (same chain, new arguments) and launches it: one dispatch of ~25 dependent kernels instead of 25.  Every other graph, and
 * every submission while per-op timing is on, runs as before.  Results are identical.  enable < 0: environment PBR_GRAPH_REPLAY, else off. ---- */
GPU_API void GPUX_SetGraphReplay(int enable);
GPU_API void GPUX_GraphReplayStats(GPU_Graph* graph, uint64_t* launches, uint64_t* updates, uint64_t* instantiations);
/* 1:1 blits folded into the additive bloom draw that consumes their copy, since the library was loaded (env PBR_GRAPH_FOLD=0 keeps every blit) */
GPU_API uint64_t GPUX_FoldedBlitCount(void);
/* Graphs in flight: the leading ops of a graph that share no texture with the TAIL of the graph submitted before it (from its first bloom
 * draw on) start once that graph's head is done instead of after its end (GPU_GraphSubmit); everything else keeps submission order.
 * on: 1 / 0, -1 = default (env PBR_GRAPH_OVERLAP, else 1).  GPUX_OverlappedSubmitCount: submissions that started this way. */
GPU_API void GPUX_SetGraphOverlap(int on);
GPU_API uint64_t GPUX_OverlappedSubmitCount(void);

#endif
