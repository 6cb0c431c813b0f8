/*
 * gpu_hip.h -- the drop-in C boundary of the MI355X (gfx950 / HIP) backend.
 *
 * The reference renderer talks to its GPU through the `GPU_*` C API of src/gpu/gpu.h (implemented
 * once, for Vulkan, in src/gpu/gpu_vulkan.c).  This header declares the same entry points -- same
 * names, argument order, struct layouts and enum numbering, so existing callers (render.cpp,
 * asset_import.cpp, main.cpp) compile against it unchanged -- and libgpu_hip.so implements them
 * with hand-written HIP kernels for the image-based-lighting precompute and the deferred shade pass.
 * Each declaration cites the reference line it replaces as  [gpu.h:N].
 *
 * Scope: compute pipelines resolve to built-in kernels by shader identity
 * (gen_brdf_integration_map / gen_irradiance_map / gen_prefiltered_env_map); the only "graphics"
 * pipeline understood is the full-screen lighting_pass.glsl draw.  Raster-only entry points exist
 * so that callers link, and fail loudly ("GPU-ERROR: ... unsupported (raster)") when used.
 * Error convention is the reference's: no error codes; backend failures and API misuse print
 * "GPU-ERROR: ..." to stderr and abort (gpu_vulkan.c:387-392), unless a handler is installed with
 * GPUX_SetErrorHandler (gpux.h).  Single-threaded API, like the reference (gpu_vulkan.c:310).
 */
#ifndef GPU_HIP_H
#define GPU_HIP_H
#define GPU_INCLUDED 1

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define GPU_API extern "C"
#define GPU_LangAgnosticLiteral(T) T
#else
#define GPU_API
#define GPU_LangAgnosticLiteral(T) (T)
#endif

/* The reference requires its arena allocator header first (gpu.h:4-6); only the pointer type is
 * needed here (GPU_SPIRVFromGLSL / GPU_JoinGLSLErrorString receive an arena they never touch). */
#ifndef FIRE_DS_INCLUDED
typedef struct DS_Arena DS_Arena;
#endif

/* ---- opaque handles [gpu.h:11-18] ---- */
typedef struct GPU_RenderPass GPU_RenderPass;
typedef struct GPU_GraphicsPipeline GPU_GraphicsPipeline;
typedef struct GPU_ComputePipeline GPU_ComputePipeline;
typedef struct GPU_PipelineLayout GPU_PipelineLayout;
typedef struct GPU_DescriptorSet GPU_DescriptorSet;
typedef struct GPU_Sampler GPU_Sampler;
typedef struct GPU_DescriptorArena GPU_DescriptorArena;
typedef struct GPU_Graph GPU_Graph;

typedef struct GPU_String { const char* data; size_t length; } GPU_String;          /* [gpu.h:20-23] */
#define GPU_STR(x) GPU_LangAgnosticLiteral(GPU_String){x, sizeof(x) - 1}              /* [gpu.h:31] */

/* ---- pixel formats: numeric order is ABI [gpu.h:51-94] ---- */
typedef enum GPU_Format {
    GPU_Format_Invalid,
    GPU_Format_R8UN, GPU_Format_RG8UN, GPU_Format_RGBA8UN, GPU_Format_BGRA8UN,
    GPU_Format_R16F, GPU_Format_RG16F, GPU_Format_RGB16F, GPU_Format_RGBA16F,
    GPU_Format_R32F, GPU_Format_RG32F, GPU_Format_RGB32F, GPU_Format_RGBA32F,
    GPU_Format_R8I, GPU_Format_R16I, GPU_Format_RG16I, GPU_Format_RGBA16I,
    GPU_Format_R32I, GPU_Format_RG32I, GPU_Format_RGB32I, GPU_Format_RGBA32I, GPU_Format_R64I,
    GPU_Format_D16UN, GPU_Format_D32F_Or_X8D24UN, GPU_Format_D32FS8I_Or_D24UNS8I, GPU_Format_D24UNS8I_Or_D32FS8I,
    GPU_Format_BC1_RGB_UN, GPU_Format_BC1_RGBA_UN, GPU_Format_BC3_RGBA_UN, GPU_Format_BC5_UN
} GPU_Format;

typedef struct GPU_FormatInfo {                                                      /* [gpu.h:33-46] */
    uint32_t block_extent;
    uint32_t block_size;
    bool sampled, vertex_input, color_target, depth_target, stencil_target;
    bool is_int;
    const char* glsl;
} GPU_FormatInfo;

/* Bytes per block / block extent and capabilities per format [gpu.h:99-144]. Exported by the
 * library (the reference defines it `static` in the header). */
GPU_API GPU_FormatInfo GPUX_GetFormatInfo(GPU_Format format);
#ifndef GPU_NO_FORMAT_INFO_ALIAS
#define GPU_GetFormatInfo GPUX_GetFormatInfo
#endif

#define GPU_SWAPCHAIN_FORMAT GPU_Format_BGRA8UN                                      /* [gpu.h:146] */

typedef enum GPU_ShaderStage { GPU_ShaderStage_Vertex, GPU_ShaderStage_Fragment, GPU_ShaderStage_Compute } GPU_ShaderStage;   /* [gpu.h:148-152] */
typedef enum GPU_CullMode { GPU_CullMode_TwoSided, GPU_CullMode_DrawCW, GPU_CullMode_DrawCCW } GPU_CullMode;                    /* [gpu.h:154-158] */
typedef enum GPU_LayoutHint {                                                        /* [gpu.h:160-166] */
    GPU_LayoutHint_RenderTarget, GPU_LayoutHint_ShaderRead, GPU_LayoutHint_TransferSrc,
    GPU_LayoutHint_TransferDest, GPU_LayoutHint_Present
} GPU_LayoutHint;

typedef int GPU_BufferFlags;
typedef enum GPU_BufferFlag {                                                        /* [gpu.h:169-173] */
    GPU_BufferFlag_CPU = 1 << 0,            /* `data` is a persistently mapped host pointer (pinned, device visible) */
    GPU_BufferFlag_GPU = 1 << 1,
    GPU_BufferFlag_StorageBuffer = 1 << 2
} GPU_BufferFlag;

typedef int GPU_TextureFlags;
typedef enum GPU_TextureFlag {                                                       /* [gpu.h:176-186] */
    GPU_TextureFlag_StorageImage = 1 << 0,
    GPU_TextureFlag_RenderTarget = 1 << 1,
    GPU_TextureFlag_HasMipmaps = 1 << 2,
    GPU_TextureFlag_Cubemap = 1 << 3,
    GPU_TextureFlag_MSAA2x = 1 << 4,
    GPU_TextureFlag_MSAA4x = 1 << 5,
    GPU_TextureFlag_MSAA8x = 1 << 6,
    GPU_TextureFlag_PerMipBinding = 1 << 7,
    GPU_TextureFlag_SwapchainTarget = 1 << 8
} GPU_TextureFlag;

typedef struct GPU_Texture {                                                         /* [gpu.h:188-194] */
    uint32_t width, height, depth;
    uint32_t layer_count;
    uint32_t mip_level_count;
    GPU_Format format;
    GPU_TextureFlags flags;
} GPU_Texture;

typedef struct GPU_Buffer {                                                          /* [gpu.h:196-200] */
    GPU_BufferFlags flags;
    uint32_t size;
    void* data;
} GPU_Buffer;

typedef struct GPU_TextureView { GPU_Texture* texture; uint32_t mip_level; } GPU_TextureView;   /* [gpu.h:202-205] */
#define GPU_SWAPCHAIN_COLOR_TARGET ((GPU_TextureView*)-1)                            /* [gpu.h:207] */

typedef struct GPU_RenderPassDesc {                                                  /* [gpu.h:209-219] */
    uint32_t color_targets_count;
    GPU_TextureView* color_targets;
    GPU_TextureView* msaa_color_resolve_targets;
    uint32_t width, height;
    GPU_Texture* depth_stencil_target;
} GPU_RenderPassDesc;

typedef int GPU_AccessFlags;
typedef enum GPU_AccessFlag { GPU_AccessFlag_Read = 1 << 0, GPU_AccessFlag_Write = 1 << 1 } GPU_AccessFlag;   /* [gpu.h:222-225] */
typedef struct GPU_Access { GPU_AccessFlags flags; uint32_t binding; } GPU_Access;   /* [gpu.h:227-230] */

typedef bool (*GPU_ShaderIncluderFn)(DS_Arena* arena, GPU_String filepath, GPU_String* out_source, void* ctx);   /* [gpu.h:232] */

typedef struct GPU_ShaderDesc {                                                      /* [gpu.h:234-245] */
    GPU_Access* accesses;
    uint32_t accesses_count;
    GPU_String glsl_debug_filepath;     /* the HIP backend identifies the built-in kernel by this file's basename */
    GPU_ShaderIncluderFn glsl_includer;
    void* glsl_includer_ctx;
    GPU_String spirv;                   /* opaque kernel token returned by GPU_SPIRVFromGLSL */
    GPU_String glsl;
} GPU_ShaderDesc;

typedef struct GPU_GraphicsPipelineDesc {                                            /* [gpu.h:247-267] */
    GPU_PipelineLayout* layout;
    GPU_RenderPass* render_pass;
    GPU_ShaderDesc vs;
    GPU_ShaderDesc fs;
    GPU_Format* vertex_input_formats;
    uint32_t vertex_input_formats_count;
    bool enable_depth_test;
    bool enable_depth_write;
    bool enable_blending;
    bool blending_mode_additive;
    bool enable_conservative_rasterization;
    GPU_CullMode cull_mode;
} GPU_GraphicsPipelineDesc;

typedef void* GPU_WindowHandle;                                                      /* [gpu.h:271] ignored: headless */
typedef struct GPU_Color { float r, g, b, a; } GPU_Color;                            /* [gpu.h:273-278] */
typedef struct GPU_Offset3D { int32_t x, y, z; } GPU_Offset3D;                       /* [gpu.h:280-282] */
typedef enum GPU_Filter { GPU_Filter_Linear = 0, GPU_Filter_Nearest } GPU_Filter;    /* [gpu.h:284-287] */
typedef enum GPU_AddressMode { GPU_AddressMode_Wrap = 0, GPU_AddressMode_Clamp, GPU_AddressMode_Mirror } GPU_AddressMode;   /* [gpu.h:289-293] */
typedef enum GPU_CompareOp {                                                         /* [gpu.h:295-304] */
    GPU_CompareOp_Never = 0, GPU_CompareOp_Less, GPU_CompareOp_Equal, GPU_CompareOp_LessOrEqual,
    GPU_CompareOp_Greater, GPU_CompareOp_NotEqual, GPU_CompareOp_GreaterOrEqual, GPU_CompareOp_Always
} GPU_CompareOp;

typedef struct GPU_SamplerDesc {                                                     /* [gpu.h:306-315] */
    GPU_Filter min_filter, mag_filter, mipmap_mode;
    GPU_AddressMode address_modes[3];
    float mip_lod_bias, min_lod, max_lod;
    GPU_CompareOp compare_op;
} GPU_SamplerDesc;

typedef struct GPU_OpBlitInfo {                                                      /* [gpu.h:317-327] */
    GPU_Filter filter;
    GPU_Texture* src_texture;
    GPU_Texture* dst_texture;
    uint32_t src_layer, dst_layer;
    uint32_t src_mip_level, dst_mip_level;
    GPU_Offset3D src_area[2];
    GPU_Offset3D dst_area[2];
} GPU_OpBlitInfo;

typedef struct GPU_GLSLError { GPU_ShaderStage shader_stage; uint32_t line; GPU_String error_message; } GPU_GLSLError;   /* [gpu.h:329-333] */
typedef struct GPU_GLSLErrorArray { GPU_GLSLError* data; uint32_t length; } GPU_GLSLErrorArray;                          /* [gpu.h:335-338] */
typedef uint32_t GPU_Binding;                                                        /* [gpu.h:340] */
#define GPU_MIP_LEVEL_ALL 0xFFFFFFFF                                                 /* [gpu.h:498] */

/* ================================ lifetime ================================ */
GPU_API void GPU_Init(GPU_WindowHandle window);                                      /* [gpu.h:354] picks the HIP device (GPUX_SetDevice / LOCAL_RANK) */
GPU_API void GPU_Deinit(void);                                                       /* [gpu.h:355] */
GPU_API void GPU_WaitUntilIdle(void);                                                /* [gpu.h:462] hipDeviceSynchronize */

/* ================================ samplers ================================ */
/* Shared sampler objects: never pass them to GPU_DestroySampler [gpu.h:359-365]. */
GPU_API GPU_Sampler* GPU_SamplerLinearWrap(void);
GPU_API GPU_Sampler* GPU_SamplerLinearClamp(void);
GPU_API GPU_Sampler* GPU_SamplerLinearMirror(void);
GPU_API GPU_Sampler* GPU_SamplerNearestClamp(void);
GPU_API GPU_Sampler* GPU_SamplerNearestWrap(void);
GPU_API GPU_Sampler* GPU_SamplerNearestMirror(void);
GPU_API GPU_Sampler* GPU_MakeSampler(const GPU_SamplerDesc* desc);                   /* [gpu.h:367] */
GPU_API void GPU_DestroySampler(GPU_Sampler* sampler);                               /* [gpu.h:368] */

/* ============================ pipeline layouts ============================ */
/* Bindings are numbered in declaration order (gpu_vulkan.c:649-656); the names are how the
 * built-in kernels find their arguments ("TEX_ENV_CUBE", "OUTPUT", "GLOBALS", "GBUFFER_*", ...). */
GPU_API GPU_PipelineLayout* GPU_InitPipelineLayout(void);                            /* [gpu.h:370] */
GPU_API GPU_Binding GPU_TextureBinding(GPU_PipelineLayout* layout, const char* name);          /* [gpu.h:371] */
GPU_API GPU_Binding GPU_SamplerBinding(GPU_PipelineLayout* layout, const char* name);          /* [gpu.h:372] */
GPU_API GPU_Binding GPU_BufferBinding(GPU_PipelineLayout* layout, const char* name);           /* [gpu.h:373] */
GPU_API GPU_Binding GPU_StorageImageBinding(GPU_PipelineLayout* layout, const char* name, GPU_Format image_format);   /* [gpu.h:374] */
GPU_API void GPU_FinalizePipelineLayout(GPU_PipelineLayout* layout);                 /* [gpu.h:375] */
GPU_API void GPU_DestroyPipelineLayout(GPU_PipelineLayout* layout);                  /* [gpu.h:376] */

/* ============================= descriptor sets ============================ */
GPU_API GPU_DescriptorArena* GPU_MakeDescriptorArena(void);                          /* [gpu.h:378] */
GPU_API void GPU_ResetDescriptorArena(GPU_DescriptorArena* descriptor_arena);        /* [gpu.h:379] */
GPU_API void GPU_DestroyDescriptorArena(GPU_DescriptorArena* descriptor_arena);      /* [gpu.h:380] NULL ok */
GPU_API GPU_DescriptorSet* GPU_InitDescriptorSet(GPU_DescriptorArena* descriptor_arena, GPU_PipelineLayout* pipeline_layout);   /* [gpu.h:384] */
GPU_API void GPU_SetTextureBinding(GPU_DescriptorSet* set, GPU_Binding binding, GPU_Texture* value);                            /* [gpu.h:387] */
GPU_API void GPU_SetTextureMipBinding(GPU_DescriptorSet* set, GPU_Binding binding, GPU_Texture* value, uint32_t mip_level);     /* [gpu.h:394] */
GPU_API void GPU_SetSamplerBinding(GPU_DescriptorSet* set, GPU_Binding binding, GPU_Sampler* value);                            /* [gpu.h:396] */
GPU_API void GPU_SetBufferBinding(GPU_DescriptorSet* set, GPU_Binding binding, GPU_Buffer* value);                              /* [gpu.h:398] */
GPU_API void GPU_SetStorageImageBinding(GPU_DescriptorSet* set, GPU_Binding binding, GPU_Texture* value, uint32_t mip_level);   /* [gpu.h:400] */
GPU_API void GPU_FinalizeDescriptorSet(GPU_DescriptorSet* set);                      /* [gpu.h:402] asserts every slot is set (gpu_vulkan.c:841,849) */
GPU_API void GPU_DestroyDescriptorSet(GPU_DescriptorSet* set);                       /* [gpu.h:406] NULL ok; arena sets die with the arena */

/* ================================ resources =============================== */
/* data != NULL: synchronous upload, and the mip chain is generated when HasMipmaps is set; cube
 * data = 6 faces tightly packed (+X,-X,+Y,-Y,+Z,-Z) [gpu.h:410-413, gpu_vulkan.c:1431-1453]. */
GPU_API GPU_Texture* GPU_MakeTexture(GPU_Format format, uint32_t width, uint32_t height, uint32_t depth, GPU_TextureFlags flags, const void* data);
GPU_API void GPU_DestroyTexture(GPU_Texture* texture);                               /* [gpu.h:416] NULL ok */
GPU_API GPU_Buffer* GPU_MakeBuffer(uint32_t size, GPU_BufferFlags flags, const void* data);    /* [gpu.h:419] */
GPU_API void GPU_DestroyBuffer(GPU_Buffer* buffer);                                  /* [gpu.h:422] NULL ok */

/* ================================ pipelines =============================== */
/* There is no GLSL compiler on this backend: the shader is matched to a built-in HIP kernel by the
 * basename of desc->glsl_debug_filepath (set by render.cpp:16/42-43), cross-checked against
 * signature strings in desc->glsl when given.  On success a non-empty opaque token is returned so
 * that caller logic (render.cpp:19-24) is unchanged; an unknown shader yields the empty string and
 * one entry in *out_errors (or an assert when out_errors is NULL) [gpu.h:424-428]. */
GPU_API GPU_String GPU_SPIRVFromGLSL(DS_Arena* arena, GPU_ShaderStage stage, GPU_PipelineLayout* pipeline_layout, const GPU_ShaderDesc* desc, GPU_GLSLErrorArray* out_errors);
GPU_API GPU_String GPU_JoinGLSLErrorString(DS_Arena* arena, GPU_GLSLErrorArray errors);

GPU_API GPU_RenderPass* GPU_MakeRenderPass(const GPU_RenderPassDesc* desc);          /* [gpu.h:431] */
GPU_API void GPU_DestroyRenderPass(GPU_RenderPass* render_pass);                     /* [gpu.h:432] */

static inline GPU_Access GPU_Read(uint32_t binding) { GPU_Access x = { GPU_AccessFlag_Read, binding }; return x; }                            /* [gpu.h:434] */
static inline GPU_Access GPU_Write(uint32_t binding) { GPU_Access x = { GPU_AccessFlag_Write, binding }; return x; }                          /* [gpu.h:435] */
static inline GPU_Access GPU_ReadWrite(uint32_t binding) { GPU_Access x = { GPU_AccessFlag_Read | GPU_AccessFlag_Write, binding }; return x; } /* [gpu.h:436] */

GPU_API GPU_GraphicsPipeline* GPU_MakeGraphicsPipeline(const GPU_GraphicsPipelineDesc* desc);  /* [gpu.h:438] lighting_pass.glsl only */
GPU_API void GPU_DestroyGraphicsPipeline(GPU_GraphicsPipeline* pipeline);            /* [gpu.h:441] NULL ok */
GPU_API GPU_ComputePipeline* GPU_MakeComputePipeline(GPU_PipelineLayout* layout, const GPU_ShaderDesc* cs);   /* [gpu.h:443] */
GPU_API void GPU_DestroyComputePipeline(GPU_ComputePipeline* pipeline);              /* [gpu.h:446] NULL ok */

/* ================================== graphs ================================ */
/* A graph = one HIP stream + a recorded op list.  Submit launches the ops asynchronously; Wait
 * blocks on the stream and resets the graph [gpu.h:450-453]. */
GPU_API GPU_Graph* GPU_MakeGraph(void);
GPU_API void GPU_GraphSubmit(GPU_Graph* graph);
GPU_API void GPU_GraphWait(GPU_Graph* graph);
GPU_API void GPU_DestroyGraph(GPU_Graph* graph);
GPU_API void GPU_MakeSwapchainGraphs(uint32_t count, GPU_Graph** out_graphs);        /* [gpu.h:456] plain graphs: headless */
GPU_API GPU_Texture* GPU_GetBackbuffer(GPU_Graph* graph);                            /* [gpu.h:460] always NULL: headless */

GPU_API void GPU_OpBindComputePipeline(GPU_Graph* graph, GPU_ComputePipeline* pipeline);       /* [gpu.h:467] */
GPU_API void GPU_OpBindComputeDescriptorSet(GPU_Graph* graph, GPU_DescriptorSet* set);         /* [gpu.h:470] */
/* data is copied at the call; at most 128 bytes (gpu_vulkan.c:710) [gpu.h:486-487] */
GPU_API void GPU_OpPushGraphicsConstants(GPU_Graph* graph, GPU_PipelineLayout* pipeline_layout, void* data, uint32_t size);
GPU_API void GPU_OpPushComputeConstants(GPU_Graph* graph, GPU_PipelineLayout* pipeline_layout, void* data, uint32_t size);
/* Group counts are in units of the shaders' 8x8x6 local size (render.cpp:532,578,610): the kernel
 * covers min(gx*8, image width) x min(gy*8, image height) texels of all faces [gpu.h:483]. */
GPU_API void GPU_OpDispatch(GPU_Graph* graph, uint32_t group_count_x, uint32_t group_count_y, uint32_t group_count_z);

/* Shade pass through the reference's raster vocabulary (render.cpp:1119-1127) [gpu.h:472-481]:
 * PrepareRenderPass -> PrepareDrawParams(pipeline, set) -> BeginRenderPass -> BindDrawParams ->
 * Draw(3,1,0,0) -> EndRenderPass launches the shade kernel over the pass's width x height. */
GPU_API void GPU_OpPrepareRenderPass(GPU_Graph* graph, GPU_RenderPass* render_pass);
GPU_API uint32_t GPU_OpPrepareDrawParams(GPU_Graph* graph, GPU_GraphicsPipeline* pipeline, GPU_DescriptorSet* descriptor_set);
GPU_API void GPU_OpBeginRenderPass(GPU_Graph* graph);
GPU_API void GPU_OpEndRenderPass(GPU_Graph* graph);
GPU_API void GPU_OpBindDrawParams(GPU_Graph* graph, uint32_t draw_params);
GPU_API void GPU_OpDraw(GPU_Graph* graph, uint32_t vertex_count, uint32_t instance_count, uint32_t first_vertex, uint32_t first_instance);
GPU_API void GPU_OpDrawIndexed(GPU_Graph* graph, uint32_t index_count, uint32_t instance_count, uint32_t first_index, uint32_t vertex_offset, uint32_t first_instance);   /* unsupported (raster) */
GPU_API void GPU_OpBindVertexBuffer(GPU_Graph* graph, GPU_Buffer* buffer);           /* [gpu.h:464] unsupported (raster) */
GPU_API void GPU_OpBindIndexBuffer(GPU_Graph* graph, GPU_Buffer* buffer);            /* [gpu.h:465] unsupported (raster) */

/* ================================ transfers =============================== */
/* Texture<->buffer copies are tightly packed, row major, layers consecutive (gpu_vulkan.c:2925-2932). */
GPU_API void GPU_OpCopyBufferToBuffer(GPU_Graph* graph, GPU_Buffer* src, GPU_Buffer* dst, uint32_t dst_offset, uint32_t src_offset, uint32_t size);   /* [gpu.h:491] */
GPU_API void GPU_OpCopyBufferToTexture(GPU_Graph* graph, GPU_Buffer* src, GPU_Texture* dst, uint32_t dst_first_layer, uint32_t dst_layer_count, uint32_t dst_mip_level);   /* [gpu.h:492] */
GPU_API void GPU_OpCopyTextureToBuffer(GPU_Graph* graph, GPU_Texture* src, GPU_Buffer* dst);   /* [gpu.h:493] mip 0, all layers (gpu_vulkan.c:2945-2951) */
GPU_API void GPU_OpBlit(GPU_Graph* graph, const GPU_OpBlitInfo* info);               /* [gpu.h:495] whole 2-D subresources: 1:1 copy, 2:1 box, any other size = linear resample (RGBA32F) */
GPU_API void GPU_OpGenerateMipmaps(GPU_Graph* graph, GPU_Texture* texture);          /* [gpu.h:496] */
GPU_API void GPU_OpClearColorF(GPU_Graph* graph, GPU_Texture* dst, uint32_t mip_level, float r, float g, float b, float a);   /* [gpu.h:502] */
GPU_API void GPU_OpClearColorI(GPU_Graph* graph, GPU_Texture* dst, uint32_t mip_level, uint32_t r, uint32_t g, uint32_t b, uint32_t a);   /* [gpu.h:503] */
GPU_API void GPU_OpClearDepthStencil(GPU_Graph* graph, GPU_Texture* dst, uint32_t mip_level);   /* [gpu.h:504] depth := 1.0 */

#endif /* GPU_HIP_H */
