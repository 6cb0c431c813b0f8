"""pbrhip -- thin ctypes binding of libgpu_hip.so (the GPU_* / GPUX_* / PBR_* / pbrk_* C ABI).

Used by tests/, bench.py and __graft_entry__.py.  There is no CPU fallback: importing is free, but
`lib()` raises if the HIP library has not been built, and every GPU call needs a real device.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_PKG))                 # vulkan-pbr-renderer_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.path.join(PKG_ROOT, "libgpu_hip.so")

# ---- enums (include/gpu_hip.h) ----
(Format_Invalid, Format_R8UN, Format_RG8UN, Format_RGBA8UN, Format_BGRA8UN, Format_R16F, Format_RG16F, Format_RGB16F,
 Format_RGBA16F, Format_R32F, Format_RG32F, Format_RGB32F, Format_RGBA32F, Format_R8I, Format_R16I, Format_RG16I,
 Format_RGBA16I, Format_R32I, Format_RG32I, Format_RGB32I, Format_RGBA32I, Format_R64I, Format_D16UN,
 Format_D32F_Or_X8D24UN) = range(24)
TextureFlag_StorageImage, TextureFlag_RenderTarget, TextureFlag_HasMipmaps, TextureFlag_Cubemap = 1, 2, 4, 8
BufferFlag_CPU, BufferFlag_GPU, BufferFlag_StorageBuffer = 1, 2, 4
Shade_IBL, Shade_LightShafts = 1, 2
Unit_Prefilter, Unit_Irradiance, Unit_BrdfLut = 0, 1, 2


class GPU_Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32), ("layer_count", C.c_uint32),
                ("mip_level_count", C.c_uint32), ("format", C.c_int), ("flags", C.c_int)]


class GPU_Buffer(C.Structure):
    _fields_ = [("flags", C.c_int), ("size", C.c_uint32), ("data", C.c_void_p)]


class GPU_String(C.Structure):
    _fields_ = [("data", C.c_char_p), ("length", C.c_size_t)]


class GPU_ShaderDesc(C.Structure):
    _fields_ = [("accesses", C.c_void_p), ("accesses_count", C.c_uint32), ("glsl_debug_filepath", GPU_String),
                ("glsl_includer", C.c_void_p), ("glsl_includer_ctx", C.c_void_p), ("spirv", GPU_String), ("glsl", GPU_String)]


class GPU_GLSLError(C.Structure):
    _fields_ = [("shader_stage", C.c_int), ("line", C.c_uint32), ("error_message", GPU_String)]


class GPU_GLSLErrorArray(C.Structure):
    _fields_ = [("data", C.POINTER(GPU_GLSLError)), ("length", C.c_uint32)]


class PBR_IBLMaps(C.Structure):
    _fields_ = [("irradiance_map", C.POINTER(GPU_Texture)), ("brdf_lut", C.POINTER(GPU_Texture)),
                ("tex_specular_env_map", C.POINTER(GPU_Texture))]


class PBR_WorkUnit(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mip", C.c_uint32), ("face0", C.c_uint32), ("face1", C.c_uint32),
                ("row0", C.c_uint32), ("row1", C.c_uint32), ("cost", C.c_double)]


class PBR_XferRange(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_uint64), ("peer", C.c_int)]


class PBR_Globals(C.Structure):
    _fields_ = [(n, C.c_float * 16) for n in ("clip_space_from_world", "clip_space_from_view", "world_space_from_clip",
                                              "view_space_from_clip", "view_space_from_world", "world_space_from_view",
                                              "sun_space_from_world", "old_clip_space_from_world")] + [
        ("sun_direction", C.c_float * 4), ("camera_pos", C.c_float * 3), ("frame_idx_mod_59", C.c_float),
        ("lightgrid_scale", C.c_float), ("visualize_lightgrid", C.c_uint32)]


assert C.sizeof(PBR_Globals) == 552


class PBR_GBuffer(C.Structure):
    _fields_ = [(n, C.POINTER(GPU_Texture)) for n in ("base_color", "normal", "orm", "emissive", "depth", "lighting_result")]


class GPUX_IBLConstants(C.Structure):
    _fields_ = [("mip_level", C.c_int32), ("roughness", C.c_float), ("src_lod", C.c_float), ("sample_count", C.c_int32)]


class PbrkTex2D(C.Structure):
    _fields_ = [("data", C.c_void_p), ("format", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class PbrkTaaArgs(C.Structure):
    _fields_ = [("lighting_result", PbrkTex2D), ("gbuffer_depth", PbrkTex2D), ("gbuffer_velocity", PbrkTex2D),
                ("gbuffer_velocity_prev", PbrkTex2D), ("prev_frame_result", PbrkTex2D), ("out", C.c_void_p),
                ("out_format", C.c_int), ("width", C.c_int), ("height", C.c_int), ("y0", C.c_int), ("y1", C.c_int)]


class PbrkFinalArgs(C.Structure):
    _fields_ = [("src", PbrkTex2D), ("out", C.c_void_p), ("out_format", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("y0", C.c_int), ("y1", C.c_int)]


Shade_IBL, Shade_LightShafts, Shade_SunShadows, Shade_VoxelGI = 1, 2, 4, 8
PBRK_FMT_RG16F, PBRK_FMT_RG32F, PBRK_FMT_RGBA16F, PBRK_FMT_RGBA32F, PBRK_FMT_R32F, PBRK_FMT_RGBA8UN, PBRK_FMT_BGRA8UN = 1, 2, 3, 4, 5, 6, 7


class PbrkShadeArgs(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("x0", C.c_int), ("x1", C.c_int), ("y0", C.c_int), ("y1", C.c_int),
                ("base_color", C.c_void_p), ("normal", C.c_void_p), ("orm", C.c_void_p), ("emissive", C.c_void_p),
                ("depth", C.c_void_p), ("irradiance_bordered", C.c_void_p), ("irradiance_size", C.c_int),
                ("prefiltered_bordered", C.c_void_p), ("prefiltered_size", C.c_int), ("prefiltered_levels", C.c_int),
                ("lut", C.c_void_p), ("lut_size", C.c_int), ("irradiance_cells", C.c_void_p), ("prefiltered_cells", C.c_void_p),
                ("prefiltered_cells_first", C.c_int), ("lut_cells", C.c_void_p),
                ("sun_depth", C.c_void_p), ("sun_depth_w", C.c_int), ("sun_depth_h", C.c_int),
                ("lightgrid", C.c_void_p), ("lightgrid_size", C.c_int), ("prev_frame", C.c_void_p * 8),
                ("prev_frame_w", C.c_int), ("prev_frame_h", C.c_int), ("prev_frame_levels", C.c_int),
                ("out", C.c_void_p), ("out_format", C.c_int), ("flags", C.c_int),
                ("globals", C.c_float * 138)]


TexP = C.POINTER(GPU_Texture)
BufP = C.POINTER(GPU_Buffer)
VP = C.c_void_p
U32 = C.c_uint32

# name -> (restype, argtypes); this table is also what tests use to check the exported symbol set
PROTOTYPES = {
    # --- GPU_* boundary (include/gpu_hip.h) ---
    "GPU_Init": (None, [VP]), "GPU_Deinit": (None, []), "GPU_WaitUntilIdle": (None, []),
    "GPU_SamplerLinearWrap": (VP, []), "GPU_SamplerLinearClamp": (VP, []), "GPU_SamplerLinearMirror": (VP, []),
    "GPU_SamplerNearestClamp": (VP, []), "GPU_SamplerNearestWrap": (VP, []), "GPU_SamplerNearestMirror": (VP, []),
    "GPU_MakeSampler": (VP, [VP]), "GPU_DestroySampler": (None, [VP]),
    "GPU_InitPipelineLayout": (VP, []), "GPU_TextureBinding": (U32, [VP, C.c_char_p]), "GPU_SamplerBinding": (U32, [VP, C.c_char_p]),
    "GPU_BufferBinding": (U32, [VP, C.c_char_p]), "GPU_StorageImageBinding": (U32, [VP, C.c_char_p, C.c_int]),
    "GPU_FinalizePipelineLayout": (None, [VP]), "GPU_DestroyPipelineLayout": (None, [VP]),
    "GPU_MakeDescriptorArena": (VP, []), "GPU_ResetDescriptorArena": (None, [VP]), "GPU_DestroyDescriptorArena": (None, [VP]),
    "GPU_InitDescriptorSet": (VP, [VP, VP]), "GPU_DestroyDescriptorSet": (None, [VP]),
    "GPU_SetTextureBinding": (None, [VP, U32, TexP]), "GPU_SetTextureMipBinding": (None, [VP, U32, TexP, U32]),
    "GPU_SetSamplerBinding": (None, [VP, U32, VP]), "GPU_SetBufferBinding": (None, [VP, U32, BufP]),
    "GPU_SetStorageImageBinding": (None, [VP, U32, TexP, U32]), "GPU_FinalizeDescriptorSet": (None, [VP]),
    "GPU_MakeTexture": (TexP, [C.c_int, U32, U32, U32, C.c_int, VP]), "GPU_DestroyTexture": (None, [TexP]),
    "GPU_MakeBuffer": (BufP, [U32, C.c_int, VP]), "GPU_DestroyBuffer": (None, [BufP]),
    "GPU_SPIRVFromGLSL": (GPU_String, [VP, C.c_int, VP, C.POINTER(GPU_ShaderDesc), C.POINTER(GPU_GLSLErrorArray)]),
    "GPU_JoinGLSLErrorString": (GPU_String, [VP, GPU_GLSLErrorArray]),
    "GPU_MakeRenderPass": (VP, [VP]), "GPU_DestroyRenderPass": (None, [VP]),
    "GPU_MakeGraphicsPipeline": (VP, [VP]), "GPU_DestroyGraphicsPipeline": (None, [VP]),
    "GPU_MakeComputePipeline": (VP, [VP, C.POINTER(GPU_ShaderDesc)]), "GPU_DestroyComputePipeline": (None, [VP]),
    "GPU_MakeGraph": (VP, []), "GPU_GraphSubmit": (None, [VP]), "GPU_GraphWait": (None, [VP]), "GPU_DestroyGraph": (None, [VP]),
    "GPU_MakeSwapchainGraphs": (None, [U32, C.POINTER(VP)]), "GPU_GetBackbuffer": (TexP, [VP]),
    "GPU_OpBindComputePipeline": (None, [VP, VP]), "GPU_OpBindComputeDescriptorSet": (None, [VP, VP]),
    "GPU_OpPushGraphicsConstants": (None, [VP, VP, VP, U32]), "GPU_OpPushComputeConstants": (None, [VP, VP, VP, U32]),
    "GPU_OpDispatch": (None, [VP, U32, U32, U32]),
    "GPU_OpPrepareRenderPass": (None, [VP, VP]), "GPU_OpPrepareDrawParams": (U32, [VP, VP, VP]),
    "GPU_OpBeginRenderPass": (None, [VP]), "GPU_OpEndRenderPass": (None, [VP]), "GPU_OpBindDrawParams": (None, [VP, U32]),
    "GPU_OpDraw": (None, [VP, U32, U32, U32, U32]), "GPU_OpDrawIndexed": (None, [VP, U32, U32, U32, U32, U32]),
    "GPU_OpBindVertexBuffer": (None, [VP, BufP]), "GPU_OpBindIndexBuffer": (None, [VP, BufP]),
    "GPU_OpCopyBufferToBuffer": (None, [VP, BufP, BufP, U32, U32, U32]),
    "GPU_OpCopyBufferToTexture": (None, [VP, BufP, TexP, U32, U32, U32]), "GPU_OpCopyTextureToBuffer": (None, [VP, TexP, BufP]),
    "GPU_OpBlit": (None, [VP, VP]), "GPU_OpGenerateMipmaps": (None, [VP, TexP]),
    "GPU_OpClearColorF": (None, [VP, TexP, U32, C.c_float, C.c_float, C.c_float, C.c_float]),
    "GPU_OpClearColorI": (None, [VP, TexP, U32, U32, U32, U32, U32]), "GPU_OpClearDepthStencil": (None, [VP, TexP, U32]),
    # --- GPUX_* extensions (include/gpux.h) ---
    "GPUX_GetFormatInfo": (None, None),
    "GPUX_SetDevice": (None, [C.c_int]), "GPUX_GetDevice": (C.c_int, []), "GPUX_SetErrorHandler": (None, [VP, VP]),
    "GPUX_BackendName": (C.c_char_p, []), "GPUX_OpDispatchRows": (None, [VP, U32, U32, U32, U32]),
    "GPUX_OpDispatchLines": (None, [VP, U32, U32, U32, U32]),
    "GPUX_SetShadeFlags": (None, [VP, C.c_int]), "GPUX_OpDrawRows": (None, [VP, U32, U32]),
    "GPUX_OpCopyTextureMipToBuffer": (None, [VP, TexP, U32, BufP, U32]), "GPUX_OpCopyBufferToTextureMip": (None, [VP, BufP, U32, TexP, U32]),
    "GPUX_FoldedBlitCount": (C.c_uint64, []), "GPUX_SetGraphOverlap": (None, [C.c_int]), "GPUX_OverlappedSubmitCount": (C.c_uint64, []), "GPUX_TextureMipBytes": (C.c_uint64, [TexP, U32]), "GPUX_TextureDevicePtr": (VP, [TexP, U32]), "GPUX_BufferDevicePtr": (VP, [BufP]),
    "GPUX_MakeTextureExternal": (TexP, [C.c_int, U32, U32, U32, C.c_int, VP, C.c_uint64]),
    "GPUX_InvalidateTexture": (None, [TexP]),
    "GPUX_TextureTotalBytes": (C.c_uint64, [TexP]), "GPUX_TextureMipOffset": (C.c_uint64, [TexP, U32]),
    "GPUX_MakeCubemapFromEquirect": (TexP, [VP, U32, U32, U32, C.c_int]),
    "GPUX_GraphStream": (VP, [VP]), "GPUX_EnableOpTiming": (None, [C.c_int]), "GPUX_SetTileStreams": (None, [C.c_int]), "GPUX_GraphTimedOpCount": (U32, [VP]),
    "GPUX_GraphTimedOpName": (C.c_char_p, [VP, U32]), "GPUX_GraphTimedOpMs": (C.c_float, [VP, U32]), "GPUX_GraphSpanMs": (C.c_float, [VP]),
    # --- host layer (include/pbr_host.h) ---
    "PBR_DecodeHDR": (VP, [VP, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "PBR_MakeTextureFromHDRIMemory": (TexP, [VP, C.c_size_t]), "PBR_MakeTextureFromHDRIFile": (TexP, [C.c_char_p]),
    "PBR_MakeTextureFromEquirectHDRIMemory": (TexP, [VP, C.c_size_t, U32]), "PBR_MakeTextureFromEquirectHDRIFile": (TexP, [C.c_char_p, U32]),
    "PBR_EncodeHDR": (VP, [VP, C.c_int, C.c_int, C.POINTER(C.c_size_t)]), "PBR_WriteHDRFile": (C.c_int, [C.c_char_p, VP, C.c_int, C.c_int]),
    "PBR_WriteCubeStripHDR": (C.c_int, [C.c_char_p, TexP, U32]),
    "PBR_MakeIBLMaps": (None, [C.POINTER(PBR_IBLMaps), U32, U32, U32]), "PBR_DestroyIBLMaps": (None, [C.POINTER(PBR_IBLMaps)]),
    "PBR_GenIrradianceMap": (None, [TexP, TexP]), "PBR_GenPrefilteredEnvMap": (None, [TexP, TexP, U32]),
    "PBR_GenBRDFIntegrationMap": (None, [TexP]),
    "PBR_MakeIBLPipelines": (VP, []), "PBR_DestroyIBLPipelines": (None, [VP]),
    "PBR_RecordUnits": (None, [VP, VP, VP, TexP, C.POINTER(PBR_IBLMaps), C.POINTER(PBR_WorkUnit), U32]),
    "PBR_PartitionIBL": (U32, [U32, U32, U32, U32, C.c_int, C.c_int, C.POINTER(PBR_WorkUnit), U32]),
    "PBR_ExchangeRanges": (C.c_int, [VP, VP, VP, U32, VP, U32]),
    "PBR_SetRcclLibrary": (C.c_int, [C.c_char_p]), "PBR_RcclInfo": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "PBR_CommInfo": (C.c_int, [VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "PBR_UnitByteRange": (C.c_int, [C.POINTER(PBR_IBLMaps), C.POINTER(PBR_WorkUnit), C.POINTER(TexP), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "PBR_GatherUnits": (C.c_int64, [VP, VP, C.c_int, C.c_int, C.c_int, C.POINTER(PBR_IBLMaps), U32, U32]),
    "GPUX_SetGraphReplay": (None, [C.c_int]),
    "GPUX_GraphReplayStats": (None, [VP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "PBR_GatherUnitsMasked": (C.c_int64, [VP, VP, C.c_int, C.c_int, C.c_int, C.POINTER(PBR_IBLMaps), U32, U32, U32]),
    "PBR_GatherPlan": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.POINTER(PBR_IBLMaps), U32, U32, U32, VP, U32]),
    "PBR_SelectUnits": (U32, [C.POINTER(PBR_WorkUnit), U32, U32, C.POINTER(PBR_WorkUnit)]),
    "PBR_RunPartitionedIBL": (C.c_int64, [VP, VP, VP, VP, TexP, C.POINTER(PBR_IBLMaps), VP, C.c_int, C.c_int, C.c_int, U32, U32]),
    "PBR_BandRows": (None, [U32, C.c_int, C.c_int, C.POINTER(U32), C.POINTER(U32)]),
    "PBR_GatherBands": (C.c_int64, [VP, VP, C.c_int, C.c_int, C.c_int, TexP]),
    "PBR_FillGlobals": (None, [C.POINTER(PBR_Globals), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_float, C.c_float, U32]),
    "PBR_MakeGBuffer": (None, [C.POINTER(PBR_GBuffer), U32, U32, C.c_int]), "PBR_DestroyGBuffer": (None, [C.POINTER(PBR_GBuffer)]),
    "PBR_MakeLightingPass": (VP, [C.POINTER(PBR_GBuffer), C.POINTER(PBR_IBLMaps), U32, U32]), "PBR_DestroyLightingPass": (None, [VP]),
    "PBR_MakeLightingPassEx": (VP, [C.POINTER(PBR_GBuffer), C.POINTER(PBR_IBLMaps), U32, U32, TexP]),
    "PBR_MakeLightingPassLive": (VP, [C.POINTER(PBR_GBuffer), C.POINTER(PBR_IBLMaps), U32, U32, TexP, TexP, TexP]),
    "PBR_LightingGlobalsBuffer": (BufP, [VP]), "PBR_LightingPipeline": (VP, [VP]),
    "PBR_RecordLightingPass": (None, [VP, VP, C.POINTER(PBR_Globals), U32, U32]),
    "PBR_MakePostProcess": (VP, [C.POINTER(PBR_GBuffer), U32, U32, C.c_int]), "PBR_DestroyPostProcess": (None, [VP]),
    "PBR_PostVelocity": (TexP, [VP, U32]), "PBR_PostTaaOutput": (TexP, [VP, U32]), "PBR_PostBackbuffer": (TexP, [VP]),
    "PBR_RecordBloom": (None, [VP, VP, U32]), "PBR_RecordFinalPostProcessBloom": (None, [VP, VP, U32]),
    "PBR_PostBloomDownscale": (TexP, [VP]), "PBR_PostBloomUpscale": (TexP, [VP]), "PBR_PostBloomPassCount": (U32, [VP]),
    "PBR_RecordTaaResolve": (None, [VP, VP, U32]), "PBR_RecordTaaResolveRows": (None, [VP, VP, U32, U32, U32]), "PBR_RecordFinalPostProcess": (None, [VP, VP, U32]),
    "PBR_MakeLightgrid": (VP, [U32]), "PBR_DestroyLightgrid": (None, [VP]), "PBR_LightgridTexture": (TexP, [VP]),
    "PBR_LightgridSweepDirection": (U32, [VP]), "PBR_RecordLightgridClear": (None, [VP, VP]),
    "PBR_RecordLightgridSweep": (None, [VP, VP]), "PBR_RecordLightgridSweepLines": (None, [VP, VP, U32, U32, U32, U32, U32]),
    # --- low-level kernel ABI (include/pbr_kernels.h) ---
    "pbrk_level_offset": (C.c_size_t, [C.c_int, C.c_int]), "pbrk_pyramid_texels": (C.c_size_t, [C.c_int, C.c_int]),
    "pbrk_bordered_level_offset": (C.c_size_t, [C.c_int, C.c_int]), "pbrk_bordered_pyramid_texels": (C.c_size_t, [C.c_int, C.c_int]),
    "pbrk_mip_count": (C.c_int, [C.c_int, C.c_int]),
    "pbrk_host_sample_angles": (None, [C.c_int, VP]), "pbrk_host_prefilter_table": (C.c_int, [C.c_int, C.c_float, VP, C.POINTER(C.c_float)]),
    "pbrk_host_irradiance_table": (C.c_int, [C.c_int, VP]),
    "pbrk_mip_chain": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "pbrk_level_minmax": (C.c_int, [VP, C.c_size_t, VP, VP]),
    "GPUX_SetPrefilterTolerance": (None, [C.c_float]), "GPUX_PrefilterKeptSamples": (C.c_int, [U32]),
    "pbrk_set_cube_sampler_snap": (None, [C.c_int]), "pbrk_get_cube_sampler_snap": (C.c_int, []), "pbrk_box_downsample": (C.c_int, [VP, C.c_int, VP, C.c_int, VP]),
    "pbrk_blit_linear": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, VP]),
    "pbrk_fill_pattern": (C.c_int, [VP, C.c_ulonglong, VP, C.c_int, VP]),
    "pbrk_bloom_set_thresholds": (None, [C.c_longlong, C.c_longlong]),
    "pbrk_mc_set_kernels": (None, [C.c_int, C.c_int]), "pbrk_shade_set_fast": (None, [C.c_int]),
    "pbrk_border_build": (C.c_int, [VP, VP, C.c_int, C.c_int, VP]),
    "pbrk_border_build_range": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "pbrk_debug_sample": (C.c_int, [C.c_int, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, VP, VP]),
    "pbrk_bloom_pass": (C.c_int, [VP, VP]),
    "pbrk_taa_resolve": (C.c_int, [VP, VP]), "pbrk_final_post_process": (C.c_int, [VP, VP]),
    "pbrk_lightgrid_sweep": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "pbrk_brdf_lut": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, VP]),
    "pbrk_prefilter_copy": (C.c_int, [VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "pbrk_mc_filter": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, C.c_float, C.c_float, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "pbrk_cells_bytes": (C.c_size_t, [C.c_int]), "pbrk_cells_build": (C.c_int, [VP, C.c_int, VP, VP]),
    "pbrk_shade": (C.c_int, [C.POINTER(PbrkShadeArgs), VP]),
    "pbrk_shade_set_tile_min_pixels": (None, [C.c_longlong]), "pbrk_shade_tables_ready": (C.c_int, [C.c_int, C.c_int]),
    "pbrk_shade_needs_tables": (C.c_int, [C.c_int, C.c_int]),
    "pbrk_mc_region_stats": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "pbrk_mc_region_flag_stats": (C.c_int, [C.POINTER(C.c_uint64)]), "pbrk_mc_region_window_stats": (C.c_int, [C.POINTER(C.c_uint64)]),
    "pbrk_lut_cells_build": (C.c_int, [VP, C.c_int, VP, VP]),
    "pbrk_equirect_to_cube": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, VP]),
}

_LIB = None


def lib():
    """Load libgpu_hip.so.  No fallback: a missing library is a hard error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(PKG_ROOT, 'csrc')}` "
                           "(or __graft_entry__.build()); there is no CPU fallback for the HIP path")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)
        if args is not None:
            fn.restype = res
            fn.argtypes = args
    _LIB = L
    return L


# ---- conveniences used by tests / bench ------------------------------------------------------
_FMT_NP = {Format_RGBA32F: (np.float32, 4), Format_RG32F: (np.float32, 2), Format_RG16F: (np.float16, 2),
           Format_RGBA16F: (np.float16, 4), Format_RGBA8UN: (np.uint8, 4), Format_BGRA8UN: (np.uint8, 4), Format_D32F_Or_X8D24UN: (np.float32, 1),
           Format_R32F: (np.float32, 1)}


def init(device=None):
    L = lib()
    if device is not None:
        L.GPUX_SetDevice(int(device))
    L.GPU_Init(None)
    return L


def make_texture(fmt, w, h, flags, data=None, depth=1):
    L = lib()
    ptr = None
    if data is not None:
        data = np.ascontiguousarray(data)
        ptr = data.ctypes.data_as(VP)
    t = L.GPU_MakeTexture(fmt, w, h, depth, flags, ptr)
    if not t:
        raise RuntimeError("GPU_MakeTexture failed")
    return t


def upload_mip(tex, mip, array):
    """Host array -> all layers of one mip (staging buffer + GPUX_OpCopyBufferToTextureMip)."""
    L = lib()
    array = np.ascontiguousarray(array)
    nbytes = L.GPUX_TextureMipBytes(tex, mip)
    assert array.nbytes == nbytes, (array.nbytes, nbytes)
    buf = L.GPU_MakeBuffer(nbytes, BufferFlag_CPU, array.ctypes.data_as(VP))
    g = L.GPU_MakeGraph()
    L.GPUX_OpCopyBufferToTextureMip(g, buf, 0, tex, mip)
    L.GPU_GraphSubmit(g)
    L.GPU_GraphWait(g)
    L.GPU_DestroyGraph(g)
    L.GPU_DestroyBuffer(buf)


def read_mip(tex, mip=0):
    """All layers of one mip -> numpy array [layers][h][w][c] (squeezed for single-layer textures; [d][h][w][c] for 3-D)."""
    L = lib()
    t = tex.contents
    nbytes = L.GPUX_TextureMipBytes(tex, mip)
    buf = L.GPU_MakeBuffer(nbytes, BufferFlag_CPU, None)
    g = L.GPU_MakeGraph()
    L.GPUX_OpCopyTextureMipToBuffer(g, tex, mip, buf, 0)
    L.GPU_GraphSubmit(g)
    L.GPU_GraphWait(g)
    dt, ch = _FMT_NP[t.format]
    w, h = max(1, t.width >> mip), max(1, t.height >> mip)
    raw = (C.c_char * nbytes).from_address(buf.contents.data)
    d = max(1, t.depth >> mip)
    arr = np.frombuffer(raw, dtype=dt).reshape(t.layer_count * d, h, w, ch).copy()
    L.GPU_DestroyGraph(g)
    L.GPU_DestroyBuffer(buf)
    return arr if (t.layer_count > 1 or d > 1) else arr[0]


def fill_globals(pos, ori=None, fov=75.0, aspect=16.0 / 9.0, near=0.02, far=1.0e4, sun_angle=(56.5, 97.0), frame_idx=0):
    g = PBR_Globals()
    p = (C.c_float * 3)(*[float(v) for v in pos])
    o = None if ori is None else (C.c_float * 4)(*[float(v) for v in ori])
    lib().PBR_FillGlobals(C.byref(g), p, o, fov, aspect, near, far, sun_angle[0], sun_angle[1], frame_idx)
    return g


def partition(specular_size, min_size, irradiance_size, env_size, world, rank):
    L = lib()
    n = L.PBR_PartitionIBL(specular_size, min_size, irradiance_size, env_size, world, rank, None, 0)
    arr = (PBR_WorkUnit * max(1, n))()
    m = L.PBR_PartitionIBL(specular_size, min_size, irradiance_size, env_size, world, rank, arr, n)
    assert m == n
    return arr, n


# ---- RCCL bootstrap for tests / bench.py (the library itself never creates a communicator: include/pbr_host.h) ----
class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_RCCL = None


def rccl_info():
    """(ncclGetVersion code, path of the librccl the C host layer bound) -- PBR_RcclInfo; raises if none can be loaded."""
    ver, path = C.c_int(0), C.c_char_p()
    if lib().PBR_RcclInfo(C.byref(ver), C.byref(path)) != 0:
        raise RuntimeError("no librccl.so.1 could be bound (PBR_RCCL_LIB / PBR_SetRcclLibrary select one explicitly)")
    return int(ver.value), (path.value or b"").decode()


def rccl():
    """The SAME librccl the C host layer calls (PBR_RcclInfo names the file; one copy per process by SONAME), so that a
    communicator made here is valid for PBR_GatherUnits / PBR_GatherBands / PBR_ExchangeRanges."""
    global _RCCL
    if _RCCL is None:
        _, path = rccl_info()
        R = C.CDLL(path or "librccl.so.1")
        R.ncclGetUniqueId.argtypes = [C.POINTER(NcclUniqueId)]; R.ncclGetUniqueId.restype = C.c_int
        R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]; R.ncclCommInitRank.restype = C.c_int
        R.ncclCommDestroy.argtypes = [C.c_void_p]; R.ncclCommDestroy.restype = C.c_int
        R.ncclGetErrorString.argtypes = [C.c_int]; R.ncclGetErrorString.restype = C.c_char_p
        _RCCL = R
    return _RCCL


def comm_info(comm):
    """(rank count, this rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
    n, r = C.c_int(-1), C.c_int(-1)
    if lib().PBR_CommInfo(comm, C.byref(n), C.byref(r)) != 0:
        raise RuntimeError("PBR_CommInfo failed")
    return int(n.value), int(r.value)


def rccl_unique_id():
    uid = NcclUniqueId()
    rc = rccl().ncclGetUniqueId(C.byref(uid))
    if rc != 0:
        raise RuntimeError("ncclGetUniqueId: " + rccl().ncclGetErrorString(rc).decode())
    return C.string_at(C.byref(uid), 128)              # (a c_char array read as bytes stops at the first NUL)


def rccl_comm_init(world, rank, unique_id: bytes):
    """ncclCommInitRank -> opaque communicator handle (c_void_p) for PBR_GatherUnits / PBR_GatherBands / PBR_ExchangeRanges."""
    uid = NcclUniqueId()
    C.memmove(C.byref(uid), unique_id, 128)
    comm = C.c_void_p()
    rc = rccl().ncclCommInitRank(C.byref(comm), int(world), uid, int(rank))
    if rc != 0:
        raise RuntimeError("ncclCommInitRank: " + rccl().ncclGetErrorString(rc).decode())
    return comm


def rccl_comm_destroy(comm):
    rccl().ncclCommDestroy(comm)
