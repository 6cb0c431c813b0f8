#!/usr/bin/env python3
"""Where the live lighting shader's time goes: K5 on the 1920x1080 GI scene under each combination of its optional blocks
(light shafts, sun shadows, voxel GI), IBL mode as the base line.   python3 tools/live_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python"))
from pbrhip import synth  # noqa: E402

W, H = 1920, 1080
gbd, grid, levels, sun = synth.synth_gi_scene(W, H)
import pbrhip  # noqa: E402
if os.environ.get("PBRHIP_LIB"):
    pbrhip.LIB_PATH = os.environ["PBRHIP_LIB"]

L = pbrhip.init(0)
L.GPUX_EnableOpTiming(1)
env = synth.synth_env(64, seed=0x5EED00AA)
env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
maps = pbrhip.PBR_IBLMaps()
L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)
L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map); L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16); L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
gb = pbrhip.PBR_GBuffer()
L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
    pbrhip.upload_mip(getattr(gb, name), 0, gbd[key])
n = grid.shape[0]
grid_tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, n, n, pbrhip.TextureFlag_StorageImage, depth=n)
pbrhip.upload_mip(grid_tex, 0, grid)
prev_tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, levels[0].shape[1], levels[0].shape[0], pbrhip.TextureFlag_RenderTarget | pbrhip.TextureFlag_HasMipmaps)
for m in range(min(prev_tex.contents.mip_level_count, len(levels))):
    pbrhip.upload_mip(prev_tex, m, levels[m])
sun_tex = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, sun.shape[1], sun.shape[0], pbrhip.TextureFlag_RenderTarget)
pbrhip.upload_mip(sun_tex, 0, sun)
lp = L.PBR_MakeLightingPassLive(C.byref(gb), C.byref(maps), W, H, sun_tex, grid_tex, prev_tex)
glob = pbrhip.fill_globals(synth.GI_SCENE_CAMERA, aspect=W / H, frame_idx=3)
glob.lightgrid_scale = 1.0 / synth.GI_SCENE_EXTENT
g = L.GPU_MakeGraph()
S, D, G, I = pbrhip.Shade_LightShafts, pbrhip.Shade_SunShadows, pbrhip.Shade_VoxelGI, pbrhip.Shade_IBL
print(f"surface pixels: {float((gbd['depth'] < 1).mean()):.3f}")
for name, flags in (("IBL", I), ("shafts", S), ("shadows", D), ("shafts+shadows", S | D), ("GI", G), ("GI+shadows", G | D), ("shafts+shadows+GI (live)", S | D | G)):
    L.GPUX_SetShadeFlags(L.PBR_LightingPipeline(lp), flags)
    for _ in range(7):
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    ms = [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g)) if L.GPUX_GraphTimedOpName(g, i).decode() == "K5.shade"]
    print(f"{name:28s} {np.median(ms[1:]) * 1e3:8.1f} us", flush=True)
L.GPU_WaitUntilIdle(); L.GPU_Deinit()
