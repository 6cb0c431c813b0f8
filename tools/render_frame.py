#!/usr/bin/env python3
"""Renders one viewable frame through the whole path (synthetic HDR cube -> IBL precompute -> metal-rough spheres G-buffer ->
shade -> TAA x3 -> bloom -> tone map) and writes it as a PNG:   python3 tools/render_frame.py out.png [width height [live]]
With `live` the lighting pass is the reference's complete live shader (shafts, sun shadows, voxel GI) inside the reference's
frame loop: light-grid sweep -> lighting (reading last frame's bloom_downscale_rt) -> TAA -> bloom -> final, eight frames."""
import ctypes as C
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python"))
import pbrhip  # noqa: E402
from pbrhip import synth  # noqa: E402


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    out = sys.argv[1]
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1280, 720)
    env = synth.synth_env(512, seed=0x5EED0001, workers=6)       # forked workers: before the HIP runtime is initialised
    L = pbrhip.init()
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 512, 512, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
    L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16)
    L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    live = len(sys.argv) > 4 and sys.argv[4] == "live"
    if live:
        gbd, grid, _, sun = synth.synth_gi_scene(W, H)
    else:
        gbd = synth.synth_gbuffer_spheres(W, H)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    for name, arr in (("base_color", gbd["base"]), ("normal", gbd["normal"]), ("orm", gbd["orm"]), ("emissive", gbd["emissive"]), ("depth", gbd["depth"])):
        pbrhip.upload_mip(getattr(gb, name), 0, arr)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_RGBA8UN)
    lg = None
    if live:
        lg = L.PBR_MakeLightgrid(grid.shape[0])
        pbrhip.upload_mip(L.PBR_LightgridTexture(lg), 0, grid)
        sun_tex = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, sun.shape[1], sun.shape[0], pbrhip.TextureFlag_RenderTarget)
        pbrhip.upload_mip(sun_tex, 0, sun)
        lp = L.PBR_MakeLightingPassLive(C.byref(gb), C.byref(maps), W, H, sun_tex, L.PBR_LightgridTexture(lg), L.PBR_PostBloomDownscale(pp))
        L.GPUX_SetShadeFlags(L.PBR_LightingPipeline(lp), pbrhip.Shade_LightShafts | pbrhip.Shade_SunShadows | pbrhip.Shade_VoxelGI)
    else:
        lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    g = L.GPU_MakeGraph()
    for frame in range(8 if live else 3):
        glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H, frame_idx=frame)
        if live:
            glob.lightgrid_scale = 1.0 / synth.GI_SCENE_EXTENT
            L.PBR_RecordLightgridSweep(lg, g)                                   # render.cpp:1061-1072
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
        L.PBR_RecordTaaResolve(pp, g, frame); L.PBR_RecordBloom(pp, g, frame); L.PBR_RecordFinalPostProcessBloom(pp, g, frame)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
    write_png(out, np.ascontiguousarray(bb[..., :3]))
    print("wrote", out, bb.shape, "mean", bb[..., :3].mean())
    L.GPU_WaitUntilIdle()


if __name__ == "__main__":
    main()
