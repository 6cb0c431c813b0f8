#!/usr/bin/env python3
"""Soak test of the complete live lighting shader (K5 with shafts + sun shadows + voxel GI) against the oracle over many
cameras / frames / sun angles / seeds: counts pixels beyond 1e-4 (a branch flip along a ray would show up as a large error).
   python3 tools/soak_live.py [runs] [width height]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pbrhip  # noqa: E402
import pbr_oracle as O  # noqa: E402
from pbrhip import synth  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (192, 108)
    L = pbrhip.init()
    env = synth.synth_env(64, seed=0x5EED00AA)
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 16, 64, 32)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map); L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 1); L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    irr = pbrhip.read_mip(maps.irradiance_map, 0)
    nm = maps.tex_specular_env_map.contents.mip_level_count
    pyr = np.concatenate([pbrhip.read_mip(maps.tex_specular_env_map, m).ravel() for m in range(nm)])
    lut = pbrhip.read_mip(maps.brdf_lut, 0).view(np.uint16)
    rng = np.random.default_rng(0x50AC)
    total_bad = total_px = 0
    worst = 0.0
    for run in range(runs):
        cam = (float(rng.uniform(-2, 2)), float(rng.uniform(-7.5, -4.5)), float(rng.uniform(-2, 2)))
        synth.GI_SCENE_CAMERA = cam                                   # synth_gi_scene reads the module-level camera
        gbd, grid, levels, sun = synth.synth_gi_scene(W, H, seed=0x5EED0100 + run)
        if run % 2:                                                   # random materials on the same geometry
            for key in ("base", "orm"):
                gbd[key] = np.where(gbd["depth"][..., None] < 1, rng.integers(0, 256, gbd[key].shape, dtype=np.uint8), gbd[key])
        gb = pbrhip.PBR_GBuffer()
        L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA32F)
        for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
            pbrhip.upload_mip(getattr(gb, name), 0, gbd[key])
        n = grid.shape[0]
        grid_tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, n, n, pbrhip.TextureFlag_StorageImage, depth=n)
        pbrhip.upload_mip(grid_tex, 0, grid)
        prev_tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, levels[0].shape[1], levels[0].shape[0], pbrhip.TextureFlag_RenderTarget | pbrhip.TextureFlag_HasMipmaps)
        nlev = min(prev_tex.contents.mip_level_count, len(levels))
        for m in range(nlev):
            pbrhip.upload_mip(prev_tex, m, levels[m])
        sun_tex = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, sun.shape[1], sun.shape[0], pbrhip.TextureFlag_RenderTarget)
        pbrhip.upload_mip(sun_tex, 0, sun)
        lp = L.PBR_MakeLightingPassLive(C.byref(gb), C.byref(maps), W, H, sun_tex, grid_tex, prev_tex)
        L.GPUX_SetShadeFlags(L.PBR_LightingPipeline(lp), pbrhip.Shade_LightShafts | pbrhip.Shade_SunShadows | pbrhip.Shade_VoxelGI)
        glob = pbrhip.fill_globals(cam, aspect=W / H, frame_idx=int(rng.integers(0, 59)),
                                   sun_angle=(float(rng.uniform(20, 80)), float(rng.uniform(0, 360))))
        glob.lightgrid_scale = 1.0 / synth.GI_SCENE_EXTENT
        g = L.GPU_MakeGraph()
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
        got = pbrhip.read_mip(gb.lighting_result, 0)
        og = O.OrcGlobals.from_buffer_copy(bytes(glob))
        O.gi_exit_counts()
        want = O.shade(og, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"], flags=O.SHADE_SHAFTS | O.SHADE_SHADOWS | O.SHADE_GI,
                       irradiance_cube=irr, prefiltered_pyr=pyr, prefiltered_size=maps.tex_specular_env_map.contents.width, lut_half=lut,
                       sun_depth_map=sun, lightgrid=grid, prev_frame_levels=levels[:nlev])
        exits = O.gi_exit_counts()
        err = np.abs(got[..., :3].astype(np.float64) - want[..., :3]) / np.maximum(np.abs(want[..., :3]), 1e-2)
        bad = int((err.max(-1) >= 1e-4).sum())
        total_bad += bad; total_px += W * H; worst = max(worst, float(err.max()))
        print(f"run {run}: cam {tuple(round(c, 2) for c in cam)} surface {float((gbd['depth'] < 1).mean()):.2f} exits {exits} beyond 1e-4: {bad} max rel {err.max():.2e}", flush=True)
        L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb))
        for t in (grid_tex, prev_tex, sun_tex):
            L.GPU_DestroyTexture(t)
    print(f"TOTAL pixels {total_px} beyond 1e-4: {total_bad} worst {worst:.3e}")
    return 0 if total_bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
