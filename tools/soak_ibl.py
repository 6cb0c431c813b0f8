#!/usr/bin/env python3
"""Soak of the IBL precompute (K2, K4a, K4b, K3) against the oracle over several procedural HDR environments (different seeds,
sun positions at 5e4:1 contrast) and output sizes:   python3 tools/soak_ibl.py [runs]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pbrhip  # noqa: E402
import pbr_oracle as O  # noqa: E402
from pbrhip import synth  # noqa: E402


def rel(a, b, floor=1e-3):
    return float((np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), floor)).max())


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    L = pbrhip.init()
    rng = np.random.default_rng(0x1B1)
    worst = 0.0
    for run in range(runs):
        W = int(rng.choice([32, 64, 128]))
        out = int(rng.choice([16, 32, 64]))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        synth.SUN_DIR = d                                            # a new sun position per run
        env = synth.synth_env(W, seed=int(rng.integers(1, 2 ** 31)))
        pyr = O.build_pyramid(env)
        env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
        maps = pbrhip.PBR_IBLMaps()
        L.PBR_MakeIBLMaps(C.byref(maps), 16, 64, out)
        L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 1)
        L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
        errs = []
        for m in range(maps.tex_specular_env_map.contents.mip_level_count):
            errs.append(rel(pbrhip.read_mip(maps.tex_specular_env_map, m), O.prefilter_mip(pyr, W, out, m)))
        e_irr = rel(pbrhip.read_mip(maps.irradiance_map, 0)[..., :3], O.irradiance(pyr, W, 16)[..., :3])
        nlev = env_tex.contents.mip_level_count
        mip_exact = all(np.array_equal(pbrhip.read_mip(env_tex, l), O.pyramid_level(pyr, W, l)) for l in range(nlev))
        worst = max(worst, max(errs), e_irr)
        print(f"run {run}: env {W}^2 -> specular {out}^2: max rel per mip {[f'{e:.1e}' for e in errs]} irradiance {e_irr:.1e} mip chain bit-exact {mip_exact}", flush=True)
        assert mip_exact
        L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(env_tex)
    # the sizes the region kernel serves (source level >= 16^2, output level >= 256^2): random rows of every such level vs the oracle
    for run in range(max(1, runs // 2)):
        W, out = [(256, 512), (512, 1024), (1024, 1024), (2048, 512), (512, 512)][run % 5]
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        synth.SUN_DIR = d
        env = synth.synth_env(W, seed=int(rng.integers(1, 2 ** 31)))
        pyr = O.build_pyramid(env)
        env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
        maps = pbrhip.PBR_IBLMaps()
        L.PBR_MakeIBLMaps(C.byref(maps), 16, 64, out)
        L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 256)
        errs = {}
        for m in range(maps.tex_specular_env_map.contents.mip_level_count):
            size = out >> m
            if size < 256:
                break
            got = pbrhip.read_mip(maps.tex_specular_env_map, m)
            for _ in range(3):
                f, y = int(rng.integers(0, 6)), int(rng.integers(0, size))
                want = O.prefilter_mip(pyr, W, out, m, faces=(f, f + 1), rows=(y, y + 1))[f, y]
                errs[m] = max(errs.get(m, 0.0), rel(got[f, y], want))
        worst = max(worst, max(errs.values()))
        print(f"region sizes, run {run}: env {W}^2 -> specular {out}^2 sun {np.round(d, 2)}: max rel over 3 random rows per mip {({m: f'{e:.1e}' for m, e in errs.items()})}", flush=True)
        L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(env_tex)
    st = (C.c_uint64 * 2)()
    if L.pbrk_mc_region_stats(st, 0) == 0:
        print(f"region kernel self-check: {st[0]} of {st[1]} wave-slices recomputed with direct loads")
        if st[0]:
            return 1
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
