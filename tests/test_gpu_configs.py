"""BASELINE.json's configurations at their FULL sizes against the CPU oracle (spot rows: the oracle needs ~1 s per row of
the big Monte-Carlo levels) -- C3 1920x1080 spheres, C4 4096^2 + 128^2 from a textured 2048^2 environment, C5 7680x4320 temple.
Shaders under test: gen_prefiltered_env_map.glsl:103-152, gen_irradiance_map.glsl:73-102, lighting_pass.glsl:432-716 (IBL mode).
Inputs come from tests/conftest.py (built before the HIP runtime starts)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4          # BASELINE.json north_star: 1e-4 relative per texel


def _rel(got, want, floor):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    return float((np.abs(got - want) / np.maximum(np.abs(want), floor)).max())


def test_c4_textured_environment_oracle_rows(gpu, c4_env):
    """C4 (4096^2 specular, 13 mips, + 128^2 irradiance from the seeded 2048^2 HDR cube): rows of mips 0, 1, 2, 3, 12 and of
    the irradiance map equal the oracle to 1e-4 -- the sizes at which K4a reads a 1024^2 level and K4b runs its region kernel
    (quarter-face regions at mip 1, whole faces at mips 2 and 3); a constant environment cannot see a tap-addressing error."""
    import pbrhip, pbr_oracle as O
    L = gpu
    W, S = 2048, 4096
    env = c4_env
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    spec = pbrhip.make_texture(pbrhip.Format_RGBA32F, S, S, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps | pbrhip.TextureFlag_StorageImage)
    L.PBR_GenPrefilteredEnvMap(tex, spec, 1)
    irr = pbrhip.make_texture(pbrhip.Format_RGBA32F, 128, 128, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage)
    L.PBR_GenIrradianceMap(tex, irr)
    pyr = O.build_pyramid(env)
    worst = {}
    for mip in (0, 1, 2, 3, 12):
        size = S >> mip
        got = pbrhip.read_mip(spec, mip)
        assert np.isfinite(got).all()
        for (f, y) in ((0, 0), (3, size // 2), (5, size - 1)) if size > 1 else ((0, 0), (3, 0), (5, 0)):
            want = O.prefilter_mip(pyr, W, S, mip, faces=(f, f + 1), rows=(y, y + 1))[f, y]
            e = _rel(got[f, y], want, 1e-3)
            worst[mip] = max(worst.get(mip, 0.0), e)
            assert e < REL, (mip, f, y, e)
        del got
    got = pbrhip.read_mip(irr, 0)
    for (f, y) in ((1, 0), (2, 64), (4, 127)):
        want = O.irradiance(pyr, W, 128, faces=(f, f + 1), rows=(y, y + 1))[f, y]
        e = _rel(got[f, y, :, :3], want[:, :3], 1e-3)
        assert e < REL and np.all(got[f, y, :, 3] == 0), (f, y, e)
    L.GPU_DestroyTexture(irr); L.GPU_DestroyTexture(spec); L.GPU_DestroyTexture(tex)


def test_region_kernel_ragged_row_shards_equal_full(gpu, c2_env, c4_env):
    """K4b's region kernel (taps from LDS-staged regions; levels with a source of >= 16^2 and an output of >= 256^2) in all its
    shapes -- 18^2 / 34^2 / 66^2 whole-face regions and 66^2 quarter-face regions -- dispatched as ragged row shards (rows that
    are no multiple of the 16-row tile, three shards per face) reproduces the whole-level dispatch bit for bit, and no wave had to
    fall back to direct loads (its completeness self-check).  Oracle parity of the same levels: test_c4_* / test_c2_*."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    st = (C.c_uint64 * 2)()
    assert L.pbrk_mc_region_stats(st, 1) in (0, -1)                     # reset (available once the kernel has run with PBR_MC_STATS=1)
    env256 = synth.synth_env(256, seed=0x5EED00AD)
    cases = ((env256, 256, 512, (1,)),            # mip 1: 256^2 from a 16^2 level  -> 18^2 regions
             (c2_env, 1024, 1024, (1, 2)),        # mip 1: 512^2 from 64^2 -> 66^2 whole-face; mip 2: 256^2 from 32^2 -> 34^2
             (c4_env, 2048, 512, (1,)))           # mip 1: 256^2 from 128^2 -> 66^2 quarter-face regions
    pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena(); g = L.GPU_MakeGraph()
    for env, W, S, mips in cases:
        tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
        maps = pbrhip.PBR_IBLMaps()
        L.PBR_MakeIBLMaps(C.byref(maps), 8, 64, S)
        spec = maps.tex_specular_env_map
        L.PBR_GenPrefilteredEnvMap(tex, spec, 256)
        full = {m: pbrhip.read_mip(spec, m).copy() for m in mips}
        for m in mips:
            L.GPU_OpClearColorF(g, spec, m, 0.0, 0.0, 0.0, 0.0)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
        units = []
        for m in mips:
            size = S >> m
            for f in range(6):
                cuts = (0, 5 + f, 131 - 2 * f, size)
                units += [pbrhip.PBR_WorkUnit(pbrhip.Unit_Prefilter, m, f, f + 1, cuts[k], cuts[k + 1], 0.0) for k in range(3)]
        arr = (pbrhip.PBR_WorkUnit * len(units))(*units)
        L.PBR_RecordUnits(pipes, g, arena, tex, C.byref(maps), arr, len(units))
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_ResetDescriptorArena(arena)
        for m in mips:
            got = pbrhip.read_mip(spec, m)
            assert np.array_equal(got.view(np.uint32), full[m].view(np.uint32)), (W, S, m)
            assert float(np.abs(got[..., :3]).max()) > 0.0
        L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(tex)
    assert L.pbrk_mc_region_stats(st, 0) == 0
    assert st[1] > 0 and st[0] == 0, (int(st[0]), int(st[1]))          # the region kernel ran; nothing had to be recomputed
    L.GPU_DestroyGraph(g); L.GPU_DestroyDescriptorArena(arena); L.PBR_DestroyIBLPipelines(pipes)


def _ibl_maps(L, pbrhip, env64):
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env64)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)                  # render.cpp:794-796
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
    L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16)
    L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    return env_tex, maps


def _oracle_rows(L, pbrhip, O, gbd, maps, glob, rows):
    irr = pbrhip.read_mip(maps.irradiance_map, 0)
    n = maps.tex_specular_env_map.contents.mip_level_count
    size = maps.tex_specular_env_map.contents.width
    pyr = np.concatenate([pbrhip.read_mip(maps.tex_specular_env_map, m).ravel() for m in range(n)])
    lut = pbrhip.read_mip(maps.brdf_lut, 0).view(np.uint16)
    g = O.OrcGlobals.from_buffer_copy(bytes(glob))
    H, W = gbd["depth"].shape
    out = {}
    for y in rows:
        full = O.shade(g, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"], flags=O.SHADE_IBL,
                       irradiance_cube=irr, prefiltered_pyr=pyr, prefiltered_size=size, lut_half=lut, region=(0, W, y, y + 1))
        out[y] = full[y].copy()
        del full
    return out


def _shade_full_size(L, gbd, bands, oracle_rows, expect_sky, expect_overflow=False):
    """Upload a full-size G-buffer, shade it into the reference's RGBA16F target in one draw and again as `bands` row bands
    (bit-identical), then shade the oracle rows into an RGBA32F target and compare at 1e-4."""
    import pbrhip, pbr_oracle as O
    from pbrhip import synth
    H, W = gbd["depth"].shape
    env_tex, maps = _ibl_maps(L, pbrhip, synth.synth_env(64, seed=0x5EED00AA))
    rt = pbrhip.TextureFlag_RenderTarget
    gb = pbrhip.PBR_GBuffer()
    gb.base_color = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt); gb.normal = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt)
    gb.orm = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt); gb.emissive = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt)
    gb.depth = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, W, H, rt)
    gb.lighting_result = pbrhip.make_texture(pbrhip.Format_RGBA16F, W, H, rt)                # render.cpp:693
    for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, name), 0, gbd[key])
    gb32 = pbrhip.PBR_GBuffer(gb.base_color, gb.normal, gb.orm, gb.emissive, gb.depth, pbrhip.make_texture(pbrhip.Format_RGBA32F, W, H, rt))
    lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    lp32 = L.PBR_MakeLightingPass(C.byref(gb32), C.byref(maps), W, H)
    glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H)
    g = L.GPU_MakeGraph()
    L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)            # the reference's draw: render.cpp:1117-1127
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    full = pbrhip.read_mip(gb.lighting_result, 0).view(np.uint16)
    L.GPU_OpClearColorF(g, gb.lighting_result, 0, 0.0, 0.0, 0.0, 0.0)
    step = H // bands
    for b in range(bands):                                          # the 8-GPU screen split, on one GPU
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), b * step, H if b == bands - 1 else (b + 1) * step)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    banded = pbrhip.read_mip(gb.lighting_result, 0).view(np.uint16)
    assert np.array_equal(full, banded)
    del banded
    f32 = full.view(np.float16)
    assert not np.isnan(f32).any() and (f32[..., 3] == 1).all() and (f32[..., :3] >= 0).all()
    # EVERY pixel: the fast instantiation (k_shade_fast: buffer loads from the cells twins, shared-reciprocal chain) against the general
    # kernel (k_shade: the shader statement by statement) on the whole frame -- two implementations of the same draw, each within 1e-4
    # of the oracle on its rows, so within 2e-4 of each other wherever they are compared
    L.PBR_RecordLightingPass(lp32, g, C.byref(glob), 0, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    fast32 = pbrhip.read_mip(gb32.lighting_result, 0)[..., :3].copy()
    try:
        L.pbrk_shade_set_fast(0)
        L.PBR_RecordLightingPass(lp32, g, C.byref(glob), 0, 0)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    finally:
        L.pbrk_shade_set_fast(-1)
    gen32 = pbrhip.read_mip(gb32.lighting_result, 0)[..., :3]
    worst = 0.0
    for y0 in range(0, H, 270):                                     # in slabs: the float64 temporaries of an 8K frame would be 2.4 GB
        a, b = fast32[y0:y0 + 270].astype(np.float64), gen32[y0:y0 + 270].astype(np.float64)
        worst = max(worst, float((np.abs(a - b) / np.maximum(np.abs(b), 1e-2)).max()))
    assert worst < 2 * REL, worst
    del fast32, gen32
    for y in oracle_rows:
        L.PBR_RecordLightingPass(lp32, g, C.byref(glob), y, y + 1)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    got = pbrhip.read_mip(gb32.lighting_result, 0)
    want = _oracle_rows(L, pbrhip, O, gbd, maps, glob, oracle_rows)
    n_surface = n_sky = n_overflow = 0
    for y in oracle_rows:
        sky = gbd["depth"][y] == 1.0
        n_sky += int(sky.sum()); n_surface += int((~sky).sum())
        e = _rel(got[y, :, :3], want[y][:, :3], 1e-2)
        assert e < REL, (y, e)
        # the RGBA16F frame holds the same values converted as DESIGN.md 7 defines the store: IEEE round-to-nearest-even, values
        # from 65520 on become +inf (0x7C00; lighting_pass.glsl:712 clamps only from below).  Compared as bit patterns, which are
        # monotone for non-negative halves (0x7BFF = 65504 is followed by 0x7C00 = inf): equal or one ulp off everywhere, and
        # exactly inf wherever the oracle's fp32 value is beyond the rounding boundary by more than the 1e-4 tolerance.
        hb = full[y, :, :3].astype(np.int32)
        with np.errstate(over="ignore"):
            wb = want[y][:, :3].astype(np.float32).astype(np.float16).view(np.uint16).astype(np.int32)
        assert np.abs(hb - wb).max() <= 1, (y, int(np.abs(hb - wb).max()))
        over = want[y][:, :3] >= 65520.0 * (1.0 + 2e-4)
        n_overflow += int(over.sum())
        assert (hb[over] == 0x7C00).all(), y
    assert n_surface > 0 and (n_sky > 0) == expect_sky
    if expect_overflow:
        assert n_overflow > 0, "the rows were chosen to contain sun highlights beyond the half range"
    L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyLightingPass(lp32)
    L.GPU_DestroyTexture(gb32.lighting_result); L.PBR_DestroyGBuffer(C.byref(gb))
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(env_tex)


def test_c4_every_texel_region_kernel_against_direct_kernel(gpu, c4_env):
    """C4 at full size, EVERY texel of the Monte-Carlo levels 1-5 (33.5 M texels; the oracle covers three rows per level): the
    shipped kernels (LDS-staged regions for mips 1-4, level-in-LDS below) against the round-1 direct kernel -- different code
    for face selection, tap addressing and staging, the same taps and weights -- agree to the order of their fp32 sums."""
    import pbrhip
    L = gpu
    W, S = 2048, 4096
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, c4_env)
    spec = pbrhip.make_texture(pbrhip.Format_RGBA32F, S, S, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps | pbrhip.TextureFlag_StorageImage)
    try:
        L.pbrk_mc_set_kernels(1, 1)
        L.PBR_GenPrefilteredEnvMap(tex, spec, 128)
        shipped = [pbrhip.read_mip(spec, m) for m in range(1, 6)]
        g = L.GPU_MakeGraph()
        for m in range(1, 6):
            L.GPU_OpClearColorF(g, spec, m, 0.0, 0.0, 0.0, 0.0)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        L.pbrk_mc_set_kernels(0, 0)
        L.PBR_GenPrefilteredEnvMap(tex, spec, 128)
        for m in range(1, 6):
            a = shipped[m - 1]
            b = pbrhip.read_mip(spec, m)
            assert float(np.abs(a[..., :3]).max()) > 0 and np.array_equal(a[..., 3], b[..., 3])
            err = np.abs(a[..., :3].astype(np.float64) - b[..., :3]) / np.maximum(np.abs(a[..., :3]), 1e-3)
            assert float(err.max()) < 2e-5, (m, float(err.max()))
            shipped[m - 1] = None
    finally:
        L.pbrk_mc_set_kernels(-1, -1)
    L.GPU_DestroyTexture(spec); L.GPU_DestroyTexture(tex)


def test_c3_shade_spheres_oracle_rows(gpu):
    """C3 (1920x1080 metal-rough spheres): full draw == 8 bands of 135 rows, bit for bit; rows through the top sphere row, the
    middle and the bottom (each holds sphere and sky pixels) equal the oracle to 1e-4."""
    from pbrhip import synth
    gbd = synth.synth_gbuffer_spheres(1920, 1080)
    _shade_full_size(gpu, gbd, 8, (150, 540, 930), expect_sky=True)


def test_c5_shade_temple(gpu, c5_gbuffer):
    """C5 (7680x4320 'temple' G-buffer, the draw of render.cpp:1117-1127 at 8K): full draw == 8 bands of 540 rows (the 8-GPU
    screen split), bit for bit; rows through the dome, the column ring and the ground equal the oracle to 1e-4.  The synthetic
    temple is closed by its dome: it has no sky pixels (the sky branch is covered at C3)."""
    _shade_full_size(gpu, c5_gbuffer, 8, (300, 2200, 4100), expect_sky=False, expect_overflow=True)


_VARIANT_CHILD = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "vulkan-pbr-renderer_amd", "python"))
from pbrhip import synth
envs = {W: synth.synth_env(W, seed=0x5EED00AE + W) for W in (256, 1024)}
import pbrhip
L = pbrhip.init(0)
out = {}
for W, S in ((256, 512), (1024, 1024)):
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, envs[W])
    spec = pbrhip.make_texture(pbrhip.Format_RGBA32F, S, S, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps | pbrhip.TextureFlag_StorageImage)
    L.PBR_GenPrefilteredEnvMap(tex, spec, 256)
    for m in range(1, 3):
        if (S >> m) >= 256:
            out[f"{W}_{S}_{m}"] = pbrhip.read_mip(spec, m)[:, ::37].copy()          # every 37th row of every face
    L.GPU_DestroyTexture(spec); L.GPU_DestroyTexture(tex)
st = (C.c_uint64 * 2)()
rc = L.pbrk_mc_region_stats(st, 0)
out["stats"] = np.array([rc, st[0], st[1]], np.int64)
np.savez(sys.argv[2], **out)
L.GPU_WaitUntilIdle(); L.GPU_Deinit()
"""


def test_region_kernel_recorded_variants_agree(gpu, tmp_path):
    """The variant of K4b's region kernel that is kept as the record of a measured dead end -- the fp32-MFMA frame transform
    (PBR_MC_MFMA=1) -- and the level-in-LDS / direct kernels that served these levels before (PBR_MC_REGION=0) still
    compute the same maps as the shipped kernel (same taps and weights; only the order of the additions differs: <= 1e-5
    relative), the region variants with no wave recomputed.  The switches are read once per process: every variant runs in a child."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "variant_child.py"
    script.write_text(_VARIANT_CHILD)
    res = {}
    for name, env in (("default", {}), ("mfma", {"PBR_MC_MFMA": "1"}), ("noregion", {"PBR_MC_REGION": "0"})):
        out = tmp_path / f"{name}.npz"
        e = dict(os.environ, PBR_MC_STATS="1", **env)
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        z = np.load(out)
        res[name] = {k: z[k] for k in z.files}
        rc, healed, total = (int(v) for v in res[name]["stats"])
        if name != "noregion":
            assert rc == 0 and total > 0 and healed == 0, (name, rc, healed, total)
    keys = [k for k in res["default"] if k != "stats"]
    assert len(keys) == 3
    for name in ("mfma", "noregion"):
        for k in keys:
            a, b = res["default"][k].astype(np.float64), res[name][k].astype(np.float64)
            assert float(np.abs(a[..., :3]).max()) > 0
            err = float((np.abs(a - b)[..., :3] / np.maximum(np.abs(a[..., :3]), 1e-3)).max())
            assert err < 1e-5, (name, k, err)
