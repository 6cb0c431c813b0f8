"""GPU parity tests: the HIP path, driven through the C ABI (GPU_* / PBR_* in libgpu_hip.so), against the
CPU oracle and the committed Oracle-A fixtures.  Tolerance from BASELINE.json north_star: 1e-4 relative."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def rmse_rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


@pytest.fixture(scope="module")
def env64(gpu):
    import pbrhip
    from pbrhip import synth
    env = synth.synth_env(64, seed=0x5EED00AA)
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    yield env, tex
    gpu.GPU_DestroyTexture(tex)


def test_mip_chain_bit_exact(gpu, env64):
    """K2 vs oracle (gpu_vulkan.c:1458-1483 semantics): every level of the pyramid, bit for bit."""
    import pbrhip, pbr_oracle as O
    env, tex = env64
    pyr = O.build_pyramid(env)
    assert tex.contents.mip_level_count == 7
    for m in range(7):
        got = pbrhip.read_mip(tex, m)
        want = O.pyramid_level(pyr, 64, m)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"level {m}"


@pytest.mark.parametrize("W", [96, 1000])
def test_mip_chain_of_faces_that_are_not_a_power_of_two(gpu, W):
    """the reference builds 1 + floor(log2 W) levels of max(1, W >> l) texels by linear blits for ANY face size
    (gpu_vulkan.c:1344-1351, 1458-1483, 2786-2826; asset_import.cpp:21 only asserts y == 6x).  W = 96: 96, 48, 24, 12, 6, 3, 1 (the
    last level resamples an odd source); W = 1000: 1000, 500, 250, 125, 62, 31, 15, 7, 3, 1 (five odd sources).  Even steps are the
    2x2 box, odd ones the linear resample stated in oracle/pbr_oracle.c A2; GPU == oracle bit for bit on every level, through the
    uploading GPU_MakeTexture, through GPU_OpGenerateMipmaps and through explicit GPU_OpBlit calls, as the reference issues them."""
    import pbrhip, pbr_oracle as O
    from pbrhip import synth
    L = gpu
    env = synth.synth_env(W, seed=0x5EED00B0 + W, workers=1)
    pyr = O.build_pyramid(env)
    levels = O.mip_count(W)
    assert levels == 1 + int(np.floor(np.log2(W)))
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)     # upload + chain
    assert tex.contents.mip_level_count == levels
    for m in range(levels):
        got = pbrhip.read_mip(tex, m)
        want = O.pyramid_level(pyr, W, m)
        assert got.shape == want.shape == (6, max(1, W >> m), max(1, W >> m), 4)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"upload path, level {m}"
    g = L.GPU_MakeGraph()
    for m in range(1, levels):
        L.GPU_OpClearColorF(g, tex, m, 0.0, 0.0, 0.0, 0.0)
    L.GPU_OpGenerateMipmaps(g, tex)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    for m in range(levels):
        assert np.array_equal(pbrhip.read_mip(tex, m).view(np.uint32), O.pyramid_level(pyr, W, m).view(np.uint32)), f"GPU_OpGenerateMipmaps, level {m}"
    if W <= 128:                                    # the reference's own loop: one blit per (level, face)
        class Off(C.Structure):
            _fields_ = [("x", C.c_int), ("y", C.c_int), ("z", C.c_int)]

        class Blit(C.Structure):                    # GPU_OpBlitInfo [gpu.h:317-327]
            _fields_ = [("filter", C.c_int), ("src_texture", pbrhip.TexP), ("dst_texture", pbrhip.TexP), ("src_layer", C.c_uint32), ("dst_layer", C.c_uint32),
                        ("src_mip_level", C.c_uint32), ("dst_mip_level", C.c_uint32), ("src_area", Off * 2), ("dst_area", Off * 2)]
        for m in range(1, levels):
            L.GPU_OpClearColorF(g, tex, m, 0.0, 0.0, 0.0, 0.0)
        sw = W
        for m in range(1, levels):
            for layer in range(6):
                b = Blit(); b.filter = 0; b.src_texture = tex; b.src_mip_level = m - 1; b.src_layer = layer; b.dst_texture = tex; b.dst_mip_level = m; b.dst_layer = layer
                b.src_area[1] = Off(sw, sw, 1); b.dst_area[1] = Off(sw // 2, sw // 2, 1)
                L.GPU_OpBlit(g, C.byref(b))
            sw //= 2
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
        for m in range(levels):
            assert np.array_equal(pbrhip.read_mip(tex, m).view(np.uint32), O.pyramid_level(pyr, W, m).view(np.uint32)), f"GPU_OpBlit chain, level {m}"
    L.GPU_DestroyGraph(g); L.GPU_DestroyTexture(tex)


def test_linear_blit_between_rectangles_and_equirect_cube_of_any_size(gpu):
    """The linear resample of GPU_OpBlit beyond mip chains: whole RGBA32F 2-D subresources of unrelated sizes (37 x 21 -> 16 x 9 down,
    5 x 3 -> 12 x 8 up: clamp to edge) equal the oracle's rule bit for bit; and the equirect loader builds a 96^2 cube (restriction to
    powers of two lifted) whose levels 96 .. 1 equal the oracle's chain of its level 0."""
    import pbrhip, pbr_oracle as O
    L = gpu
    rng = np.random.default_rng(0x5EED00C7)

    class Off(C.Structure):
        _fields_ = [("x", C.c_int), ("y", C.c_int), ("z", C.c_int)]

    class Blit(C.Structure):                    # GPU_OpBlitInfo [gpu.h:317-327]
        _fields_ = [("filter", C.c_int), ("src_texture", pbrhip.TexP), ("dst_texture", pbrhip.TexP), ("src_layer", C.c_uint32), ("dst_layer", C.c_uint32),
                    ("src_mip_level", C.c_uint32), ("dst_mip_level", C.c_uint32), ("src_area", Off * 2), ("dst_area", Off * 2)]
    g = L.GPU_MakeGraph()
    for (sw, sh, dw, dh) in ((37, 21, 16, 9), (5, 3, 12, 8), (64, 64, 64, 17)):
        src = (rng.random((sh, sw, 4), dtype=np.float32) * 9.0).astype(np.float32)
        ts = pbrhip.make_texture(pbrhip.Format_RGBA32F, sw, sh, pbrhip.TextureFlag_RenderTarget)
        td = pbrhip.make_texture(pbrhip.Format_RGBA32F, dw, dh, pbrhip.TextureFlag_RenderTarget)
        pbrhip.upload_mip(ts, 0, src)
        b = Blit(); b.filter = 0; b.src_texture = ts; b.dst_texture = td
        b.src_area[1] = Off(sw, sh, 1); b.dst_area[1] = Off(dw, dh, 1)
        L.GPU_OpBlit(g, C.byref(b)); L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
        got = pbrhip.read_mip(td, 0)
        want = O.blit_linear(src[None], dw, dh)[0]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (sw, sh, dw, dh)
        L.GPU_DestroyTexture(ts); L.GPU_DestroyTexture(td)
    L.GPU_DestroyGraph(g)
    h, w = 64, 128
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    pano = np.ones((h, w, 4), np.float32)
    pano[..., 0] = 0.5 + 0.4 * np.sin(xx * 2 * np.pi / w * 3); pano[..., 1] = 0.2 + yy / h; pano[..., 2] = rng.random((h, w)).astype(np.float32)
    tex = L.GPUX_MakeCubemapFromEquirect(pano.ctypes.data_as(C.c_void_p), w, h, 96, 0)
    assert tex and tex.contents.width == 96 and tex.contents.mip_level_count == 7
    l0 = pbrhip.read_mip(tex, 0)
    want0 = O.equirect_to_cube(pano, 96)
    assert (np.abs(l0.astype(np.float64) - want0) / np.maximum(np.abs(want0), 1e-3)).max() < 1e-4
    pyr = O.build_pyramid(l0)
    for m in range(7):
        assert np.array_equal(pbrhip.read_mip(tex, m).view(np.uint32), O.pyramid_level(pyr, 96, m).view(np.uint32)), m
    L.GPU_DestroyTexture(tex)


def test_precompute_from_an_environment_that_is_not_a_power_of_two(gpu):
    """The whole precompute from a 96^2 HDR cube (levels 96 .. 1): K4a copies from the 48^2 level, K4b filters the 12^2, 6^2, 3^2
    levels, K3 the 1^2 level -- source sizes none of the power-of-two tests reach -- vs the oracle at 1e-4."""
    import pbrhip, pbr_oracle as O
    from pbrhip import synth
    L = gpu
    W = 96
    env = synth.synth_env(W, seed=0x5EED00B7, workers=1)
    pyr = O.build_pyramid(env)
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    spec = _run_prefilter(L, tex, 64, 1)
    for m in range(7):
        got = pbrhip.read_mip(spec, m)
        want = O.prefilter_mip(pyr, W, 64, m)
        assert rel_err(got, want, floor=1e-3) < REL, f"mip {m}: {rel_err(got, want, floor=1e-3)}"
    irr = pbrhip.make_texture(pbrhip.Format_RGBA32F, 16, 16, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage)
    L.PBR_GenIrradianceMap(tex, irr)
    want = O.irradiance(pyr, W, 16)
    assert rel_err(pbrhip.read_mip(irr, 0)[..., :3], want[..., :3], floor=1e-3) < REL
    L.GPU_DestroyTexture(irr); L.GPU_DestroyTexture(spec); L.GPU_DestroyTexture(tex)


def test_brdf_lut_fp32_and_fp16(gpu, golden_dir):
    """K1 vs the Oracle-A LUT (reference shader text executed on the CPU)."""
    import pbrhip, pbr_oracle as O
    want = np.load(os.path.join(golden_dir, "oracle_a_lut256.npy"))
    t32 = pbrhip.make_texture(pbrhip.Format_RG32F, 256, 256, pbrhip.TextureFlag_StorageImage)
    gpu.PBR_GenBRDFIntegrationMap(t32)
    got = pbrhip.read_mip(t32, 0)
    gpu.GPU_DestroyTexture(t32)
    assert not np.isnan(got).any()
    assert rmse_rel(got, want) < REL
    # per-texel: relative to the texel's own magnitude, with a floor for the near-zero bias entries
    err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    assert err.max() < REL, f"max rel {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    # RG16F store (render.cpp:795): bits equal to RTE(oracle) or one ulp away
    t16 = pbrhip.make_texture(pbrhip.Format_RG16F, 256, 256, pbrhip.TextureFlag_StorageImage)
    gpu.PBR_GenBRDFIntegrationMap(t16)
    h = pbrhip.read_mip(t16, 0).view(np.uint16).astype(np.int32)
    gpu.GPU_DestroyTexture(t16)
    ref = want.astype(np.float16).view(np.uint16).astype(np.int32)
    assert np.abs(h - ref).max() <= 1
    assert (h != ref).mean() < 0.01


def _run_prefilter(gpu, env_tex, out_size, min_size):
    import pbrhip
    spec = pbrhip.make_texture(pbrhip.Format_RGBA32F, out_size, out_size,
                               pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps | pbrhip.TextureFlag_StorageImage)
    gpu.PBR_GenPrefilteredEnvMap(env_tex, spec, min_size)
    return spec


@pytest.mark.parametrize("nsamples", [5000, 12288])
def test_brdf_lut_any_sample_count(gpu, nsamples):
    """K1 keeps the sample directions in LDS; since round 3 they pass through it in chunks of 4096, so sample counts
    beyond one chunk (a ragged second chunk: 5000; three chunks: 12288 -- more than the 9600 that once fitted) work and equal the
    oracle.  Low-level kernel ABI (pbrk_brdf_lut) with the host tables, 64^2 map, fp32 target."""
    import pbrhip, pbr_oracle as O
    L = gpu
    size = 64
    ang = np.zeros((nsamples, 4), np.float32)
    L.pbrk_host_sample_angles(nsamples, ang.ctypes.data_as(C.c_void_p))
    libm = C.CDLL("libm.so.6")                                                      # the host's fp32 libm, as the backend builds its table (gpu_hip.cpp)
    for fn in (libm.acosf, libm.cosf, libm.sinf):
        fn.restype = C.c_float; fn.argtypes = [C.c_float]
    vcs = np.zeros((size, 2), np.float32)
    for x in range(size):                                                           # V = Rotate((0,0,1), (1,0,0), acos(NdotV)): cos / sin of that angle
        th = libm.acosf(C.c_float(float((np.float32(x) + np.float32(0.5)) / np.float32(size))))
        vcs[x] = (libm.cosf(th), libm.sinf(th))
    d_ang = L.GPU_MakeBuffer(ang.nbytes, pbrhip.BufferFlag_GPU, ang.ctypes.data_as(C.c_void_p))
    d_vcs = L.GPU_MakeBuffer(vcs.nbytes, pbrhip.BufferFlag_GPU, vcs.ctypes.data_as(C.c_void_p))
    out = L.GPU_MakeBuffer(size * size * 8, pbrhip.BufferFlag_GPU, None)
    host = L.GPU_MakeBuffer(size * size * 8, pbrhip.BufferFlag_CPU, None)
    assert L.pbrk_brdf_lut(L.GPUX_BufferDevicePtr(out), pbrhip.PBRK_FMT_RG32F, size, nsamples, L.GPUX_BufferDevicePtr(d_ang), L.GPUX_BufferDevicePtr(d_vcs), 0, size, None) == 0
    g = L.GPU_MakeGraph()
    L.GPU_WaitUntilIdle()
    L.GPU_OpCopyBufferToBuffer(g, out, host, 0, 0, size * size * 8); L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    got = np.frombuffer((C.c_char * (size * size * 8)).from_address(host.contents.data), np.float32).reshape(size, size, 2).copy()
    want = O.brdf_lut(size, nsamples)[..., :2]
    e = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    assert e.max() < REL, float(e.max())
    L.GPU_DestroyGraph(g)
    for b in (d_ang, d_vcs, out, host):
        L.GPU_DestroyBuffer(b)


def test_prefilter_env64_vs_oracle_a(gpu, env64, golden_dir):
    """K4a/K4b on the textured W=64 environment vs Oracle-A fixtures (mips 0-2 of a 64^2 cube)."""
    import pbrhip
    env, tex = env64
    spec = _run_prefilter(gpu, tex, 64, 16)
    for m in (0, 1, 2):
        want = np.load(os.path.join(golden_dir, f"oracle_a_prefilter_env64_out64_mip{m}.npy"))
        got = pbrhip.read_mip(spec, m)
        assert got.shape == want.shape
        assert rmse_rel(got, want) < REL, f"mip {m}"
        assert rel_err(got, want, floor=1e-3) < REL, f"mip {m}: {rel_err(got, want, floor=1e-3)}"
        if m > 0:   # alpha channel = sum of weights, texel independent (shader order on the host)
            assert np.all(got[..., 3] == got[0, 0, 0, 3])
            assert got[0, 0, 0, 3] == want[0, 0, 0, 3]
    gpu.GPU_DestroyTexture(spec)


def test_prefilter_full_chain_vs_oracle(gpu, env64):
    """Whole chain down to 1x1 (mips >= 5 use the documented extension rule), vs Oracle-B."""
    import pbrhip, pbr_oracle as O
    env, tex = env64
    pyr = O.build_pyramid(env)
    spec = _run_prefilter(gpu, tex, 32, 1)
    for m in range(6):
        got = pbrhip.read_mip(spec, m)
        want = O.prefilter_mip(pyr, 64, 32, m)
        assert rel_err(got, want, floor=1e-3) < REL, f"mip {m}: {rel_err(got, want, floor=1e-3)}"
    gpu.GPU_DestroyTexture(spec)


def test_irradiance_env256_vs_oracle_a(gpu, golden_dir):
    """K3 (32^2, 1024 samples, env LOD 6 of a 256^2 cube) vs the Oracle-A fixture."""
    import pbrhip
    from pbrhip import synth
    env = synth.synth_env(256, seed=0x5EED00AB)
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 256, 256, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    irr = pbrhip.make_texture(pbrhip.Format_RGBA32F, 32, 32, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage)
    gpu.PBR_GenIrradianceMap(tex, irr)
    got = pbrhip.read_mip(irr, 0)
    want = np.load(os.path.join(golden_dir, "oracle_a_irradiance_env256.npy"))
    assert rmse_rel(got, want) < REL
    assert rel_err(got[..., :3], want[..., :3], floor=1e-3) < REL
    assert np.all(got[..., 3] == 0)
    gpu.GPU_DestroyTexture(irr); gpu.GPU_DestroyTexture(tex)


def _shade_setup(gpu, W, H, result_format):
    import pbrhip
    from pbrhip import synth
    L = gpu
    gbd = synth.synth_gbuffer_spheres(W, H)
    env = synth.synth_env(64, seed=0x5EED00AA)
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 64)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
    L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16)
    L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, result_format)
    for name, arr in (("base_color", gbd["base"]), ("normal", gbd["normal"]), ("orm", gbd["orm"]), ("emissive", gbd["emissive"]),
                      ("depth", gbd["depth"])):
        pbrhip.upload_mip(getattr(gb, name), 0, arr)
    lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H)
    return gbd, env_tex, maps, gb, lp, glob


def _oracle_shade(gpu, gbd, maps, glob, flags):
    import pbrhip, pbr_oracle as O
    irr = pbrhip.read_mip(maps.irradiance_map, 0)
    n = maps.tex_specular_env_map.contents.mip_level_count
    size = maps.tex_specular_env_map.contents.width
    pyr = np.concatenate([pbrhip.read_mip(maps.tex_specular_env_map, m).ravel() for m in range(n)])
    lut = pbrhip.read_mip(maps.brdf_lut, 0).view(np.uint16)
    g = O.OrcGlobals.from_buffer_copy(bytes(glob))
    return O.shade(g, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"], flags=flags,
                   irradiance_cube=irr, prefiltered_pyr=pyr, prefiltered_size=size, lut_half=lut)


@pytest.mark.parametrize("mode", ["ibl", "live", "live_shafts"])
def test_shade_fp32_vs_oracle(gpu, mode):
    """K5 on a 256x144 synthetic metal-rough-spheres G-buffer, fp32 target, vs Oracle-B fed the same GPU-built maps."""
    import pbrhip, pbr_oracle as O
    W, H = 256, 144
    gbd, env_tex, maps, gb, lp, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA32F)
    flags = {"ibl": pbrhip.Shade_IBL, "live": 0, "live_shafts": pbrhip.Shade_LightShafts}[mode]
    gpu.GPUX_SetShadeFlags(gpu.PBR_LightingPipeline(lp), flags)
    g = gpu.GPU_MakeGraph()
    gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    got = pbrhip.read_mip(gb.lighting_result, 0)
    oflags = {"ibl": O.SHADE_IBL, "live": 0, "live_shafts": O.SHADE_SHAFTS}[mode]
    want = _oracle_shade(gpu, gbd, maps, glob, oflags)
    assert (gbd["depth"] < 1).mean() > 0.1 and (gbd["depth"] == 1).mean() > 0.1   # both geometry and sky are present
    assert rmse_rel(got, want) < REL
    assert rel_err(got[..., :3], want[..., :3], floor=1e-2) < REL, rel_err(got[..., :3], want[..., :3], floor=1e-2)
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex)


def test_shade_fp16_target(gpu):
    """RGBA16F target (render.cpp:693): bits equal to RTE(oracle fp32) or one ulp away."""
    import pbrhip, pbr_oracle as O
    W, H = 128, 72
    gbd, env_tex, maps, gb, lp, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA16F)
    g = gpu.GPU_MakeGraph()
    gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    got = pbrhip.read_mip(gb.lighting_result, 0).view(np.uint16).astype(np.int32)
    want = _oracle_shade(gpu, gbd, maps, glob, O.SHADE_IBL).astype(np.float16).view(np.uint16).astype(np.int32)
    assert np.abs(got - want).max() <= 1
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex)


def test_shade_fp16_overflow_store(gpu):
    """The RGBA16F store of values beyond the half range: lighting_pass.glsl:712 clamps the colour
    from below only, and at low roughness the sun term reaches 1e5 (D = 1 / (pi a^2) at N.H = 1).  Defined here as the IEEE
    round-to-nearest-even conversion: fp32 values from 65520 on store +inf (0x7C00), below that the nearest half (65504 = 0x7BFF at
    most).  A mirror-like band whose normals point along the half vector of sun and view produces such pixels; K5's target must hold
    exactly the oracle's converted bits there (and be within one ulp everywhere else)."""
    import pbrhip, pbr_oracle as O
    W, H = 128, 72
    gbd, env_tex, maps, gb, lp, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA16F)
    # per pixel: N = normalize(L + V) quantised to bytes, roughness 13/255, white dielectric, depth of a plane in front of the camera
    g32 = np.frombuffer(bytes(glob), np.float32)
    wfc = g32[32:48].reshape(4, 4).T.astype(np.float64)
    sun = g32[128:131].astype(np.float64); cam = g32[132:135].astype(np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    depth = np.full((H, W), 0.9990, np.float32)
    ndc = np.stack([(xs + 0.5) / W * 2 - 1, (ys + 0.5) / H * 2 - 1, depth.astype(np.float64), np.ones((H, W))], -1)
    pw = ndc @ wfc.T
    P = pw[..., :3] / pw[..., 3:]
    V = cam - P; V /= np.linalg.norm(V, axis=-1, keepdims=True)
    Hh = V - sun; Hh /= np.linalg.norm(Hh, axis=-1, keepdims=True)
    nrm = np.zeros((H, W, 4), np.uint8); nrm[..., :3] = np.clip(np.rint((Hh * 0.5 + 0.5) * 255), 0, 255); nrm[..., 3] = 255
    orm = np.zeros((H, W, 4), np.uint8); orm[..., 0] = 255; orm[..., 1] = 13; orm[H // 2:, :, 1] = 40
    base = np.full((H, W, 4), 255, np.uint8); emi = np.zeros((H, W, 4), np.uint8)
    gbd = dict(gbd, base=base, normal=nrm, orm=orm, emissive=emi, depth=depth)
    for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, name), 0, gbd[key])
    g = gpu.GPU_MakeGraph()
    gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    got = pbrhip.read_mip(gb.lighting_result, 0).view(np.uint16).astype(np.int32)
    want = _oracle_shade(gpu, gbd, maps, glob, O.SHADE_IBL)
    wb = O.f32_to_f16_bits(want.astype(np.float32)).astype(np.int32)            # the oracle's own conversion (== numpy's RTE, inf from 65520 on)
    over = want[..., :3] >= 65520.0 * (1.0 + 2e-4)
    assert over.sum() > 50 and (want[..., :3] < 60000.0).sum() > 50, (int(over.sum()), float(want[..., :3].max()))
    assert (got[..., :3][over] == 0x7C00).all()
    assert np.abs(got[..., :3] - wb[..., :3]).max() <= 1
    assert (got[..., 3] == 0x3C00).all()
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex)


def test_lighting_tile_vs_oracle_a(gpu, golden_dir):
    """K5 through the low-level kernel ABI on the Oracle-A lighting tile is covered on CPU for the oracle;
    here: the sharded draw (GPUX_OpDrawRows) equals the full draw."""
    import pbrhip
    W, H = 128, 72
    gbd, env_tex, maps, gb, lp, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA32F)
    g = gpu.GPU_MakeGraph()
    gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    full = pbrhip.read_mip(gb.lighting_result, 0)
    gpu.GPU_OpClearColorF(g, gb.lighting_result, 0, 0.0, 0.0, 0.0, 0.0)
    for r0, r1 in ((0, 17), (17, 50), (50, 72)):
        gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), r0, r1)
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    banded = pbrhip.read_mip(gb.lighting_result, 0)
    assert np.array_equal(full.view(np.uint32), banded.view(np.uint32))
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex)


def test_sharded_units_equal_full_dispatch(gpu, env64):
    """Work-unit dispatch (multi-GPU sharding path) reproduces the single full dispatch bit for bit."""
    import pbrhip
    env, tex = env64
    L = gpu
    full = _run_prefilter(gpu, tex, 64, 1)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 64, 64)
    pipes = L.PBR_MakeIBLPipelines()
    arena = L.GPU_MakeDescriptorArena()
    g = L.GPU_MakeGraph()
    for rank in range(3):
        units, n = pbrhip.partition(64, 1, 32, 64, 3, rank)
        L.PBR_RecordUnits(pipes, g, arena, tex, C.byref(maps), units, n)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    for m in range(7):
        a = pbrhip.read_mip(full, m); b = pbrhip.read_mip(maps.tex_specular_env_map, m)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"mip {m}"
    irr_full = pbrhip.make_texture(pbrhip.Format_RGBA32F, 32, 32, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage)
    L.PBR_GenIrradianceMap(tex, irr_full)
    assert np.array_equal(pbrhip.read_mip(irr_full, 0).view(np.uint32), pbrhip.read_mip(maps.irradiance_map, 0).view(np.uint32))
    L.GPU_DestroyGraph(g); L.GPU_DestroyDescriptorArena(arena); L.PBR_DestroyIBLPipelines(pipes)
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(full); L.GPU_DestroyTexture(irr_full)


def test_tile_streams_keep_order_between_dependent_dispatches(gpu, env64):
    """Small independent tile dispatches overlap on side streams (GPUX_SetTileStreams); a dispatch that samples what an earlier
    one in the same graph wrote must still see it.  Chain env -> A -> B recorded in one graph: bit-identical with 0, 3, 4 streams."""
    import pbrhip
    env, tex = env64
    L = gpu
    results = []
    for streams in (0, 4, 3, -1):
        L.GPUX_SetTileStreams(streams)
        a = pbrhip.PBR_IBLMaps(); b = pbrhip.PBR_IBLMaps()
        L.PBR_MakeIBLMaps(C.byref(a), 32, 64, 64)
        L.PBR_MakeIBLMaps(C.byref(b), 16, 64, 32)
        pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena(); g = L.GPU_MakeGraph()
        for rank in range(3):
            units, n = pbrhip.partition(64, 1, 32, 64, 3, rank)
            L.PBR_RecordUnits(pipes, g, arena, tex, C.byref(a), units, n)
        for rank in range(2):                                  # second stage samples the map the first stage is still writing
            units, n = pbrhip.partition(32, 1, 16, 64, 2, rank)
            L.PBR_RecordUnits(pipes, g, arena, a.tex_specular_env_map, C.byref(b), units, n)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
        results.append([pbrhip.read_mip(b.tex_specular_env_map, m).copy() for m in range(6)] + [pbrhip.read_mip(b.irradiance_map, 0).copy()])
        L.GPU_DestroyGraph(g); L.GPU_DestroyDescriptorArena(arena); L.PBR_DestroyIBLPipelines(pipes)
        L.PBR_DestroyIBLMaps(C.byref(a)); L.PBR_DestroyIBLMaps(C.byref(b))
    assert float(np.abs(results[0][1]).max()) > 0.0
    for r in results[1:]:
        for x, y in zip(results[0], r):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_shade_random_bytes_all_decodes(gpu):
    """Every unorm8 value in every channel (exact b/255 decode), non-unit normals, random depths incl. sky."""
    import pbrhip, pbr_oracle as O
    W, H = 96, 64
    gbd, env_tex, maps, gb, lp, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA32F)
    rng = np.random.default_rng(0x5EED00AE)
    for key in ("base", "normal", "orm", "emissive"):
        gbd[key] = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    gbd["base"][0, :256 % W] = 0
    gbd["orm"][1, :, 1] = np.arange(W, dtype=np.uint8) * 2            # a roughness ramp
    gbd["emissive"][rng.random((H, W)) > 0.1] = 0
    gbd["depth"] = (0.9980 + 0.0019 * rng.random((H, W))).astype(np.float32)
    gbd["depth"][rng.random((H, W)) < 0.15] = 1.0
    for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, name), 0, gbd[key])
    glob = pbrhip.fill_globals((0, 0, 5), aspect=W / H, frame_idx=17)
    g = gpu.GPU_MakeGraph()
    gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    got = pbrhip.read_mip(gb.lighting_result, 0)
    want = _oracle_shade(gpu, gbd, maps, glob, O.SHADE_IBL)
    assert not np.isnan(got).any()
    assert rel_err(got[..., :3], want[..., :3], floor=1e-2) < REL, rel_err(got[..., :3], want[..., :3], floor=1e-2)
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex)


@pytest.mark.parametrize("size,mode", [((333, 131), "ibl"), ((1000, 564), "ibl"), ((640, 360), "shafts"), ((256, 144), "sun_only")])
def test_shade_tile_kernel_equals_fast_kernel_and_oracle(gpu, size, mode):
    """K5's tiled instantiation (k_shade_tile.hip: prefiltered taps from LDS-staged windows placed by each 64 x 16 tile's centre
    pixel, per-column / per-row host tables) against the fast instantiation: the same bits on every pixel, for frame sizes that are
    not multiples of the tile, row bands cut inside tiles, all three fast modes; plus the oracle at 1e-4.  The spheres scene at these
    sizes has tiles whose lanes fall outside their windows (sphere silhouettes, face changes), so both the LDS and the fallback
    path of one wave are exercised; random bytes make every lane a fallback."""
    import pbrhip, pbr_oracle as O
    W, H = size
    gbd, env_tex, maps, gb, lp, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA32F)
    flags, oflags = {"ibl": (pbrhip.Shade_IBL, O.SHADE_IBL), "shafts": (pbrhip.Shade_IBL | pbrhip.Shade_LightShafts, O.SHADE_IBL | O.SHADE_SHAFTS),
                     "sun_only": (0, 0)}[mode]
    gpu.GPUX_SetShadeFlags(gpu.PBR_LightingPipeline(lp), flags)
    glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H, frame_idx=23)
    g = gpu.GPU_MakeGraph()

    def run(tile, bands):
        gpu.pbrk_shade_set_tile_min_pixels(0 if tile else 1 << 40)
        gpu.GPU_OpClearColorF(g, gb.lighting_result, 0, -7.0, -7.0, -7.0, -7.0)
        for (r0, r1) in bands:
            gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), r0, r1)
        gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
        return pbrhip.read_mip(gb.lighting_result, 0).copy()

    try:
        fast = run(False, [(0, 0)])
        tile = run(True, [(0, 0)])
        assert np.array_equal(fast.view(np.uint32), tile.view(np.uint32)), int((fast.view(np.uint32) != tile.view(np.uint32)).any(-1).sum())
        banded = run(True, [(0, 5), (5, 37), (37, H - 1), (H - 1, H)])            # bands that start and end inside tiles
        assert np.array_equal(banded.view(np.uint32), tile.view(np.uint32))
        want = _oracle_shade(gpu, gbd, maps, glob, oflags)
        assert rel_err(tile[..., :3], want[..., :3], floor=1e-2) < REL, rel_err(tile[..., :3], want[..., :3], floor=1e-2)
        if mode == "ibl":                                                           # wild inputs: every lane its own direction
            rng = np.random.default_rng(0x5EED00AF)
            for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive")):
                pbrhip.upload_mip(getattr(gb, name), 0, rng.integers(0, 256, (H, W, 4), dtype=np.uint8))
            depth = (0.9980 + 0.0019 * rng.random((H, W))).astype(np.float32); depth[rng.random((H, W)) < 0.2] = 1.0
            pbrhip.upload_mip(gb.depth, 0, depth)
            a, b = run(False, [(0, 0)]), run(True, [(0, 0)])
            assert not np.isnan(b).any() and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        gpu.pbrk_shade_set_tile_min_pixels(-1)
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex)


def test_fill_pattern_every_size_and_alignment(gpu):
    """pbrk_fill_pattern (the clear of GPU_OpClearColorF / I): every pattern size, destinations that start 0 .. 15 bytes past a 16-byte
    boundary, lengths with and without a sub-16-byte head and tail, guard bytes on both sides untouched; and through the API: an
    all-levels clear of a texture whose byte count is not a multiple of 16 (the 1080p bloom target's case) to a non-zero value."""
    import pbrhip
    rng = np.random.default_rng(0x5EED00C1)
    N = 1 << 16
    hb = gpu.GPU_MakeBuffer(N, pbrhip.BufferFlag_CPU, None)                     # pinned host memory the device can write
    host = np.frombuffer((C.c_uint8 * N).from_address(hb.contents.data), dtype=np.uint8)
    dptr = gpu.GPUX_BufferDevicePtr(hb)
    for pb in (1, 2, 4, 8, 16):
        pat = rng.integers(1, 255, pb, dtype=np.uint8)
        cpat = (C.c_uint8 * pb)(*pat.tolist())
        for start in range(0, 32, pb):
            for nbytes in (0, pb, 16, 48, 4096 + pb, 40000 // pb * pb + pb):
                host[:] = 0xEE
                assert gpu.pbrk_fill_pattern(dptr + 64 + start, nbytes, cpat, pb, None) == 0
                gpu.GPU_WaitUntilIdle()
                want = np.full(N, 0xEE, np.uint8)
                want[64 + start:64 + start + nbytes] = np.tile(pat, nbytes // pb)
                assert np.array_equal(host, want), (pb, start, nbytes)
    assert gpu.pbrk_fill_pattern(dptr + 1, 8, (C.c_uint8 * 4)(1, 2, 3, 4), 4, None) != 0                # misaligned for its pattern
    assert gpu.pbrk_fill_pattern(dptr, 6, (C.c_uint8 * 4)(1, 2, 3, 4), 4, None) != 0                    # not a whole number of patterns
    gpu.GPU_DestroyBuffer(hb)
    t = gpu.GPU_MakeTexture(pbrhip.Format_RGBA32F, 30, 17, 1, pbrhip.TextureFlag_HasMipmaps | pbrhip.TextureFlag_RenderTarget, None)
    g = gpu.GPU_MakeGraph()
    gpu.GPU_OpClearColorF(g, t, 0xFFFFFFFF, 1.5, -2.0, 3.25, 4.0)                 # GPU_MIP_LEVEL_ALL
    gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    for m in range(t.contents.mip_level_count):
        lv = pbrhip.read_mip(t, m)
        assert np.array_equal(lv.reshape(-1, 4), np.tile(np.float32([1.5, -2.0, 3.25, 4.0]), (lv.size // 4, 1))), m
    gpu.GPU_DestroyGraph(g); gpu.GPU_DestroyTexture(t)


def test_cube_sampler_convention_switch(gpu, env64):
    """the reference leaves the cube sampler's arithmetic to the driver (gpu_vulkan.c:613-634); this repo's default is
    exact fp32 tap weights, real texture units resolve 8 sub-texel / LOD-fraction bits.  pbrk_set_cube_sampler_snap(1) /
    orc_set_cube_sampler_snap(1) switch kernels (their general instantiations) and oracle to the snapped convention: (a) under EITHER
    convention GPU == oracle at 1e-4 (prefilter mips 0-3, irradiance, a shaded frame), (b) the two conventions differ by far more
    than that -- the numbers DESIGN.md 7 quotes for the full-size configs come from tools/sampler_delta.py, same switch."""
    import pbrhip, pbr_oracle as O
    L = gpu
    env, tex = env64
    pyr = O.build_pyramid(env)
    res = {}
    try:
        for snap in (0, 1):
            L.pbrk_set_cube_sampler_snap(snap); O.set_cube_sampler_snap(snap)
            spec = _run_prefilter(L, tex, 64, 8)
            for m in range(4):
                got = pbrhip.read_mip(spec, m); want = O.prefilter_mip(pyr, 64, 64, m)
                assert rel_err(got, want, floor=1e-3) < REL, (snap, m, rel_err(got, want, floor=1e-3))
                res[(snap, "pre", m)] = got
            irr = pbrhip.make_texture(pbrhip.Format_RGBA32F, 16, 16, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage)
            L.PBR_GenIrradianceMap(tex, irr)
            got = pbrhip.read_mip(irr, 0); want = O.irradiance(pyr, 64, 16)
            assert rel_err(got[..., :3], want[..., :3], floor=1e-3) < REL, snap
            res[(snap, "irr")] = got
            L.GPU_DestroyTexture(irr); L.GPU_DestroyTexture(spec)
            gbd, env_tex, maps, gb, lp, glob = _shade_setup(L, 128, 72, pbrhip.Format_RGBA32F)
            g = L.GPU_MakeGraph()
            L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
            L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
            got = pbrhip.read_mip(gb.lighting_result, 0).copy()
            want = _oracle_shade(L, gbd, maps, glob, O.SHADE_IBL)
            assert rel_err(got[..., :3], want[..., :3], floor=1e-2) < REL, (snap, rel_err(got[..., :3], want[..., :3], floor=1e-2))
            res[(snap, "frame")] = got
            L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb))
            L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(env_tex)
    finally:
        L.pbrk_set_cube_sampler_snap(0); O.set_cube_sampler_snap(0)
    d_copy = rel_err(res[(1, "pre", 0)], res[(0, "pre", 0)], floor=1e-3)          # texel centres of a 2:1 copy sit on multiples of 1/4 texel: immune
    d_mc = rel_err(res[(1, "pre", 2)], res[(0, "pre", 2)], floor=1e-3)            # 8192 arbitrary tap positions per texel
    d_frame = rel_err(res[(1, "frame")][..., :3], res[(0, "frame")][..., :3], floor=1e-2)
    print(f"sampler convention delta (env 64): copy level {d_copy:.2e}, Monte-Carlo level {d_mc:.2e}, shaded frame {d_frame:.2e}")
    assert d_copy < REL and d_mc > REL and d_frame > REL, (d_copy, d_mc, d_frame)


def test_prefilter_tolerance_budgeted_sample_cut(gpu, env64):
    """GPUX_SetPrefilterTolerance (opt-in): a prefilter dispatch keeps the shortest prefix of its weight table whose
    dropped tail is rigorously bounded -- tail * max(level) <= rel * head * min(level) -- so every texel moves by at most `rel`
    relative.  On the textured 64^2 environment with rel = 1e-7: mip 1 (roughness 0.03) keeps a fraction of its 1389 samples, the result
    stays within 1e-6 of the exact GPU sum (rel + fp32 rounding of a shorter sum) and within 1e-4 of the oracle; rougher mips keep
    (almost) everything; an environment with one black texel is never cut; rel = 0 restores the exact sums bit for bit."""
    import pbrhip, pbr_oracle as O
    L = gpu
    env, tex = env64
    pyr = O.build_pyramid(env)
    exact = _run_prefilter(L, tex, 64, 8)
    ref = [pbrhip.read_mip(exact, m).copy() for m in range(4)]
    full = [L.GPUX_PrefilterKeptSamples(m) for m in range(4)]
    assert full[1] == 1389 and full[2] == 8192
    try:
        L.GPUX_SetPrefilterTolerance(1e-7)
        cut = _run_prefilter(L, tex, 64, 8)
        kept = [L.GPUX_PrefilterKeptSamples(m) for m in range(4)]
        assert 100 < kept[1] < 900, kept                                  # exp(-i / 14.7) decay: a few hundred samples carry all but 1e-7 / range
        assert kept[2] <= full[2] and kept[3] <= full[3]
        for m in range(1, 4):
            got = pbrhip.read_mip(cut, m)
            assert rel_err(got[..., :3], ref[m][..., :3], floor=1e-3) < 1e-6, (m, rel_err(got[..., :3], ref[m][..., :3], floor=1e-3))
            assert np.array_equal(got[..., 3], ref[m][..., 3])             # alpha = the full weight sum, evaluated on the host: untouched
            assert rel_err(got, O.prefilter_mip(pyr, 64, 64, m), floor=1e-3) < REL
        L.GPU_DestroyTexture(cut)
        # a black texel anywhere in the sampled level: the bound is void, nothing is cut
        env0 = env.copy(); env0[2, :16, :16, :3] = 0.0
        tex0 = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env0)
        s0 = _run_prefilter(L, tex0, 64, 8)
        assert L.GPUX_PrefilterKeptSamples(1) == 1389
        L.GPU_DestroyTexture(s0); L.GPU_DestroyTexture(tex0)
    finally:
        L.GPUX_SetPrefilterTolerance(0.0)
    again = _run_prefilter(L, tex, 64, 8)
    for m in range(4):
        assert np.array_equal(pbrhip.read_mip(again, m).view(np.uint32), ref[m].view(np.uint32))
    L.GPU_DestroyTexture(again); L.GPU_DestroyTexture(exact)


@pytest.mark.parametrize("mode", ["live_shafts_shadows", "ibl_shadows"])
def test_shade_sun_shadows_vs_oracle(gpu, mode):
    """lighting_pass.glsl:594-608 (4-tap PCF sun shadow) and :646 (light-shaft visibility) with a synthetic sun depth map bound to
    SUN_DEPTH_MAP: K5 vs the oracle, which reproduces the shader text bit for bit on this block (tests/test_oracle_cpu.py)."""
    import pbrhip, pbr_oracle as O
    from pbrhip import synth
    W, H = 256, 144
    gbd, env_tex, maps, gb, lp0, glob = _shade_setup(gpu, W, H, pbrhip.Format_RGBA32F)
    # centre the synthetic map on the scene's depth as the sun sees it, so that lit and shadowed pixels both occur
    M = lambda a: np.array(a, np.float64).reshape(4, 4).T
    ys, xs = np.nonzero(gbd["depth"] < 1)
    ndc = np.stack([(xs + 0.5) / W * 2 - 1, (ys + 0.5) / H * 2 - 1, gbd["depth"][ys, xs], np.ones(len(xs))], -1)
    pw = ndc @ M(glob.world_space_from_clip).T
    ps = np.concatenate([pw[:, :3] / pw[:, 3:], np.ones((len(xs), 1))], 1) @ M(glob.sun_space_from_world).T
    v, u = np.mgrid[0:512, 0:512] / 512.0                           # the whole scene covers ~0.14 of the map: use fine ripples
    sun = (np.median(ps[:, 2]) + 0.08 * np.sin(2 * np.pi * 40 * u) * np.cos(2 * np.pi * 36 * v)).astype(np.float32)
    sun_tex = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, 512, 512, pbrhip.TextureFlag_RenderTarget)
    pbrhip.upload_mip(sun_tex, 0, sun)
    lp = gpu.PBR_MakeLightingPassEx(C.byref(gb), C.byref(maps), W, H, sun_tex)
    gflags, oflags = {"live_shafts_shadows": (pbrhip.Shade_LightShafts | pbrhip.Shade_SunShadows, O.SHADE_SHAFTS | O.SHADE_SHADOWS),
                      "ibl_shadows": (pbrhip.Shade_IBL | pbrhip.Shade_SunShadows, O.SHADE_IBL | O.SHADE_SHADOWS)}[mode]
    g = gpu.GPU_MakeGraph()
    res = {}
    for name, fl in (("shadowed", gflags), ("lit", gflags & ~pbrhip.Shade_SunShadows)):
        gpu.GPUX_SetShadeFlags(gpu.PBR_LightingPipeline(lp), fl)
        gpu.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
        gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
        res[name] = pbrhip.read_mip(gb.lighting_result, 0)
    irr = pbrhip.read_mip(maps.irradiance_map, 0)
    n = maps.tex_specular_env_map.contents.mip_level_count
    pyr = np.concatenate([pbrhip.read_mip(maps.tex_specular_env_map, m).ravel() for m in range(n)])
    og = O.OrcGlobals.from_buffer_copy(bytes(glob))
    want = O.shade(og, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"], flags=oflags, irradiance_cube=irr,
                   prefiltered_pyr=pyr, prefiltered_size=maps.tex_specular_env_map.contents.width,
                   lut_half=pbrhip.read_mip(maps.brdf_lut, 0).view(np.uint16), sun_depth_map=sun)
    got = res["shadowed"]
    assert rmse_rel(got, want) < REL
    assert rel_err(got[..., :3], want[..., :3], floor=1e-2) < REL, rel_err(got[..., :3], want[..., :3], floor=1e-2)
    darker = (got[..., :3].sum(-1) < res["lit"][..., :3].sum(-1) * 0.999) & (gbd["depth"] < 1)
    assert darker.mean() > 0.02, darker.mean()                       # the map really shadows part of the spheres
    # the flag without a usable map is refused at record time
    msgs = []
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    cb = CB(lambda m, u: msgs.append(m.decode()))
    gpu.GPUX_SetErrorHandler(C.cast(cb, C.c_void_p), None)
    try:
        gpu.GPUX_SetShadeFlags(gpu.PBR_LightingPipeline(lp0), pbrhip.Shade_SunShadows)
        gpu.PBR_RecordLightingPass(lp0, g, C.byref(glob), 0, 0)                          # lp0 has the 1x1 stand-in: fine (D32F)
        assert not msgs
        gpu.GPU_GraphSubmit(g); gpu.GPU_GraphWait(g)
    finally:
        gpu.GPUX_SetErrorHandler(None, None)
    gpu.GPU_DestroyGraph(g); gpu.PBR_DestroyLightingPass(lp); gpu.PBR_DestroyLightingPass(lp0); gpu.PBR_DestroyGBuffer(C.byref(gb))
    gpu.PBR_DestroyIBLMaps(C.byref(maps)); gpu.GPU_DestroyTexture(env_tex); gpu.GPU_DestroyTexture(sun_tex)


@pytest.mark.parametrize("size", [(96, 54), (256, 144)])
def test_shade_complete_live_shader_vs_oracle(gpu, size):
    """the reference's complete live lighting shader (light shafts + sun shadows + voxel-GI ambient / specular with
    the screen-space trace) in K5, against the oracle that reproduces the shader text bit for bit (tests/test_oracle_cpu.py).
    Rays take data-dependent exits; every pixel must still agree to 1e-4."""
    import pbrhip, pbr_oracle as O
    from pbrhip import synth
    W, H = size
    L = gpu
    gbd, grid, levels, sun = synth.synth_gi_scene(W, H)
    env = synth.synth_env(64, seed=0x5EED00AA)
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 16, 64, 32)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map); L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 1); L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
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
    glob = pbrhip.fill_globals(synth.GI_SCENE_CAMERA, aspect=W / H, frame_idx=3)
    glob.lightgrid_scale = 1.0 / synth.GI_SCENE_EXTENT
    irr = pbrhip.read_mip(maps.irradiance_map, 0)
    nm = maps.tex_specular_env_map.contents.mip_level_count
    pyr = np.concatenate([pbrhip.read_mip(maps.tex_specular_env_map, m).ravel() for m in range(nm)])
    lut = pbrhip.read_mip(maps.brdf_lut, 0).view(np.uint16)
    og = O.OrcGlobals.from_buffer_copy(bytes(glob))
    g = L.GPU_MakeGraph()
    for gflags, oflags in ((pbrhip.Shade_LightShafts | pbrhip.Shade_SunShadows | pbrhip.Shade_VoxelGI, O.SHADE_SHAFTS | O.SHADE_SHADOWS | O.SHADE_GI),
                           (pbrhip.Shade_IBL | pbrhip.Shade_VoxelGI, O.SHADE_IBL | O.SHADE_GI)):
        L.GPUX_SetShadeFlags(L.PBR_LightingPipeline(lp), gflags)
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
        got = pbrhip.read_mip(gb.lighting_result, 0)
        O.gi_exit_counts()
        want = O.shade(og, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"], flags=oflags, irradiance_cube=irr, prefiltered_pyr=pyr,
                       prefiltered_size=maps.tex_specular_env_map.contents.width, lut_half=lut, sun_depth_map=sun, lightgrid=grid, prev_frame_levels=levels[:nlev])
        exits = O.gi_exit_counts()
        assert min(exits) > 0, exits
        err = np.abs(got[..., :3].astype(np.float64) - want[..., :3]) / np.maximum(np.abs(want[..., :3]), 1e-2)
        bad = (err.max(-1) >= REL)
        assert not bad.any(), (int(bad.sum()), float(err.max()), np.argwhere(bad)[:5].tolist())
    # row bands == full frame (the traces read other pixels' depth, never other pixels' results: no halo of outputs needed)
    full = got.copy()
    L.GPU_OpClearColorF(g, gb.lighting_result, 0, 0.0, 0.0, 0.0, 0.0)
    for r0, r1 in ((0, H // 3), (H // 3, H // 2), (H // 2, H)):
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), r0, r1)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    assert np.array_equal(pbrhip.read_mip(gb.lighting_result, 0).view(np.uint32), full.view(np.uint32))
    L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb)); L.PBR_DestroyIBLMaps(C.byref(maps))
    for t in (env_tex, grid_tex, prev_tex, sun_tex):
        L.GPU_DestroyTexture(t)


def test_device_samplers_at_wild_coordinates(gpu):
    """The 3-D / shadow / post-process samplers at NaN, +-inf, +-1e30, -0.0, exact texel centres and edges: no out-of-bounds read (a
    marching ray can reach such coordinates; an earlier version converted them to int unclamped and faulted) and, wherever the
    result is defined, the same bits as the oracle."""
    import pbrhip, pbr_oracle as O
    L = gpu
    rng = np.random.default_rng(0xC00D)
    special = np.array([0.0, -0.0, 1.0, 0.5, 0.25, 1e-30, -1e-30, 1e30, -1e30, np.inf, -np.inf, np.nan, 3.0e9, -3.0e9, 2.0 ** 31, 1.0 - 2.0 ** -24], np.float32)
    coords = np.concatenate([rng.uniform(-0.5, 1.5, (4096, 3)), rng.choice(special, (4096, 3)),
                             np.stack([(np.arange(64) + 0.5) / 64] * 3, -1)]).astype(np.float32)
    n = len(coords)
    cbuf = L.GPU_MakeBuffer(coords.nbytes, pbrhip.BufferFlag_CPU, coords.ctypes.data_as(C.c_void_p))
    obuf = L.GPU_MakeBuffer(n * 16, pbrhip.BufferFlag_CPU, None)

    def run(which, tex, w, h, d):
        assert L.pbrk_debug_sample(which, L.GPUX_TextureDevicePtr(tex, 0), w, h, d, L.GPUX_BufferDevicePtr(cbuf), n, L.GPUX_BufferDevicePtr(obuf), None) == 0
        L.GPU_WaitUntilIdle()
        return np.frombuffer((C.c_char * (n * 16)).from_address(obuf.contents.data), np.float32).reshape(n, 4).copy()

    grid = (rng.random((16, 16, 16, 4)) * 4).astype(np.float16)
    gtex = pbrhip.make_texture(pbrhip.Format_RGBA16F, 16, 16, pbrhip.TextureFlag_StorageImage, depth=16)
    pbrhip.upload_mip(gtex, 0, grid)
    got = run(0, gtex, 16, 16, 16)
    want = np.zeros_like(got)
    g16 = np.ascontiguousarray(grid.view(np.uint16))
    for i in range(n):
        p = np.ascontiguousarray(coords[i])
        O.lib().orc_tex3d_sample(g16.ctypes.data_as(C.c_void_p), 16, p.ctypes.data_as(C.c_void_p), want[i].ctypes.data_as(C.c_void_p))
    assert np.array_equal(got, want, equal_nan=True)
    depth = rng.random((24, 40)).astype(np.float32)
    dtex = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, 40, 24, pbrhip.TextureFlag_RenderTarget)
    pbrhip.upload_mip(dtex, 0, depth)
    got = run(1, dtex, 40, 24, 1)[:, 0]
    t = O._tex2d(depth, O.TEX_R32F)[0]
    want = np.array([O.lib().orc_shadow_sample(t, float(c[0]), float(c[1]), float(c[2])) for c in coords], np.float32)
    assert np.array_equal(got, want, equal_nan=True)
    img = (rng.random((24, 40, 4)) * 4).astype(np.float16)
    itex = pbrhip.make_texture(pbrhip.Format_RGBA16F, 40, 24, pbrhip.TextureFlag_RenderTarget)
    pbrhip.upload_mip(itex, 0, img)
    got = run(2, itex, 40, 24, 1)
    want = np.stack([O.tex2d_sample(img, O.TEX_RGBA16F, float(c[0]), float(c[1])) for c in coords])
    assert np.array_equal(got, want, equal_nan=True)
    for t_ in (gtex, dtex, itex):
        L.GPU_DestroyTexture(t_)
    L.GPU_DestroyBuffer(cbuf); L.GPU_DestroyBuffer(obuf)
