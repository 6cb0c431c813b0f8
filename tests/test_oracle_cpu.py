"""CPU tests (no GPU): the oracle against the committed Oracle-A fixtures (reference shader text executed
on the CPU, oracle/gen_oracle_a.py) and against the invariants SURVEY.md section 4 lists."""
import json
import os

import numpy as np
import pytest


def rel(a, b, floor=1e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


@pytest.fixture(scope="module")
def O():
    import pbr_oracle
    pbr_oracle.lib()
    return pbr_oracle


@pytest.fixture(scope="module")
def meta(golden_dir):
    with open(os.path.join(golden_dir, "oracle_a_meta.json")) as f:
        return json.load(f)


# ---- SURVEY 8c known answers (typed in from SURVEY.md, independent of the fixture files) -----------
SURVEY_LUT = {(0, 0): (0, 0), (255, 255): (0.455724776, 6.05484565e-05), (128, 128): (0.867852032, 0.0186930094),
              (10, 200): (0.784716249, 0.203443512), (200, 10): (0.999417484, 0.00051532808), (255, 12): (1.02073967, 8.80392287e-12)}
SURVEY_PREFILTER = {(0, 0, 0, 0): (1.28942835, 1.16623127, 1.16688311, 1), (1, 0, 0, 0): (1.69529569, 1.53246474, 1.53319418, 1.31460953),
                    (1, 64, 17, 3): (1.31848633, 1.74375546, 1.31687999, 1.31460953), (2, 63, 0, 5): (0.884181321, 1.42153156, 1.40017962, 1.22008407),
                    (3, 5, 20, 2): (0.751303554, 1.19718492, 0.927152216, 0.957092762), (4, 15, 15, 4): (0.854548454, 0.827704906, 0.751518488, 0.710959077),
                    (4, 8, 3, 1): (0.48960039, 0.818277121, 0.705310225, 0.710959077)}
SURVEY_IRRADIANCE = {(0, 0, 0): (0.598888576, 0.583451807, 0.521617532, 0), (31, 31, 5): (0.405151099, 0.583455205, 0.521616101, 0),
                     (16, 7, 2): (0.50508374, 0.611887932, 0.499680549, 0), (3, 28, 4): (0.412977099, 0.580234051, 0.478413224, 0)}


def test_lut_matches_oracle_a_and_survey(O, golden_dir):
    want = np.load(os.path.join(golden_dir, "oracle_a_lut256.npy"))
    got = O.brdf_lut(256, 4096)
    assert np.array_equal(got, want)                     # same libm, same op order: bit for bit
    for (x, y), v in SURVEY_LUT.items():
        assert np.allclose(got[y, x], v, rtol=1e-5, atol=1e-12)
    assert abs(got[..., 0].astype(np.float64).sum() - 54016.6149) < 1e-3
    assert abs(got[..., 1].astype(np.float64).sum() - 4991.87747) < 1e-3
    assert not np.isnan(got).any() and (got[..., 0] == 0).sum() == 37 and (got[..., 0] > 1).sum() == 1951


def test_prefilter_analytic_env(O, golden_dir, meta):
    for mip in (3, 4):
        want = np.load(os.path.join(golden_dir, f"oracle_a_prefilter_analytic_mip{mip}.npy"))
        assert rel(O.prefilter_mip(None, 1, 256, mip), want) <= 1e-6
    for (mip, x, y, f), v in SURVEY_PREFILTER.items():
        got = O.prefilter_mip(None, 1, 256, mip, faces=(f, f + 1), rows=(y, y + 1))[f, y, x]
        assert np.allclose(got, v, rtol=1e-5), (mip, x, y, f)
    for k in meta["prefilter_analytic_kats"]["texels"]:
        got = O.prefilter_mip(None, 1, 256, k["mip"], faces=(k["face"], k["face"] + 1), rows=(k["y"], k["y"] + 1))[k["face"], k["y"], k["x"]]
        assert rel(got, k["rgba"]) <= 1e-6


def test_prefilter_alpha_is_texel_independent_weight_sum(O):
    """alpha = sum of the weights; values 1.3146, 1.2201, 0.9571, 0.7110 for mips 1-4."""
    # fp32 sequential sums as the shader forms them (SURVEY 8c KATs); the exact sums are 1.3146100, 1.2200896, ...
    for mip, a in zip((1, 2, 3, 4), (1.31460953, 1.22008407, 0.957092762, 0.710959077)):
        out = O.prefilter_mip(None, 1, 256, mip, faces=(2, 3), rows=(5, 7))[2, 5:7]
        assert np.all(out[..., 3] == out[0, 0, 3])
        assert abs(out[0, 0, 3] - a) < 2e-7


def test_prefilter_literal_equals_hoisted(O, golden_dir):
    """Hoisting the per-sample transcendentals out of the texel loop changes no bits."""
    from pbrhip import synth
    pyr = O.build_pyramid(synth.synth_env(64, seed=0x5EED00AA))
    a = O.prefilter_mip(pyr, 64, 64, 2, faces=(1, 2), rows=(3, 5), literal=True)
    b = O.prefilter_mip(pyr, 64, 64, 2, faces=(1, 2), rows=(3, 5), literal=False)
    assert np.array_equal(a[1, 3:5], b[1, 3:5])


def test_prefilter_textured_env(O, golden_dir):
    from pbrhip import synth
    pyr = O.build_pyramid(synth.synth_env(64, seed=0x5EED00AA))
    for mip in (0, 1, 2):
        want = np.load(os.path.join(golden_dir, f"oracle_a_prefilter_env64_out64_mip{mip}.npy"))
        assert rel(O.prefilter_mip(pyr, 64, 64, mip), want) <= 1e-6


def test_irradiance(O, golden_dir):
    from pbrhip import synth
    want = np.load(os.path.join(golden_dir, "oracle_a_irradiance_analytic.npy"))
    got = O.irradiance(None, 1, 32)
    assert rel(got, want) <= 1e-6
    for (x, y, f), v in SURVEY_IRRADIANCE.items():
        assert np.allclose(got[f, y, x], v, rtol=1e-5, atol=1e-7)
    pyr = O.build_pyramid(synth.synth_env(256, seed=0x5EED00AB))
    want = np.load(os.path.join(golden_dir, "oracle_a_irradiance_env256.npy"))
    assert rel(O.irradiance(pyr, 256, 32), want) <= 1e-6


def test_constant_environment_invariants(O):
    """irradiance of a constant env c is c*(N+1)/(2N); prefilter of a constant env is c*alpha."""
    c = np.array([0.25, 2.0, 7.5, 1.0], np.float32)
    env = np.broadcast_to(c, (6, 16, 16, 4)).copy()
    pyr = O.build_pyramid(env)
    irr = O.irradiance(pyr, 16, 8, src_lod=2.0)
    assert np.allclose(irr[..., :3], c[:3] * (1025.0 / 2048.0), rtol=2e-5)
    pre = O.prefilter_mip(pyr, 16, 32, 3, src_lod=1.0, faces=(0, 1), rows=(0, 1))[0, 0]
    assert np.allclose(pre[:, :3], c[:3] * pre[0, 3], rtol=2e-5)


def test_lighting_kats_and_tile(O, golden_dir, meta):
    """Shade oracle vs the reference lighting_pass.glsl executed on the CPU (live, live+shafts, IBL-uncommented)."""
    g = O.OrcGlobals.from_buffer_copy(np.load(os.path.join(golden_dir, "ref_globals_default.npy")).tobytes()[:552])
    tile = np.load(os.path.join(golden_dir, "oracle_a_lighting_tile_inputs.npy"))
    W, H = 1920, 1080
    variants = (("live_noshaft", O.SHADE_ANALYTIC), ("live_shaft", O.SHADE_ANALYTIC | O.SHADE_SHAFTS),
                ("ibl", O.SHADE_ANALYTIC | O.SHADE_IBL))

    def run(pixels, flags, sun=None):
        base = np.zeros((H, W, 4), np.uint8); nrm = base.copy(); orm = base.copy(); emi = base.copy()
        dep = np.ones((H, W), np.float32)
        res = np.zeros((len(pixels), 4), np.float32)
        for k, (x, y, b, n, o, e, d) in enumerate(pixels):
            base[y, x] = b; nrm[y, x] = n; orm[y, x] = o; emi[y, x] = e; dep[y, x] = d
            res[k] = O.shade(g, base, nrm, orm, emi, dep, flags=flags, region=(x, x + 1, y, y + 1), sun_depth_map=sun)[y, x]
        return res

    kat = meta["lighting_kats"]
    px = []
    for p, a in zip(kat["pixels"], kat["alpha_bytes"]):
        px.append((p["x"], p["y"], p["base"] + [a], p["normal"] + [a], p["orm"] + [a], p["emissive"] + [a], p["depth"]))
    for name, flags in variants:
        want = np.array(kat["results"][name], np.float32)
        assert rel(run(px, flags)[:, :3], want[:, :3], floor=1e-3) <= 1e-6, name
    sub = tile[::9]          # 256 of the 2304 tile pixels keep the CPU suite fast
    pixels = [(int(t["x"]), int(t["y"]), t["base"], t["nrm"], t["orm"], t["emi"], float(t["depth"])) for t in sub]
    for name, flags in variants:
        want = np.load(os.path.join(golden_dir, f"oracle_a_lighting_tile_{name}.npy"))[::9]
        assert rel(run(pixels, flags)[:, :3], want[:, :3], floor=1e-3) <= 1e-6, name
    # the sun-shadow block + shaft visibility (lighting_pass.glsl:594-608, 646) against a synthetic sun depth map
    from pbrhip import synth
    sun = synth.synth_sun_depth(256, 0x5EED00E0)
    want = np.load(os.path.join(golden_dir, "oracle_a_lighting_tile_live_shadow.npy"))[::9]
    got = run(pixels, O.SHADE_ANALYTIC | O.SHADE_SHAFTS | O.SHADE_SHADOWS, sun)
    assert np.array_equal(got[:, :3].view(np.uint32), want[:, :3].view(np.uint32))
    lit = np.load(os.path.join(golden_dir, "oracle_a_lighting_tile_live_shaft.npy"))[::9]
    assert (want[:, :3].sum(1) < lit[:, :3].sum(1) - 1e-6).sum() > 40          # a good share of the pixels is in shadow
    t = O._tex2d(sun, O.TEX_R32F)[0]
    assert O.lib().orc_shadow_sample(t, 0.3, 0.3, -1.0) == 1.0 and O.lib().orc_shadow_sample(t, 0.3, 0.3, 2.0) == 0.0


def test_cube_neighbor_geometry(O):
    """The integer edge fold agrees with folding the cube in 3-D with floats, for every edge texel."""
    import ctypes as C
    L = O.lib()

    def face_point(f, sc, tc):
        return [(1, -tc, -sc), (-1, -tc, sc), (sc, 1, tc), (sc, -1, -tc), (sc, -tc, 1), (-sc, -tc, -1)][f]

    n = 8
    for f in range(6):
        for k in range(n):
            for (i, j) in ((-1, k), (n, k), (k, -1), (k, n)):
                ni, nj = C.c_int(), C.c_int()
                nf = L.orc_cube_neighbor(f, n, i, j, C.byref(ni), C.byref(nj))
                # 3-D: the outside texel centre, moved slightly inside the cube along the face normal
                sc, tc = (2 * i + 1 - n) / n, (2 * j + 1 - n) / n
                p = np.array(face_point(f, sc, tc), float)
                major = f // 2
                over = int(np.argmax(np.abs(np.where(np.arange(3) == major, 0, p))))
                q = p.copy(); q[over] = np.sign(p[over]); q[major] = np.sign(p[major]) * (1 - 1.0 / n)
                want_face = over * 2 + (1 if q[over] < 0 else 0)
                assert nf == want_face
                d = q / np.linalg.norm(q)
                out = np.zeros(3, np.float32)
                L.orc_face_dir(nf, C.c_float((ni.value + 0.5) / n), C.c_float((nj.value + 0.5) / n), out)
                assert np.allclose(out, d, atol=1e-6)


def test_cube_sampler_is_seamless_and_exact_at_centres(O):
    rng = np.random.default_rng(7)
    env = rng.random((6, 8, 8, 4)).astype(np.float32)
    pyr = O.build_pyramid(env)
    # texel centres return the texel
    from pbrhip import synth
    d = synth.face_dirs(8).astype(np.float32)
    for f in range(6):
        got = O.cube_sample(pyr, 8, d[f].reshape(-1, 3), 0.0).reshape(8, 8, 4)
        assert np.allclose(got, env[f], rtol=0, atol=3e-6)     # fp32 directions: weights are within ~1e-6 of (1, 0)
    # continuity across every face edge: two directions a hair apart on either side of an edge
    eps = 1e-4
    for axis in range(3):
        for other in range(3):
            if other == axis:
                continue
            for s1 in (-1, 1):
                for s2 in (-1, 1):
                    for t in np.linspace(-0.9, 0.9, 7):
                        third = 3 - axis - other
                        a = np.zeros(3); b = np.zeros(3)
                        a[axis], a[other], a[third] = s1, s2 * (1 - eps), t
                        b[axis], b[other], b[third] = s1 * (1 - eps), s2, t
                        va = O.cube_sample(pyr, 8, a[None], 0.0)
                        vb = O.cube_sample(pyr, 8, b[None], 0.0)
                        assert np.allclose(va, vb, atol=5e-3), (axis, other, s1, s2, t)


def test_pyramid_and_mip_count(O):
    assert [O.mip_count(w) for w in (1, 2, 3, 256, 1024, 2048)] == [1, 2, 2, 9, 11, 12]      # gpu_vulkan.c:1344-1351
    env = np.arange(6 * 4 * 4 * 4, dtype=np.float32).reshape(6, 4, 4, 4)
    pyr = O.build_pyramid(env)
    l1 = O.pyramid_level(pyr, 4, 1)
    want = env.reshape(6, 2, 2, 2, 2, 4).transpose(0, 1, 3, 2, 4, 5).reshape(6, 2, 2, 4, 4).mean(axis=3)
    assert np.array_equal(l1, want.astype(np.float32))
    assert O.pyramid_level(pyr, 4, 2).shape == (6, 1, 1, 4)


def test_rgbe_decode_flat_rle_and_against_reference_stb(O, golden_dir):
    from pbrhip import synth
    rng = np.random.default_rng(3)
    rgbe = rng.integers(0, 256, (12, 16, 4), dtype=np.uint8)
    rgbe[2, :, :] = rgbe[2, 0, :]            # a constant row exercises the run encoder
    rgbe[5, 3:9, 3] = 0                      # zero exponent -> black
    want = synth.rgbe_decode(rgbe)
    for rle in (False, True):
        got = O.rgbe_decode(synth.hdr_file_bytes(rgbe, rle=rle))
        assert np.array_equal(got, want)
    narrow = rgbe[:, :4].copy()              # width < 8 is always flat (stb_image.h:7216)
    assert np.array_equal(O.rgbe_decode(synth.hdr_file_bytes(narrow, rle=True)), synth.rgbe_decode(narrow))
    assert np.allclose(O.rgbe_decode(synth.hdr_file_bytes(np.array([[[128, 64, 32, 129]]], np.uint8), rle=False))[0, 0], (1, .5, .25, 1))
    fix = os.path.join(golden_dir, "ref_stb_hdr_decode.npz")
    if os.path.exists(fix):                  # decoded by the reference's own stb_image.h (oracle/gen_golden.py)
        z = np.load(fix)
        assert np.array_equal(O.rgbe_decode(z["file_rle"].tobytes()), z["decoded"])
        assert np.array_equal(O.rgbe_decode(z["file_flat"].tobytes()), z["decoded"])


def test_f16_conversion(O):
    vals = np.array([0, 1, -1, 0.5, 65504, 65519.99, 65520, 1e-8, 6e-8, 6.1e-5, 3.14159, 1 / 3, 2049, 2051, -2.5e-7], np.float32)
    L = O.lib()
    for v in vals:
        assert L.orc_f32_to_f16(float(v)) == int(np.float32(v).astype(np.float16).view(np.uint16)), v
    for h in (0, 1, 0x3FF, 0x400, 0x3C00, 0x7BFF, 0x8001, 0xFC00):
        assert np.float32(L.orc_f16_to_f32(h)) == np.uint16(h).view(np.float16).astype(np.float32)


def test_full_live_lighting_with_voxel_gi(O, golden_dir, meta):
    """lighting_pass.glsl with light shafts, sun shadows and SampleRadianceWithScreenSpaceTrace live, executed as
    shader text on the CPU (oracle/gen_oracle_a.py --only gi), against the oracle: every pixel bit for bit, every exit of
    the trace exercised."""
    from pbrhip import synth
    m = meta["lighting_full_gi"]
    W, H = m["width"], m["height"]
    want = np.load(os.path.join(golden_dir, m["file"]))
    g = O.OrcGlobals.from_buffer_copy(np.load(os.path.join(golden_dir, m["globals"])).tobytes()[:552])
    assert abs(g.lightgrid_scale - 1.0 / synth.GI_SCENE_EXTENT) < 1e-9
    gbd, grid, levels, sun = synth.synth_gi_scene(W, H)
    O.gi_exit_counts()
    got = O.shade(g, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"],
                  flags=O.SHADE_ANALYTIC | O.SHADE_SHAFTS | O.SHADE_SHADOWS | O.SHADE_GI, sun_depth_map=sun, lightgrid=grid, prev_frame_levels=levels)
    exits = O.gi_exit_counts()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert min(exits) > 100 and sum(exits) == 2 * int((gbd["depth"] < 1).sum()), exits      # two traces per surface pixel
    # the GI terms matter: without them the frame is different on most surface pixels
    plain = O.shade(g, gbd["base"], gbd["normal"], gbd["orm"], gbd["emissive"], gbd["depth"],
                    flags=O.SHADE_ANALYTIC | O.SHADE_SHAFTS | O.SHADE_SHADOWS, sun_depth_map=sun)
    surf = gbd["depth"] < 1
    assert (np.abs(got - plain)[surf].max(-1) > 1e-3).mean() > 0.8
    # deterministic trig: accurate to ~1e-7 on the ranges the shader uses
    L = O.lib()
    xs = np.linspace(0, 2 * np.pi, 4001).astype(np.float32)
    assert max(abs(L.orc_sinf_det(float(x)) - np.sin(np.float64(x))) for x in xs) < 2e-7
    assert max(abs(L.orc_cosf_det(float(x)) - np.cos(np.float64(x))) for x in xs) < 2e-7
    xa = np.linspace(0, 1, 2001).astype(np.float32)
    assert max(abs(L.orc_acosf_det(float(x)) - np.arccos(np.float64(x))) for x in xa) < 3e-7


def test_linear_blit_rule_for_levels_that_are_not_2_to_1():
    """oracle/pbr_oracle.c A2: mip levels of faces that are not a power of two (125 -> 62, 3 -> 1) are genuine linear resamples
    (vkCmdBlitImage: u = (x + .5) ns / nd, taps floor(u - .5) and + 1 clamped to the edge, weights in fp32).  The C loop against a
    numpy restatement of the same operations in float32 (bit for bit), the level sizes of the reference (gpu_vulkan.c:1344-1351),
    and the properties the rule must have: constants are preserved, 3 -> 1 returns the centre texel, an exact 2:1 resample agrees
    with the 2x2 box to an ulp."""
    import pbr_oracle as O
    rng = np.random.default_rng(0x5EED00C1)
    f32 = np.float32
    for (ns, nd) in ((125, 62), (31, 15), (7, 3), (3, 1), (10, 4)):
        src = (rng.random((2, ns, ns, 4), dtype=np.float32) * 50.0).astype(np.float32)
        got = O.blit_linear(src, nd, nd)
        sc = f32(ns) / f32(nd)
        t = (np.arange(nd, dtype=np.float32) + f32(0.5)) * sc - f32(0.5)
        fl = np.floor(t)
        a = (t - fl).astype(np.float32)
        i0 = np.clip(fl.astype(np.int64), 0, ns - 1); i1 = np.clip(fl.astype(np.int64) + 1, 0, ns - 1)
        ax = a[None, None, :, None]; ay = a[None, :, None, None]
        one = f32(1.0)
        top = src[:, i0][:, :, i0] * (one - ax) + src[:, i0][:, :, i1] * ax
        bot = src[:, i1][:, :, i0] * (one - ax) + src[:, i1][:, :, i1] * ax
        want = (top * (one - ay) + bot * ay).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (ns, nd)
    c = np.full((1, 9, 9, 4), 3.25, np.float32)
    assert (O.blit_linear(c, 4, 4) == 3.25).all()
    s3 = rng.random((1, 3, 3, 4), dtype=np.float32)
    assert np.array_equal(O.blit_linear(s3, 1, 1)[0, 0, 0], s3[0, 1, 1])
    s8 = rng.random((1, 8, 8, 4), dtype=np.float32)
    box = s8.reshape(1, 4, 2, 4, 2, 4).astype(np.float64).mean(axis=(2, 4))
    assert np.abs(O.blit_linear(s8, 4, 4) - box).max() <= 2.0 ** -23
    for W in (96, 1000, 125, 1536):
        n = O.mip_count(W)
        assert n == 1 + int(np.floor(np.log2(W)))
    env = rng.random((6, 12, 12, 4), dtype=np.float32)              # 12, 6, 3, 1: box, box, resample
    pyr = O.build_pyramid(env)
    l2 = O.pyramid_level(pyr, 12, 2); l3 = O.pyramid_level(pyr, 12, 3)
    assert l3.shape == (6, 1, 1, 4) and np.array_equal(l3[:, 0, 0], l2[:, 1, 1])
