"""Post-process tail: taa_resolve.glsl and final_post_process.glsl.  The oracle against the Oracle-A
fixture (the shader text executed on the CPU, oracle/gen_oracle_a.py --only post) and the HIP kernels K8 / K9 against
both, through the GPU_* boundary.  K8 is pure fp32 arithmetic in the shader's order: bit-exact.  K9 ends in pow():
1e-5 relative on float targets, at most one code on 8-bit targets."""
import ctypes as C
import os

import numpy as np
import pytest

import pbr_oracle as O


@pytest.fixture(scope="module")
def post_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "oracle_a_post.npz"))
    return {k: z[k] for k in z.files}


def test_oracle_matches_reference_shaders(post_fixture):
    f = post_fixture
    taa = O.taa_resolve(f["lighting"], f["depth"], f["velocity"], f["velocity_prev"], f["history"])
    assert np.array_equal(taa.view(np.uint32), f["taa"].view(np.uint32))
    resolved = f["taa"].astype(np.float16)
    h, w = resolved.shape[:2]
    assert np.array_equal(O.final_post_process(resolved).view(np.uint32), f["final_same"].view(np.uint32))
    assert np.array_equal(O.final_post_process(resolved, 2 * w, 2 * h).view(np.uint32), f["final_up"].view(np.uint32))
    # the fixture exercises both rejection paths and the negative Mitchell-Netravali lobes
    lit = f["lighting"].view(np.float16).astype(np.float32)
    assert (f["taa"][..., :3] < 0).any() and np.isfinite(f["taa"]).all()
    assert np.abs(f["taa"][:, :4, :3] - lit[:, :4, :3]).max() < 400          # off-screen strip: result = source_sample only


def test_post_sampler_definition():
    rng = np.random.default_rng(5)
    tex = rng.random((9, 13, 4)).astype(np.float16)
    h, w = tex.shape[:2]
    t32 = tex.astype(np.float32)
    for (i, j) in [(0, 0), (12, 8), (5, 3), (7, 7)]:            # centre taps are exact texel fetches
        got = O.tex2d_sample(tex, O.TEX_RGBA16F, (i + 0.5) / w, (j + 0.5) / h)
        assert np.array_equal(got, t32[j, i])
    # the same at a large extent, where fp32 coordinate error would otherwise bleed a neighbour in
    big = np.zeros((2, 8192, 4), np.float16)
    big[:, 8000] = [60000, 0, 0, 1]
    got = O.tex2d_sample(big, O.TEX_RGBA16F, np.float32(7999.5) * np.float32(1.0 / 8192), 0.25)
    assert got[0] == 0.0
    assert np.array_equal(O.tex2d_sample(tex, O.TEX_RGBA16F, -3.0, 0.5 / h), t32[0, 0])       # clamp
    assert np.array_equal(O.tex2d_sample(tex, O.TEX_RGBA16F, 7.0, 2.0), t32[h - 1, w - 1])
    mid = O.tex2d_sample(tex, O.TEX_RGBA16F, 6.0 / w, 3.5 / h)                                # half way between texels 5 and 6
    assert np.allclose(mid, t32[3, 5] + np.float32(0.5) * (t32[3, 6] - t32[3, 5]), rtol=0, atol=0)


def test_taa_static_frame_is_a_fixed_point():
    # constant colour, zero velocity, history == frame: every filter weight set sums to one -> output == input
    h, w = 12, 20
    frame = np.zeros((h, w, 4), np.float16); frame[..., :3] = [0.5, 1.25, 2.0]; frame[..., 3] = 1
    vel = np.zeros((h, w, 2), np.float16)
    depth = np.full((h, w), 0.999, np.float32)
    out = O.taa_resolve(frame, depth, vel, vel, frame)
    assert np.allclose(out[..., :3], [0.5, 1.25, 2.0], rtol=1e-6) and np.all(out[..., 3] == 1)
    assert np.array_equal(O.unorm8(np.array([0.0, 0.5, 1.0, 2.0, -1.0, 0.5 / 255, 1.5 / 255])), [0, 128, 255, 255, 0, 0, 2])   # ties to even


# ------------------------------------------------------------------------------------------- GPU

def _upload(tex, arr):
    import pbrhip
    pbrhip.upload_mip(tex, 0, arr)


def _tex2d(pbrhip, L, tex, fmt):
    t = tex.contents
    return pbrhip.PbrkTex2D(L.GPUX_TextureDevicePtr(tex, 0), fmt, t.width, t.height)


@pytest.mark.gpu
def test_gpu_taa_and_final_match_reference_shaders(gpu, post_fixture):
    import pbrhip
    L, f = gpu, post_fixture
    h, w = f["depth"].shape
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), w, h, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), w, h, pbrhip.Format_RGBA8UN)
    assert pp
    _upload(gb.lighting_result, f["lighting"]); _upload(gb.depth, f["depth"])
    frame_idx = 0
    _upload(L.PBR_PostVelocity(pp, 0), f["velocity"]); _upload(L.PBR_PostVelocity(pp, 1), f["velocity_prev"])
    _upload(L.PBR_PostTaaOutput(pp, 1), f["history"])
    L.GPUX_EnableOpTiming(1)
    g = L.GPU_MakeGraph()
    L.PBR_RecordTaaResolve(pp, g, frame_idx)                                     # render.cpp:1131-1137
    L.PBR_RecordFinalPostProcess(pp, g, frame_idx)                               # render.cpp:1181-1187
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    names = [L.GPUX_GraphTimedOpName(g, i).decode() for i in range(L.GPUX_GraphTimedOpCount(g))]
    assert names == ["K8.taa_resolve", "K9.final_post_process"]
    L.GPU_DestroyGraph(g)
    taa = pbrhip.read_mip(L.PBR_PostTaaOutput(pp, 0), 0)
    want = f["taa"].astype(np.float16)
    assert np.array_equal(taa.view(np.uint16), want.view(np.uint16)), "TAA RGBA16F bits"
    bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
    want8 = O.unorm8(f["final_same"])
    diff = np.abs(bb.astype(np.int32) - want8.astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.995, (diff.max(), (diff == 0).mean())
    assert np.all(bb[..., 3] == 255)

    # raw kernels on float targets: K8 bit-exact in fp32, K9 within pow() accuracy; also a 2x upscaling final pass
    out32 = pbrhip.make_texture(pbrhip.Format_RGBA32F, w, h, pbrhip.TextureFlag_RenderTarget)
    a = pbrhip.PbrkTaaArgs(_tex2d(pbrhip, L, gb.lighting_result, pbrhip.PBRK_FMT_RGBA16F), _tex2d(pbrhip, L, gb.depth, pbrhip.PBRK_FMT_R32F),
                           _tex2d(pbrhip, L, L.PBR_PostVelocity(pp, 0), pbrhip.PBRK_FMT_RG16F), _tex2d(pbrhip, L, L.PBR_PostVelocity(pp, 1), pbrhip.PBRK_FMT_RG16F),
                           _tex2d(pbrhip, L, L.PBR_PostTaaOutput(pp, 1), pbrhip.PBRK_FMT_RGBA16F), L.GPUX_TextureDevicePtr(out32, 0),
                           pbrhip.PBRK_FMT_RGBA32F, w, h, 0, h)
    assert L.pbrk_taa_resolve(C.byref(a), None) == 0
    L.GPU_WaitUntilIdle()
    got = pbrhip.read_mip(out32, 0)
    assert np.array_equal(got.view(np.uint32), f["taa"].view(np.uint32)), "TAA fp32 bits"
    up = pbrhip.make_texture(pbrhip.Format_RGBA32F, 2 * w, 2 * h, pbrhip.TextureFlag_RenderTarget)
    for target, key, (ow, oh) in ((out32, "final_same", (w, h)), (up, "final_up", (2 * w, 2 * h))):
        fa = pbrhip.PbrkFinalArgs(_tex2d(pbrhip, L, L.PBR_PostTaaOutput(pp, 0), pbrhip.PBRK_FMT_RGBA16F), L.GPUX_TextureDevicePtr(target, 0),
                                  pbrhip.PBRK_FMT_RGBA32F, ow, oh, 0, oh)
        assert L.pbrk_final_post_process(C.byref(fa), None) == 0
        L.GPU_WaitUntilIdle()
        got = pbrhip.read_mip(target, 0)
        rel = np.abs(got - f[key]) / np.maximum(f[key], 1e-3)
        assert rel.max() < 1e-5, (key, rel.max())
    # argument checks of the raw entry points (no launch on bad shapes)
    a.y1 = h + 1; assert L.pbrk_taa_resolve(C.byref(a), None) == -1
    a.y1 = h; a.out_format = pbrhip.PBRK_FMT_RGBA8UN; assert L.pbrk_taa_resolve(C.byref(a), None) == -2
    a.out_format = pbrhip.PBRK_FMT_RGBA32F; a.gbuffer_depth.width = w - 1; assert L.pbrk_taa_resolve(C.byref(a), None) == -1
    L.GPU_DestroyTexture(out32); L.GPU_DestroyTexture(up)
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
def test_gpu_post_full_frame_sequence(gpu):
    """1920x1080: three frames of lighting -> TAA (ping-pong history) -> tone map, checked against the oracle on bands of rows
    each frame (bit-exact RGBA16F), plus row-sharded draws == full draws."""
    import pbrhip
    from pbrhip.synth import synth_post_inputs as post_inputs
    L = gpu
    W, H = 1920, 1080
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    history = np.zeros((H, W, 4), np.float16)                                    # frame 0 reads the zero-initialised taa_output_rt[1]
    vel_prev = np.zeros((H, W, 2), np.float16)
    bands = [(0, 6), (537, 543), (1074, 1080)]
    for frame in range(3):
        lighting, depth, vel, _, _ = post_inputs(0x5EED00D0 + frame, W, H)
        _upload(gb.lighting_result, lighting); _upload(gb.depth, depth)
        _upload(L.PBR_PostVelocity(pp, frame % 2), vel)
        g = L.GPU_MakeGraph()
        L.PBR_RecordTaaResolve(pp, g, frame)
        L.PBR_RecordFinalPostProcess(pp, g, frame)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        got = pbrhip.read_mip(L.PBR_PostTaaOutput(pp, frame % 2), 0)
        assert np.isfinite(got.astype(np.float32)).all()
        for (y0, y1) in bands:
            want = O.taa_resolve(lighting, depth, vel, vel_prev, history, rows=(y0, y1))[y0:y1].astype(np.float16)
            assert np.array_equal(got[y0:y1].view(np.uint16), want.view(np.uint16)), f"frame {frame} rows {y0}:{y1}"
        bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
        y0, y1 = bands[1]
        want8 = O.unorm8(O.final_post_process(got)[y0:y1])[..., [2, 1, 0, 3]]    # BGRA byte order
        assert np.abs(bb[y0:y1].astype(np.int32) - want8.astype(np.int32)).max() <= 1
        history, vel_prev = got, vel
    # row-sharded resolve (screen bands) == the full draw of the last frame
    frame = 2
    full = got
    scratch = np.zeros_like(full)
    _upload(L.PBR_PostTaaOutput(pp, frame % 2), scratch)
    g = L.GPU_MakeGraph()
    for (r0, r1) in ((0, 400), (400, 401), (401, H)):
        # the reference's own call sequence with the draw replaced by its row-restricted form
        L.PBR_RecordTaaResolveRows(pp, g, frame, r0, r1)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
    assert np.array_equal(pbrhip.read_mip(L.PBR_PostTaaOutput(pp, frame % 2), 0).view(np.uint16), full.view(np.uint16))
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
def test_gpu_post_boundary_errors(gpu):
    import pbrhip
    L = gpu
    msgs = []
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    cb = CB(lambda m, u: msgs.append(m.decode()))
    L.GPUX_SetErrorHandler(C.cast(cb, C.c_void_p), None)
    try:
        W, H = 64, 32
        gb = pbrhip.PBR_GBuffer()
        L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA32F)              # lighting result in the wrong format for the TAA pass
        pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_RGBA8UN)
        assert pp and not msgs
        g = L.GPU_MakeGraph()
        n = len(msgs); L.PBR_RecordTaaResolve(pp, g, 0); assert len(msgs) == n + 1 and "LIGHTING_RESULT" in msgs[-1]
        n = len(msgs); L.PBR_RecordFinalPostProcess(pp, g, 0); assert len(msgs) == n        # the final pass does not read it
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))
    finally:
        L.GPUX_SetErrorHandler(None, None)


# ------------------------------------------------------------------------------------------- bloom chain

@pytest.fixture(scope="module")
def bloom_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "oracle_a_bloom.npz"))
    return {k: z[k] for k in z.files}


def test_oracle_bloom_chain_matches_reference_shaders(bloom_fixture):
    f = bloom_fixture
    down, up = O.bloom_chain(f["taa"], 6)
    for m in range(6):
        assert np.array_equal(down[m].view(np.uint16), f[f"down{m}"]), f"down{m}"
        assert np.array_equal(up[m].view(np.uint16), f[f"up{m}"]), f"up{m}"
    taa = f["taa"].view(np.float16).astype(np.float32)
    glow = f["up0"].view(np.float16).astype(np.float32)[..., :3] - taa[..., :3]
    assert glow.min() > -1e-2 and glow.mean() > 0                   # level 0 = the frame + 6 % of the blurred pyramid
    assert f["down0"].view(np.float16).astype(np.float32)[..., :3].max() <= 1.0     # firefly clamp of the first downsample


def _check_bloom(L, pbrhip, pp, want_down, want_up):
    n = L.PBR_PostBloomPassCount(pp)
    assert n == len(want_down)
    for m in range(n):
        got = pbrhip.read_mip(L.PBR_PostBloomDownscale(pp), m)
        assert np.array_equal(got.view(np.uint16), np.asarray(want_down[m]).view(np.uint16)), f"bloom_downscale_rt mip {m}"
        got = pbrhip.read_mip(L.PBR_PostBloomUpscale(pp), m)
        assert np.array_equal(got.view(np.uint16), np.asarray(want_up[m]).view(np.uint16)), f"bloom_upscale_rt mip {m}"


@pytest.mark.gpu
def test_gpu_bloom_chain_matches_reference_shaders(gpu, bloom_fixture):
    """render.cpp:1139-1187 through the boundary: 6 downsamples, clear, blit, 6 additive upsamples, final pass reading the
    bloom target; all twelve RGBA16F targets bit-exact against the shader text, the 8-bit frame within one code."""
    import pbrhip
    L, f = gpu, bloom_fixture
    h, w = f["taa"].shape[:2]
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), w, h, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), w, h, pbrhip.Format_RGBA8UN)
    frame_idx = 1
    _upload(L.PBR_PostTaaOutput(pp, frame_idx % 2), f["taa"])
    L.GPUX_EnableOpTiming(1)
    g = L.GPU_MakeGraph()
    L.PBR_RecordBloom(pp, g, frame_idx)
    L.PBR_RecordFinalPostProcessBloom(pp, g, frame_idx)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    names = [L.GPUX_GraphTimedOpName(g, i).decode() for i in range(L.GPUX_GraphTimedOpCount(g))]
    # the 1:1 blit of the TAA result into bloom_upscale_rt (render.cpp:1158-1163) is folded into the last upsample, which adds onto the
    # blit's source instead of onto a copy of it (gpu_hip.cpp fold_blits): no copy in the op list, the same bits in every target (below)
    assert names == ["K10.bloom_downsample"] * 6 + ["K11.bloom_upsample"] * 6 + ["K9.final_post_process"], names
    assert L.GPUX_FoldedBlitCount() >= 1
    L.GPU_DestroyGraph(g)
    _check_bloom(L, pbrhip, pp, [f[f"down{m}"] for m in range(6)], [f[f"up{m}"] for m in range(6)])
    bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
    want8 = O.unorm8(O.final_post_process(f["up0"]))
    assert np.abs(bb.astype(np.int32) - want8.astype(np.int32)).max() <= 1
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
def test_gpu_bloom_full_frame_and_errors(gpu):
    """1920x1080 (odd mip sizes: 135 -> 67 -> 33 rows): bit-exact against the oracle chain; a second run gives the same
    bits (the clear restarts the additive targets); blending other than the bloom's is rejected."""
    import pbrhip
    from pbrhip.synth import synth_post_inputs
    L = gpu
    W, H = 1920, 1080
    taa, _, _, _, _ = synth_post_inputs(0x5EED00D7, W, H)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    _upload(L.PBR_PostTaaOutput(pp, 0), taa)
    want_down, want_up = O.bloom_chain(taa, 6)
    for _ in range(2):
        g = L.GPU_MakeGraph()
        L.PBR_RecordBloom(pp, g, 0)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        _check_bloom(L, pbrhip, pp, want_down, want_up)
    msgs = []
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    cb = CB(lambda m, u: msgs.append(m.decode()))
    L.GPUX_SetErrorHandler(C.cast(cb, C.c_void_p), None)
    try:
        g = L.GPU_MakeGraph()
        blit = (C.c_uint8 * 64)()                                                # GPU_OpBlitInfo with a size mismatch: {filter, src, dst, ...}
        n = len(msgs); L.GPU_OpBlit(g, None); assert len(msgs) == n + 1
        L.GPU_DestroyGraph(g)
    finally:
        L.GPUX_SetErrorHandler(None, None)
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
@pytest.mark.parametrize("W", [7680, 8192, 8200])
def test_gpu_post_wide_frames(gpu, W):
    """Widths at the edge of the closed-form tap assumption (the 1/256 snap must absorb the fp32 coordinate error: 7680 = the 8K frame,
    8192 = the limit) and just beyond it, where K8 / K9 / K10 / K11 switch to their general paths (every tap through the snapped
    bilinear sampler): same bits as the oracle on W x 12 frames."""
    import pbrhip
    from pbrhip.synth import synth_post_inputs
    L = gpu
    H = 12
    lighting, depth, vel, vel_prev, history = synth_post_inputs(0x5EED00DA, W, H)
    vel[:, :512] = vel_prev[:, :512] = 0                           # keep the left strip on-screen so that history is sampled there
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_RGBA8UN)
    _upload(gb.lighting_result, lighting); _upload(gb.depth, depth)
    _upload(L.PBR_PostVelocity(pp, 0), vel); _upload(L.PBR_PostVelocity(pp, 1), vel_prev)
    _upload(L.PBR_PostTaaOutput(pp, 1), history)
    g = L.GPU_MakeGraph()
    L.PBR_RecordTaaResolve(pp, g, 0); L.PBR_RecordBloom(pp, g, 0); L.PBR_RecordFinalPostProcessBloom(pp, g, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
    taa = pbrhip.read_mip(L.PBR_PostTaaOutput(pp, 0), 0)
    want = O.taa_resolve(lighting, depth, vel, vel_prev, history).astype(np.float16)
    assert np.array_equal(taa.view(np.uint16), want.view(np.uint16))
    n = L.PBR_PostBloomPassCount(pp)
    down, up = O.bloom_chain(taa, n)
    _check_bloom(L, pbrhip, pp, down, up)
    bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
    want8 = O.unorm8(O.final_post_process(up[0]))
    assert np.abs(bb.astype(np.int32) - want8.astype(np.int32)).max() <= 1
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 32), (96, 64), (256, 144), (130, 68), (520, 264)])
def test_gpu_bloom_instantiations_agree_on_small_and_ragged_frames(gpu, size):
    """The bloom chain's three instantiations (one pixel per thread | 2 x 2 pixels per thread on exact 2 : 1 passes | four lanes per pixel
    through the general sampler) forced onto frames where nearly every block touches an edge: each forced choice gives the oracle's bits
    on every level, for sizes whose chains mix even, odd and 2 : 1 / non-2 : 1 levels."""
    import pbrhip
    from pbrhip.synth import synth_post_inputs
    L = gpu
    W, H = size
    taa, _, _, _, _ = synth_post_inputs(0x5EED00DB, W, H)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    _upload(L.PBR_PostTaaOutput(pp, 0), taa)
    n = L.PBR_PostBloomPassCount(pp)
    want_down, want_up = O.bloom_chain(taa, n)
    try:
        for quad_min, small_max in ((1 << 40, 0), (0, 0), (1 << 40, 1 << 40), (0, 1 << 40)):
            L.pbrk_bloom_set_thresholds(quad_min, small_max)
            g = L.GPU_MakeGraph()
            L.PBR_RecordBloom(pp, g, 0)
            L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
            _check_bloom(L, pbrhip, pp, want_down, want_up)
    finally:
        L.pbrk_bloom_set_thresholds(-1, -1)
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
def test_gpu_graph_folds_keep_a_blit_and_a_clear_that_something_else_can_see(gpu):
    """The submit-time folds (gpu_hip.cpp fold_blits) apply to the bloom chain's own pattern only.  Here the same clear + 1:1 blit are
    followed by a read-back of the copy instead of the bloom draws: the blit must run (the buffer holds the source's texels), and the
    clear -- all levels, directly in front of a whole-level copy onto level 0 -- may skip level 0 only: the other levels read zero
    although they held data.  Then the bloom chain itself on the same textures: folded, and bit-exact as everywhere else."""
    import pbrhip
    from pbrhip.synth import synth_post_inputs
    L = gpu
    W, H = 256, 144
    taa, _, _, _, _ = synth_post_inputs(0x5EED00DC, W, H)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    src = L.PBR_PostTaaOutput(pp, 0); dst = L.PBR_PostBloomUpscale(pp)
    _upload(src, taa)
    for m in range(dst.contents.mip_level_count):                  # stale data in every level of the target
        lv = pbrhip.read_mip(dst, m)
        pbrhip.upload_mip(dst, m, np.full(lv.shape, 3.5, np.float16))

    class Off(C.Structure):
        _fields_ = [("x", C.c_int), ("y", C.c_int), ("z", C.c_int)]

    class Blit(C.Structure):                                       # GPU_OpBlitInfo [gpu.h:317-327]
        _fields_ = [("filter", C.c_int), ("src_texture", pbrhip.TexP), ("dst_texture", pbrhip.TexP), ("src_layer", C.c_uint32), ("dst_layer", C.c_uint32),
                    ("src_mip_level", C.c_uint32), ("dst_mip_level", C.c_uint32), ("src_area", Off * 2), ("dst_area", Off * 2)]
    before = L.GPUX_FoldedBlitCount()
    nbytes = L.GPUX_TextureMipBytes(dst, 0)
    buf = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_CPU, None)
    g = L.GPU_MakeGraph()
    L.GPU_OpClearColorF(g, dst, 0xFFFFFFFF, 0.0, 0.0, 0.0, 0.0)
    b = Blit(); b.filter = 0; b.src_texture = src; b.dst_texture = dst
    b.src_area[1] = Off(W, H, 1); b.dst_area[1] = Off(W, H, 1)
    L.GPU_OpBlit(g, C.byref(b))
    L.GPUX_OpCopyTextureMipToBuffer(g, dst, 0, buf, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
    assert L.GPUX_FoldedBlitCount() == before                      # nothing consumed the copy additively: the blit stays
    got = np.frombuffer((C.c_char * nbytes).from_address(buf.contents.data), dtype=np.float16).reshape(H, W, 4)
    assert np.array_equal(got.view(np.uint16), np.asarray(taa, np.float16).view(np.uint16))
    for m in range(1, dst.contents.mip_level_count):
        assert not pbrhip.read_mip(dst, m).any(), m
    L.GPU_DestroyBuffer(buf)
    n = L.PBR_PostBloomPassCount(pp)
    want_down, want_up = O.bloom_chain(taa, n)
    g = L.GPU_MakeGraph()
    L.PBR_RecordBloom(pp, g, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
    assert L.GPUX_FoldedBlitCount() == before + 1
    _check_bloom(L, pbrhip, pp, want_down, want_up)
    L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))


@pytest.mark.gpu
def test_gpu_frames_in_flight_overlap_head_with_previous_tail(gpu):
    """Two graphs in flight at 1920x1080 (kernels long enough to really run side by side): with GPUX_SetGraphOverlap(1) a frame's sweep,
    shade and TAA resolve start when the previous frame's TAA is done and run beside its bloom chain and tone map; with 0 every graph
    waits for all of its predecessor.  Same backbuffer, bit for bit, at every checkpoint of a 40-frame run with a moving camera; and the
    overlap did engage (one submission per frame after the first)."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    L.GPUX_EnableOpTiming(0)
    W, H = 1920, 1080
    gbd = synth.synth_gbuffer_spheres(W, H)
    _, _, vel, vel_prev, history = synth.synth_post_inputs(0x5EED00D2, W, H)
    env = synth.synth_env(64, seed=0x5EED00AA)
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map); L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16); L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    for nm, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, nm), 0, gbd[key])
    lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    pbrhip.upload_mip(L.PBR_PostVelocity(pp, 0), 0, vel); pbrhip.upload_mip(L.PBR_PostVelocity(pp, 1), 0, vel_prev)
    lg = L.PBR_MakeLightgrid(128)
    scene = synth.synth_lightgrid(128, lit=False).view(np.uint16)
    graphs = [L.GPU_MakeGraph(), L.GPU_MakeGraph()]
    frames = 40

    def run(overlap):
        L.GPUX_SetGraphOverlap(overlap)
        pbrhip.upload_mip(L.PBR_PostTaaOutput(pp, 1), 0, history)
        pbrhip.upload_mip(L.PBR_LightgridTexture(lg), 0, scene)
        before = L.GPUX_OverlappedSubmitCount()
        sums = []
        for f in range(frames):
            g = graphs[f % 2]
            if f >= 2:
                L.GPU_GraphWait(g)
            cam = (float(gbd["cam_pos"][0]) + 0.01 * f, float(gbd["cam_pos"][1]), float(gbd["cam_pos"][2]))
            glob = pbrhip.fill_globals(cam, aspect=W / H, frame_idx=f % 59)
            L.PBR_RecordLightgridSweep(lg, g)
            L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
            L.PBR_RecordTaaResolve(pp, g, f); L.PBR_RecordBloom(pp, g, f); L.PBR_RecordFinalPostProcessBloom(pp, g, f)
            L.GPU_GraphSubmit(g)
            if f % 8 == 7:
                L.GPU_GraphWait(graphs[(f + 1) % 2]); L.GPU_GraphWait(g)
                bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
                sums.append((int(bb.astype(np.uint64).sum()), bb[::97, ::89].copy()))
        for g in graphs:
            L.GPU_GraphWait(g)
        return sums, L.GPUX_OverlappedSubmitCount() - before

    try:
        ordered, n0 = run(0)
        overlapped, n1 = run(1)
    finally:
        L.GPUX_SetGraphOverlap(-1); L.GPUX_EnableOpTiming(1)
    assert n0 == 0 and n1 >= frames - 6, (n0, n1)
    assert len(ordered) == len(overlapped) == 5 and ordered[-1][0] > 0
    for (sa, a), (sb, b) in zip(ordered, overlapped):
        assert sa == sb and np.array_equal(a, b)
    for g in graphs:
        L.GPU_DestroyGraph(g)
    L.PBR_DestroyLightgrid(lg); L.PBR_DestroyPostProcess(pp); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb))
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(env_tex)


@pytest.mark.gpu
def test_gpu_frame_chain_hipgraph_replay_is_identical(gpu):
    """GPUX_SetGraphReplay: the per-frame chain (light-grid sweep, shade, TAA resolve, bloom, tone map; two graphs in flight as in
    main.cpp:49-51, 91-99, camera and ping-pong targets changing every frame) submitted through a captured / updated hipGraph
    produces the same backbuffer, bit for bit, as plain stream launches; the first frames (twins still to be built) and a graph
    with a host copy in it take the plain path by themselves."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    L.GPUX_EnableOpTiming(0)                                       # per-op timing keeps a graph on the plain path
    W, H = 320, 180
    gbd = synth.synth_gbuffer_spheres(W, H)
    lighting, depth, vel, vel_prev, history = synth.synth_post_inputs(0x5EED00D1, W, H)
    env = synth.synth_env(64, seed=0x5EED00AA)
    env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 16, 64, 64)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map); L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 1); L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    for nm, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, nm), 0, gbd[key])
    lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    pbrhip.upload_mip(L.PBR_PostVelocity(pp, 0), 0, vel); pbrhip.upload_mip(L.PBR_PostVelocity(pp, 1), 0, vel_prev)
    lg = L.PBR_MakeLightgrid(128)
    scene = synth.synth_lightgrid(128, lit=False).view(np.uint16)
    graphs = [L.GPU_MakeGraph(), L.GPU_MakeGraph()]
    frames = 9

    def run(replay):
        L.GPUX_SetGraphReplay(replay)
        pbrhip.upload_mip(L.PBR_PostTaaOutput(pp, 1), 0, history)
        pbrhip.upload_mip(L.PBR_LightgridTexture(lg), 0, scene)
        outs = []
        for f in range(frames):
            g = graphs[f % 2]
            if f >= 2:
                L.GPU_GraphWait(g)
            cam = (float(gbd["cam_pos"][0]) + 0.01 * f, float(gbd["cam_pos"][1]), float(gbd["cam_pos"][2]))
            glob = pbrhip.fill_globals(cam, aspect=W / H, frame_idx=f % 59)
            L.PBR_RecordLightgridSweepLines(lg, g, f % 3, 0, 128, 0, 128)
            L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
            L.PBR_RecordTaaResolve(pp, g, f); L.PBR_RecordBloom(pp, g, f); L.PBR_RecordFinalPostProcessBloom(pp, g, f)
            L.GPU_GraphSubmit(g)
            if f in (4, frames - 1):
                L.GPU_GraphWait(graphs[(f + 1) % 2])
                L.GPU_GraphWait(g)
                outs.append(pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0).copy())
        for g in graphs:
            L.GPU_GraphWait(g)
        return outs

    st0 = [C.c_uint64() for _ in range(3)]
    plain = run(0)
    L.GPUX_GraphReplayStats(graphs[0], C.byref(st0[0]), C.byref(st0[1]), C.byref(st0[2]))
    assert st0[0].value == 0
    replayed = run(1)
    st = [C.c_uint64() for _ in range(3)]
    L.GPUX_GraphReplayStats(graphs[0], C.byref(st[0]), C.byref(st[1]), C.byref(st[2]))
    L.GPUX_SetGraphReplay(0)
    assert len(plain) == len(replayed) == 2 and float(plain[1].max()) > 0
    for a, b in zip(plain, replayed):
        assert np.array_equal(a, b)
    # graph 0 carried frames 0, 2, 4, 6, 8: all through the captured path (the twins exist since the plain run), one instantiation,
    # the others in-place updates of the same executable graph
    assert st[0].value == 5 and st[2].value == 1 and st[1].value == 4, [x.value for x in st]
    # a graph with a host-visible copy in it is not captured
    host = L.GPU_MakeBuffer(16, pbrhip.BufferFlag_CPU, None); dev = L.GPU_MakeBuffer(16, pbrhip.BufferFlag_GPU, None)
    L.GPUX_SetGraphReplay(1)
    L.GPU_OpCopyBufferToBuffer(graphs[0], host, dev, 0, 0, 16); L.GPU_GraphSubmit(graphs[0]); L.GPU_GraphWait(graphs[0])
    L.GPUX_SetGraphReplay(0)
    L.GPUX_GraphReplayStats(graphs[0], C.byref(st[0]), C.byref(st[1]), C.byref(st[2]))
    assert st[0].value == 5
    L.GPU_DestroyBuffer(host); L.GPU_DestroyBuffer(dev)
    # the same GPU_Graph with a different chain (TAA + tone map only): the kept executable graph cannot be updated in place and is
    # replaced; the frame equals the plain one
    outs = []
    for replay in (0, 1):
        L.GPUX_SetGraphReplay(replay)
        pbrhip.upload_mip(L.PBR_PostTaaOutput(pp, 1), 0, history)
        L.PBR_RecordTaaResolve(pp, graphs[0], 0); L.PBR_RecordFinalPostProcess(pp, graphs[0], 0)
        L.GPU_GraphSubmit(graphs[0]); L.GPU_GraphWait(graphs[0])
        outs.append(pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0).copy())
    L.GPUX_SetGraphReplay(0)
    L.GPUX_GraphReplayStats(graphs[0], C.byref(st[0]), C.byref(st[1]), C.byref(st[2]))
    assert np.array_equal(outs[0], outs[1]) and st[0].value == 6 and st[2].value == 2, [x.value for x in st]
    # a graph whose earlier op rewrites a map that a later shade op samples (here: the mips of the prefiltered cube are
    # regenerated, which drops its apron / cells twins at execution time) must not be captured -- the shade op would rebuild the
    # twins (allocation + synchronisation) inside the capture.  It takes the plain path, and the frame equals the plain one.
    glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H, frame_idx=0)
    lit = []
    for replay in (0, 1):
        L.GPUX_SetGraphReplay(replay)
        L.GPU_OpGenerateMipmaps(graphs[0], maps.tex_specular_env_map)
        L.PBR_RecordLightingPass(lp, graphs[0], C.byref(glob), 0, 0)
        L.GPU_GraphSubmit(graphs[0]); L.GPU_GraphWait(graphs[0])
        lit.append(pbrhip.read_mip(gb.lighting_result, 0).copy())
    L.GPUX_SetGraphReplay(0)
    L.GPUX_GraphReplayStats(graphs[0], C.byref(st[0]), C.byref(st[1]), C.byref(st[2]))
    assert st[0].value == 6, "a graph that rewrites a sampled map before shading must not go through a capture"
    assert np.array_equal(lit[0].view(np.uint16), lit[1].view(np.uint16)) and float(lit[0].astype(np.float32).max()) > 0
    # the same shade alone (twins rebuilt by the plain run above) is captured again
    L.GPUX_SetGraphReplay(1)
    L.PBR_RecordLightingPass(lp, graphs[0], C.byref(glob), 0, 0)
    L.GPU_GraphSubmit(graphs[0]); L.GPU_GraphWait(graphs[0])
    L.GPUX_SetGraphReplay(0)
    L.GPUX_GraphReplayStats(graphs[0], C.byref(st[0]), C.byref(st[1]), C.byref(st[2]))
    assert st[0].value == 7 and np.array_equal(pbrhip.read_mip(gb.lighting_result, 0).view(np.uint16), lit[0].view(np.uint16))
    for g in graphs:
        L.GPU_DestroyGraph(g)
    L.PBR_DestroyLightgrid(lg); L.PBR_DestroyPostProcess(pp); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb))
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(env_tex)
