"""Light-grid sweep (shaders/lightgrid_sweep.glsl): the oracle against the Oracle-A fixtures (the reference's
shader text executed on the CPU, oracle/gen_oracle_a.py --only sweep), and the HIP kernel K7 against both through the
GPU_* boundary.  RGBA16F bit patterns must match exactly."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import pbr_oracle as O


def _fixtures(golden_dir):
    with open(os.path.join(golden_dir, "oracle_a_meta.json")) as f:
        meta = json.load(f)["lightgrid_sweep"]
    for e in meta["fixtures"]:
        z = np.load(os.path.join(golden_dir, e["file"]))
        yield e, z["grid"], z["swept"]


def test_oracle_matches_reference_shader(golden_dir):
    n = 0
    for e, grid, swept in _fixtures(golden_dir):
        mine = O.lightgrid_sweep(grid, e["direction"], e["ny"], e["nz"])
        assert np.array_equal(mine, swept), f"direction {e['direction']}"
        changed = (swept != grid).any(axis=-1)
        alpha = grid[..., 3].view(np.float16)
        assert not changed[alpha >= 0.5].any()              # :72 only voxels with old alpha < 0.5 are stored
        assert np.array_equal(swept[..., 3], grid[..., 3])  # mix(a, a, .35) rounds back to a
        n += 1
    assert n == 3


def test_oracle_sweep_properties():
    from pbrhip import synth
    g = synth.synth_lightgrid(128, lit=True).view(np.uint16)
    # the three directions are one algorithm on permuted axes: sweeping y equals sweeping x of the (x<->y)-transposed grid
    sx = O.lightgrid_sweep(g, 0)
    gy = np.ascontiguousarray(g.transpose(0, 2, 1, 3))      # [z][x][y]: old x axis is now "y"
    sy = O.lightgrid_sweep(gy, 1)
    assert np.array_equal(sy.transpose(0, 2, 1, 3), sx)
    gz = np.ascontiguousarray(g.transpose(2, 1, 0, 3))      # [x][y][z]
    sz = O.lightgrid_sweep(gz, 2)
    assert np.array_equal(sz.transpose(2, 1, 0, 3), sx)
    # an empty, dark line fills from the sky light (1, 1.2, 2) at both ends and stays symmetric left/right in its interior
    e = np.zeros((8, 8, 128, 4), np.uint16)
    s = O.lightgrid_sweep(e, 0).view(np.float16).astype(np.float64)
    assert np.all(s[..., :3] >= 0) and np.all(s[0, 0, 0, :3] > s[0, 0, 40, :3])
    assert np.allclose(s[0, 0, 0, :3] / s[0, 0, 0, 0], [1.0, 1.2, 2.0], rtol=2e-3)
    # sub-range dispatch touches only its own lines
    part = O.lightgrid_sweep(g, 2, 64, 128)
    full = O.lightgrid_sweep(g, 2)
    assert np.array_equal(part[:, :, :64], full[:, :, :64]) and np.array_equal(part[:, :, 64:], g[:, :, 64:])


# ------------------------------------------------------------------------------------------- GPU

def _sweep_pipeline(L, fmt):
    import pbrhip
    layout = L.GPU_InitPipelineLayout()
    b_img = L.GPU_StorageImageBinding(layout, b"IMG0", fmt)                     # render.cpp:816
    L.GPU_FinalizePipelineLayout(layout)
    path = b"../src/demo_pbr_renderer/shaders/lightgrid_sweep.glsl"
    desc = pbrhip.GPU_ShaderDesc()
    desc.glsl_debug_filepath = pbrhip.GPU_String(path, len(path))
    errs = pbrhip.GPU_GLSLErrorArray()
    desc.spirv = L.GPU_SPIRVFromGLSL(None, 2, layout, C.byref(desc), C.byref(errs))
    assert desc.spirv.length > 0 and errs.length == 0
    pipe = L.GPU_MakeComputePipeline(layout, C.byref(desc))
    assert pipe
    return layout, b_img, pipe


def _dispatch(L, layout, pipe, ds, direction, groups=None, lines=None):
    g = L.GPU_MakeGraph()
    L.GPU_OpBindComputePipeline(g, pipe)
    L.GPU_OpBindComputeDescriptorSet(g, ds)
    d = C.c_uint32(direction)
    L.GPU_OpPushComputeConstants(g, layout, C.byref(d), 4)                      # render.cpp:1069
    if lines is None:
        L.GPU_OpDispatch(g, *groups)                                            # render.cpp:1072
    else:
        L.GPUX_OpDispatchLines(g, *lines)
    L.GPU_GraphSubmit(g)
    L.GPU_GraphWait(g)
    L.GPU_DestroyGraph(g)


@pytest.mark.gpu
def test_gpu_sweep_matches_reference_shader(gpu, golden_dir):
    """The reference's own call sequence (render.cpp:151-187, 1067-1072) on the fixture grids: bit-exact against the shader text."""
    import pbrhip
    L = gpu
    layout, b_img, pipe = _sweep_pipeline(L, pbrhip.Format_RGBA16F)
    for e, grid, swept in _fixtures(golden_dir):
        d, h, w, _ = grid.shape
        tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, w, h, pbrhip.TextureFlag_StorageImage, depth=d)
        pbrhip.upload_mip(tex, 0, grid)
        ds = L.GPU_InitDescriptorSet(None, layout)
        L.GPU_SetStorageImageBinding(ds, b_img, tex, 0)
        L.GPU_FinalizeDescriptorSet(ds)
        _dispatch(L, layout, pipe, ds, e["direction"], groups=(1, e["ny"] // 8, e["nz"] // 8))
        got = pbrhip.read_mip(tex, 0).view(np.uint16)
        assert np.array_equal(got, swept), f"direction {e['direction']}: {(got != swept).sum()} halfs differ"
        L.GPU_DestroyDescriptorSet(ds)
        L.GPU_DestroyTexture(tex)
    L.GPU_DestroyComputePipeline(pipe)
    L.GPU_DestroyPipelineLayout(layout)


@pytest.mark.gpu
def test_gpu_sweep_full_grid_frames(gpu):
    """128^3 grid through the host layer: frame-0 clear, upload of a voxelised scene, then six frames of sweeps in the
    reference's direction order (1, 2, 0, ...); every frame bit-exact against the oracle; sharded = full."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    lg = L.PBR_MakeLightgrid(128)
    tex = L.PBR_LightgridTexture(lg)
    g = L.GPU_MakeGraph()
    L.PBR_RecordLightgridClear(lg, g)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
    assert not pbrhip.read_mip(tex, 0).view(np.uint16).any()
    scene = synth.synth_lightgrid(128, lit=False).view(np.uint16)
    pbrhip.upload_mip(tex, 0, scene)
    want = scene
    dirs = []
    for frame in range(6):
        g = L.GPU_MakeGraph()
        L.PBR_RecordLightgridSweep(lg, g)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        direction = L.PBR_LightgridSweepDirection(lg)
        dirs.append(direction)
        want = O.lightgrid_sweep(want, direction)
        got = pbrhip.read_mip(tex, 0).view(np.uint16)
        assert np.array_equal(got, want), f"frame {frame} direction {direction}: {(got != want).sum()} halfs differ"
    assert dirs == [1, 2, 0, 1, 2, 0]                                            # render.cpp:1064-1065 from a zeroed renderer
    occupied = scene[..., 3].view(np.float16) > 0.5
    assert np.array_equal(got[occupied], scene[occupied])                        # occupied voxels are never written
    lit = got.view(np.float16)[~occupied][:, :3].astype(np.float64)
    assert lit.min() >= 0 and lit.max() < 64 and lit.mean() > 0.01

    # sharded (lines are independent): two half-range dispatches of a lit grid == one full dispatch
    start = synth.synth_lightgrid(128, seed=0x5EED00B8, lit=True).view(np.uint16)
    for direction in (0, 1, 2):
        full = O.lightgrid_sweep(start, direction)
        pbrhip.upload_mip(tex, 0, start)
        g = L.GPU_MakeGraph()
        L.PBR_RecordLightgridSweepLines(lg, g, direction, 0, 128, 0, 40)
        L.PBR_RecordLightgridSweepLines(lg, g, direction, 0, 72, 40, 128)
        L.PBR_RecordLightgridSweepLines(lg, g, direction, 72, 128, 40, 128)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        got = pbrhip.read_mip(tex, 0).view(np.uint16)
        assert np.array_equal(got, full), f"sharded direction {direction}"
        # odd range boundaries: tiles start on odd voxels (the kernel's 8-byte store path) and end in partial tiles
        pbrhip.upload_mip(tex, 0, start)
        g = L.GPU_MakeGraph()
        L.PBR_RecordLightgridSweepLines(lg, g, direction, 0, 128, 0, 41)
        L.PBR_RecordLightgridSweepLines(lg, g, direction, 0, 73, 41, 128)
        L.PBR_RecordLightgridSweepLines(lg, g, direction, 73, 128, 41, 128)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyGraph(g)
        got = pbrhip.read_mip(tex, 0).view(np.uint16)
        assert np.array_equal(got, full), f"sharded (odd boundaries) direction {direction}"
    L.PBR_DestroyLightgrid(lg)


@pytest.mark.gpu
def test_gpu_sweep_boundary_errors(gpu):
    import pbrhip
    L = gpu
    msgs = []
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    cb = CB(lambda m, u: msgs.append(m.decode()))
    L.GPUX_SetErrorHandler(C.cast(cb, C.c_void_p), None)
    try:
        layout, b_img, pipe = _sweep_pipeline(L, pbrhip.Format_RGBA16F)
        small = pbrhip.make_texture(pbrhip.Format_RGBA16F, 64, 64, pbrhip.TextureFlag_StorageImage, depth=64)
        grid = pbrhip.make_texture(pbrhip.Format_RGBA16F, 128, 16, pbrhip.TextureFlag_StorageImage, depth=8)
        ds_small = L.GPU_InitDescriptorSet(None, layout); L.GPU_SetStorageImageBinding(ds_small, b_img, small, 0); L.GPU_FinalizeDescriptorSet(ds_small)
        ds = L.GPU_InitDescriptorSet(None, layout); L.GPU_SetStorageImageBinding(ds, b_img, grid, 0); L.GPU_FinalizeDescriptorSet(ds)
        assert not msgs
        g = L.GPU_MakeGraph()
        L.GPU_OpBindComputePipeline(g, pipe)
        L.GPU_OpBindComputeDescriptorSet(g, ds)
        n = len(msgs); L.GPU_OpDispatch(g, 1, 2, 1); assert len(msgs) == n + 1 and "X_direction" in msgs[-1]      # no push constant yet
        d = C.c_uint32(0)
        L.GPU_OpPushComputeConstants(g, layout, C.byref(d), 4)
        n = len(msgs); L.GPU_OpDispatch(g, 2, 2, 1); assert len(msgs) == n + 1                                     # gx != 1
        n = len(msgs); L.GPU_OpDispatch(g, 1, 3, 1); assert len(msgs) == n + 1 and "leaves" in msgs[-1]           # 24 rows of 16
        n = len(msgs); L.GPU_OpDispatch(g, 1, 2, 1); assert len(msgs) == n                                         # fits: 16 x 8 lines along x
        d = C.c_uint32(1)
        L.GPU_OpPushComputeConstants(g, layout, C.byref(d), 4)
        n = len(msgs); L.GPU_OpDispatch(g, 1, 1, 1); assert len(msgs) == n + 1                                     # y axis is only 16 long
        L.GPU_OpBindComputeDescriptorSet(g, ds_small)
        n = len(msgs); L.GPU_OpDispatch(g, 1, 1, 1); assert len(msgs) == n + 1                                     # 64^3: no axis holds a 128-voxel line
        n = len(msgs); L.GPUX_OpDispatchRows(g, 0, 1, 0, 8); assert len(msgs) == n + 1
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)                                                                   # the one valid dispatch runs
        L.GPU_DestroyGraph(g)
        # wrong image format for IMG0
        rgba32 = pbrhip.make_texture(pbrhip.Format_RGBA32F, 128, 8, pbrhip.TextureFlag_StorageImage, depth=8)
        ds32 = L.GPU_InitDescriptorSet(None, layout); L.GPU_SetStorageImageBinding(ds32, b_img, rgba32, 0); L.GPU_FinalizeDescriptorSet(ds32)
        g = L.GPU_MakeGraph()
        L.GPU_OpBindComputePipeline(g, pipe); L.GPU_OpBindComputeDescriptorSet(g, ds32)
        d = C.c_uint32(0); L.GPU_OpPushComputeConstants(g, layout, C.byref(d), 4)
        n = len(msgs); L.GPU_OpDispatch(g, 1, 1, 1); assert len(msgs) > n and "RGBA16F" in msgs[-1]
        L.GPU_DestroyGraph(g)
        for s in (ds, ds_small, ds32):
            L.GPU_DestroyDescriptorSet(s)
        for t in (small, grid, rgba32):
            L.GPU_DestroyTexture(t)
        L.GPU_DestroyComputePipeline(pipe); L.GPU_DestroyPipelineLayout(layout)
    finally:
        L.GPUX_SetErrorHandler(None, None)
