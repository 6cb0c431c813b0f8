"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker / reported baseline -- never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class OrcGlobals(C.Structure):
    _fields_ = [
        ("clip_space_from_world", C.c_float * 16),
        ("clip_space_from_view", C.c_float * 16),
        ("world_space_from_clip", C.c_float * 16),
        ("view_space_from_clip", C.c_float * 16),
        ("view_space_from_world", C.c_float * 16),
        ("world_space_from_view", C.c_float * 16),
        ("sun_space_from_world", C.c_float * 16),
        ("old_clip_space_from_world", C.c_float * 16),
        ("sun_direction", C.c_float * 4),
        ("camera_pos", C.c_float * 3),
        ("frame_idx_mod_59", C.c_float),
        ("lightgrid_scale", C.c_float),
        ("visualize_lightgrid", C.c_uint32),
    ]


assert C.sizeof(OrcGlobals) == 552


class OrcTex2D(C.Structure):
    _fields_ = [("data", C.c_void_p), ("format", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class OrcTaaInputs(C.Structure):
    _fields_ = [("lighting_result", OrcTex2D), ("gbuffer_depth", OrcTex2D), ("gbuffer_velocity", OrcTex2D),
                ("gbuffer_velocity_prev", OrcTex2D), ("prev_frame_result", OrcTex2D)]


class OrcShadeInputs(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("base_color", C.c_void_p), ("normal", C.c_void_p), ("orm", C.c_void_p),
        ("emissive", C.c_void_p), ("depth", C.c_void_p),
        ("irradiance", C.c_void_p), ("irradiance_size", C.c_int),
        ("prefiltered", C.c_void_p), ("prefiltered_size", C.c_int), ("prefiltered_levels", C.c_int),
        ("lut", C.c_void_p), ("lut_size", C.c_int),
        ("sun_depth_map", OrcTex2D),
        ("lightgrid", C.c_void_p), ("lightgrid_size", C.c_int),
        ("prev_frame", C.POINTER(OrcTex2D)), ("prev_frame_levels", C.c_int),
    ]


SHADE_IBL, SHADE_SHAFTS, SHADE_ANALYTIC, SHADE_SHADOWS, SHADE_GI = 1, 2, 4, 8, 16


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "pbr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_get_threads.restype = C.c_int
    L.orc_rgbe_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    L.orc_rgbe_decode.restype = C.c_int
    L.orc_mip_count.argtypes = [C.c_int, C.c_int]
    L.orc_mip_count.restype = C.c_int
    L.orc_level_offset.argtypes = [C.c_int, C.c_int]
    L.orc_level_offset.restype = C.c_size_t
    L.orc_pyramid_floats.argtypes = [C.c_int]
    L.orc_pyramid_floats.restype = C.c_size_t
    L.orc_build_pyramid.argtypes = [f32p, C.c_int]
    L.orc_set_cube_sampler_snap.argtypes = [C.c_int]
    L.orc_blit_linear.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int]
    L.orc_face_dir.argtypes = [C.c_int, C.c_float, C.c_float, f32p]
    L.orc_cube_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, C.c_float, f32p]
    L.orc_cube_neighbor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_cube_neighbor.restype = C.c_int
    L.orc_env_analytic.argtypes = [f32p, f32p]
    L.orc_sample_angles.argtypes = [C.c_int, f32p]
    L.orc_prefilter_D.argtypes = [C.c_int, C.c_float, f32p]
    L.orc_prefilter_mip.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_irradiance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_brdf_lut.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_f32_to_f16.argtypes = [C.c_float]
    L.orc_f32_to_f16.restype = C.c_uint16
    L.orc_f16_to_f32.argtypes = [C.c_uint16]
    L.orc_f16_to_f32.restype = C.c_float
    L.orc_equirect_to_cube.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_lightgrid_sweep.argtypes = [np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS"), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_lightgrid_sweep.restype = None
    L.orc_tex2d_sample.argtypes = [C.POINTER(OrcTex2D), C.c_float, C.c_float, f32p]
    L.orc_taa_resolve.argtypes = [C.POINTER(OrcTaaInputs), C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_final_post_process.argtypes = [C.POINTER(OrcTex2D), C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_bloom_downsample.argtypes = [C.POINTER(OrcTex2D), C.c_int, C.c_int, C.c_int, f32p]
    L.orc_bloom_upsample.argtypes = [C.POINTER(OrcTex2D), C.c_int, C.c_int, C.c_int, f32p]
    L.orc_gi_exit_counts.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.orc_sinf_det.argtypes = L.orc_cosf_det.argtypes = L.orc_acosf_det.argtypes = [C.c_float]
    L.orc_sinf_det.restype = L.orc_cosf_det.restype = L.orc_acosf_det.restype = C.c_float
    L.orc_shadow_sample.argtypes = [C.POINTER(OrcTex2D), C.c_float, C.c_float, C.c_float]
    L.orc_shadow_sample.restype = C.c_float
    L.orc_unorm8.argtypes = [C.c_float]
    L.orc_unorm8.restype = C.c_uint8
    L.orc_shade.argtypes = [C.POINTER(OrcGlobals), C.POINTER(OrcShadeInputs), C.c_int,
                            C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    _LIB = L
    return L


# ---- reference parameter rules (host side of the reference + SURVEY 8d extension) -------------
REF_ROUGHNESS = (0.0, 0.03, 0.15, 0.4, 0.6)   # gen_prefiltered_env_map.glsl:117


def prefilter_roughness(mip):
    """mips 0-4: reference table; mips >= 5: documented extension min(1, 0.6 + 0.08*(m-4))."""
    if mip < 5:
        return float(np.float32(REF_ROUGHNESS[mip]))
    return float(min(np.float32(1.0), np.float32(0.6) + np.float32(0.08) * np.float32(mip - 4)))


def prefilter_src_lod(mip):
    """gen_prefiltered_env_map.glsl:113 (mip 0 -> lod 1) and :138 (3 + mip); sampler clamps to last level."""
    return 1.0 if mip == 0 else 3.0 + mip


def set_threads(n):
    lib().orc_set_threads(int(n))


def get_threads():
    return lib().orc_get_threads()


def mip_count(w, h=None):
    return lib().orc_mip_count(int(w), int(w if h is None else h))


def level_offset(W, level):
    return lib().orc_level_offset(int(W), int(level))


def pyramid_floats(W):
    return lib().orc_pyramid_floats(int(W))


def build_pyramid(level0):
    """level0: float32 [6][W][W][4] -> flat float32 pyramid [level][face][y][x][4]."""
    level0 = np.ascontiguousarray(level0, dtype=np.float32)
    W = level0.shape[1]
    assert level0.shape == (6, W, W, 4)
    pyr = np.zeros(pyramid_floats(W), dtype=np.float32)
    pyr[: level0.size] = level0.ravel()
    lib().orc_build_pyramid(pyr, W)
    return pyr


def set_cube_sampler_snap(on):
    """0: exact fp32 tap weights (default); 1: texel coordinates and LOD fraction snapped to 1/256 (pbr_oracle.c)."""
    lib().orc_set_cube_sampler_snap(1 if on else 0)


def blit_linear(src, nd_w, nd_h):
    """src: float32 [layers][h][w][4] -> [layers][nd_h][nd_w][4], the linear-blit rule of pbr_oracle.c A2."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    layers, h, w, _ = src.shape
    out = np.zeros((layers, nd_h, nd_w, 4), np.float32)
    lib().orc_blit_linear(src, w, h, out, nd_w, nd_h, layers)
    return out


def pyramid_level(pyr, W, level):
    n = max(1, W >> level)
    off = level_offset(W, level)
    return pyr[off: off + 6 * n * n * 4].reshape(6, n, n, 4)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def cube_sample(pyr, W, dirs, lod):
    dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    levels = mip_count(W) if pyr is not None else 1
    out = np.zeros((dirs.shape[0], 4), dtype=np.float32)
    L = lib()
    for k in range(dirs.shape[0]):
        L.orc_cube_sample(_ptr(pyr), W, levels, dirs[k], float(lod), out[k])
    return out


def rgbe_decode(data: bytes):
    w, h = C.c_int(), C.c_int()
    rc = lib().orc_rgbe_decode(data, len(data), C.byref(w), C.byref(h), None)
    if rc:
        raise ValueError(f"rgbe decode failed: {rc}")
    out = np.zeros((h.value, w.value, 4), dtype=np.float32)
    rc = lib().orc_rgbe_decode(data, len(data), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(f"rgbe decode failed: {rc}")
    return out


def equirect_to_cube(eq, size):
    eq = np.ascontiguousarray(eq, dtype=np.float32)
    h, w, _ = eq.shape
    out = np.zeros((6, size, size, 4), dtype=np.float32)
    lib().orc_equirect_to_cube(eq.reshape(-1), w, h, size, out.reshape(-1))
    return out


def prefilter_mip(pyr, W, out_size, mip, roughness=None, src_lod=None, nsamples=8192, literal=False,
                  faces=(0, 6), rows=None, levels=None):
    size = max(1, out_size >> mip)
    if roughness is None:
        roughness = prefilter_roughness(mip)
    if src_lod is None:
        src_lod = prefilter_src_lod(mip)
    if rows is None:
        rows = (0, size)
    if levels is None:
        levels = mip_count(W) if pyr is not None else 1
    out = np.zeros((6, size, size, 4), dtype=np.float32)
    lib().orc_prefilter_mip(_ptr(pyr), W, levels, size, mip, roughness, src_lod, nsamples, int(literal),
                            faces[0], faces[1], rows[0], rows[1], out)
    return out


def irradiance(pyr, W, out_size=32, src_lod=6.0, nsamples=1024, literal=False, faces=(0, 6), rows=None):
    if rows is None:
        rows = (0, out_size)
    levels = mip_count(W) if pyr is not None else 1
    out = np.zeros((6, out_size, out_size, 4), dtype=np.float32)
    lib().orc_irradiance(_ptr(pyr), W, levels, out_size, src_lod, nsamples, int(literal),
                         faces[0], faces[1], rows[0], rows[1], out)
    return out


def brdf_lut(size=256, nsamples=4096, rows=None):
    if rows is None:
        rows = (0, size)
    out = np.zeros((size, size, 2), dtype=np.float32)
    lib().orc_brdf_lut(size, nsamples, rows[0], rows[1], out)
    return out


def f32_to_f16_bits(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    L = lib()
    return np.array([L.orc_f32_to_f16(float(v)) for v in a.ravel()], dtype=np.uint16).reshape(a.shape)


def make_globals(mats: dict, sun_direction, camera_pos, frame_idx_mod_59=0.0, lightgrid_scale=1.0 / 40.0):
    g = OrcGlobals()
    for name, _ in OrcGlobals._fields_[:8]:
        m = np.asarray(mats.get(name, np.eye(4, dtype=np.float32).T.ravel()), dtype=np.float32).ravel()
        getattr(g, name)[:] = m.tolist()
    sd = list(sun_direction) + [0.0] * (4 - len(sun_direction))
    g.sun_direction[:] = [float(v) for v in sd]
    g.camera_pos[:] = [float(v) for v in camera_pos]
    g.frame_idx_mod_59 = float(frame_idx_mod_59)
    g.lightgrid_scale = float(lightgrid_scale)
    g.visualize_lightgrid = 0
    return g


def shade(g, base, normal, orm, emissive, depth, flags=0, irradiance_cube=None, prefiltered_pyr=None,
          prefiltered_size=0, lut_half=None, region=None, sun_depth_map=None, lightgrid=None, prev_frame_levels=None):
    H, W = depth.shape
    keep = [np.ascontiguousarray(a, dtype=np.uint8) for a in (base, normal, orm, emissive)]
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    si = OrcShadeInputs()
    si.width, si.height = W, H
    si.base_color, si.normal, si.orm, si.emissive = [_ptr(a) for a in keep]
    si.depth = _ptr(depth)
    if irradiance_cube is not None:
        irradiance_cube = np.ascontiguousarray(irradiance_cube, dtype=np.float32)
        si.irradiance = _ptr(irradiance_cube)
        si.irradiance_size = irradiance_cube.shape[1]
    if prefiltered_pyr is not None:
        prefiltered_pyr = np.ascontiguousarray(prefiltered_pyr, dtype=np.float32)
        si.prefiltered = _ptr(prefiltered_pyr)
        si.prefiltered_size = prefiltered_size
        si.prefiltered_levels = mip_count(prefiltered_size)
    if lut_half is not None:
        lut_half = np.ascontiguousarray(lut_half, dtype=np.uint16)
        si.lut = _ptr(lut_half)
        si.lut_size = lut_half.shape[0]
    if sun_depth_map is not None:
        si.sun_depth_map, keep_sun = _tex2d(np.asarray(sun_depth_map, np.float32), TEX_R32F)
    if lightgrid is not None:
        grid = np.asarray(lightgrid)
        grid = np.ascontiguousarray(grid.view(np.uint16) if grid.dtype == np.float16 else grid, dtype=np.uint16)
        si.lightgrid = grid.ctypes.data_as(C.c_void_p)
        si.lightgrid_size = grid.shape[0]
    if prev_frame_levels is not None:
        pairs = [_tex2d(lv, TEX_RGBA16F) for lv in prev_frame_levels]
        arr = (OrcTex2D * len(pairs))(*[p[0] for p in pairs])
        si.prev_frame = C.cast(arr, C.POINTER(OrcTex2D))
        si.prev_frame_levels = len(pairs)
    out = np.zeros((H, W, 4), dtype=np.float32)
    x0, x1, y0, y1 = region if region is not None else (0, W, 0, H)
    lib().orc_shade(C.byref(g), C.byref(si), int(flags), x0, x1, y0, y1, out)
    return out


def lightgrid_sweep(grid_half, direction, ny=None, nz=None):
    """grid_half: uint16 [d][h][w][4] (RGBA16F bit patterns); returns the swept copy.  ny/nz default to the
    full extent of the two non-line axes (direction 0: (h, d); 1: (d, w); 2: (w, h))."""
    g = np.array(grid_half, dtype=np.uint16, order="C", copy=True)
    d, h, w, _ = g.shape
    full = {0: (h, d), 1: (d, w), 2: (w, h)}[int(direction)]
    ny = full[0] if ny is None else ny
    nz = full[1] if nz is None else nz
    lib().orc_lightgrid_sweep(g.reshape(-1), w, h, d, int(direction), int(ny), int(nz))
    return g


TEX_RGBA16F, TEX_RG16F, TEX_R32F, TEX_RGBA32F = 0, 1, 2, 3


def _tex2d(arr, fmt):
    """arr: [h][w][c] (or [h][w]) numpy array already in the storage type of fmt; returns (OrcTex2D, keepalive)."""
    want = {TEX_RGBA16F: (np.uint16, 4), TEX_RG16F: (np.uint16, 2), TEX_R32F: (np.float32, 1), TEX_RGBA32F: (np.float32, 4)}[fmt]
    a = np.asarray(arr)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    a = np.ascontiguousarray(a, dtype=want[0])
    h, w = a.shape[:2]
    assert a.size == h * w * want[1], (a.shape, fmt)
    t = OrcTex2D()
    t.data = a.ctypes.data_as(C.c_void_p)
    t.format, t.width, t.height = fmt, w, h
    return t, a


def tex2d_sample(arr, fmt, u, v):
    t, keep = _tex2d(arr, fmt)
    out = np.zeros(4, np.float32)
    lib().orc_tex2d_sample(C.byref(t), float(u), float(v), out)
    return out


def taa_resolve(lighting, depth, velocity, velocity_prev, history, rows=None):
    """RGBA16F lighting/history, R32F depth, RG16F velocities (float16 or uint16 bit patterns) -> float32 [h][w][4]."""
    ti = OrcTaaInputs()
    keep = []
    for name, arr, fmt in (("lighting_result", lighting, TEX_RGBA16F), ("gbuffer_depth", depth, TEX_R32F),
                           ("gbuffer_velocity", velocity, TEX_RG16F), ("gbuffer_velocity_prev", velocity_prev, TEX_RG16F),
                           ("prev_frame_result", history, TEX_RGBA16F)):
        t, k = _tex2d(arr, fmt)
        setattr(ti, name, t)
        keep.append(k)
    h, w = keep[0].shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    y0, y1 = rows if rows is not None else (0, h)
    lib().orc_taa_resolve(C.byref(ti), w, h, y0, y1, out.reshape(-1))
    return out


def final_post_process(src_rgba16f, width=None, height=None):
    t, keep = _tex2d(src_rgba16f, TEX_RGBA16F)
    h, w = keep.shape[:2]
    width, height = width or w, height or h
    out = np.zeros((height, width, 4), np.float32)
    lib().orc_final_post_process(C.byref(t), width, height, 0, height, out.reshape(-1))
    return out


def unorm8(arr):
    f = np.clip(np.asarray(arr, np.float32), 0.0, 1.0) * np.float32(255.0)
    return np.rint(f).astype(np.uint8)                    # rint = round half to even, like lrintf


def bloom_pass(src_rgba16f, dw, dh, dst_mip_level, up):
    t, keep = _tex2d(src_rgba16f, TEX_RGBA16F)
    out = np.zeros((dh, dw, 4), np.float32)
    (lib().orc_bloom_upsample if up else lib().orc_bloom_downsample)(C.byref(t), dw, dh, int(dst_mip_level), out.reshape(-1))
    return out


def bloom_chain(taa_rgba16f, passes=6):
    """render.cpp:1139-1176 on the CPU: `passes` downsamples into the mips of bloom_downscale_rt (half size), clear +
    1:1 blit of the TAA result into bloom_upscale_rt mip 0, `passes` additive upsamples.  Render targets are RGBA16F:
    every pass stores fp16 (RTE); additive blending = fp16(src + dst) with alpha = src alpha (ONE, ZERO).
    Returns (downscale_mips, upscale_mips) as lists of float16 arrays."""
    taa = np.asarray(taa_rgba16f)
    taa = taa.view(np.float16) if taa.dtype == np.uint16 else taa.astype(np.float16)
    H, W = taa.shape[:2]
    dims = lambda w, h, m: (max(1, w >> m), max(1, h >> m))
    down = []
    src = taa
    for step in range(passes):                              # :1141-1152, dst_mip_level = step + 1
        dw, dh = dims(W // 2, H // 2, step)
        down.append(bloom_pass(src, dw, dh, step + 1, up=False).astype(np.float16))
        src = down[-1]
    up = [np.zeros(dims(W, H, m)[::-1] + (4,), np.float16) for m in range(passes)]      # :1156 clear
    up[0] = taa.copy()                                      # :1158-1163 blit
    for step in range(passes):                              # :1165-1176, dst level = passes - 1 - step
        dst_level = passes - 1 - step
        src = down[passes - 1] if step == 0 else up[passes - step]
        dw, dh = dims(W, H, dst_level)
        frag = bloom_pass(src, dw, dh, dst_level, up=True)
        blended = frag[..., :3] + up[dst_level][..., :3].astype(np.float32)
        up[dst_level] = np.concatenate([blended, frag[..., 3:]], axis=-1).astype(np.float16)
    return down, up


def gi_exit_counts(reset=True):
    out = (C.c_uint64 * 4)()
    lib().orc_gi_exit_counts(out, int(reset))
    return [int(v) for v in out]
