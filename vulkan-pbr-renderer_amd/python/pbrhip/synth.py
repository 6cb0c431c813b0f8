"""Seeded synthetic inputs for the IBL / shade hot path.

The reference's real inputs (shipyard_cranes_track_cube.hdr, MetalRoughSpheres.glb, SunTemple.fbx)
are absent from the mount, so benchmarks and parity tests use these generators.  Pure numpy; no
oracle and no GPU code is involved.  Everything is deterministic in (size, seed).
"""
import numpy as np

# ---- cube geometry (Vulkan face table quoted in gen_prefiltered_env_map.glsl:12-23) ----------


def face_dirs(W, dtype=np.float64):
    """Unit directions of texel centres, shape [6][W][W][3] (face, row=y, col=x)."""
    c = (np.arange(W, dtype=dtype) + 0.5) / W
    u, v = np.meshgrid(c, c, indexing="xy")
    sc, tc = 2 * (u - 0.5), 2 * (v - 0.5)
    one = np.ones_like(sc)
    faces = [
        (one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one),
    ]
    d = np.stack([np.stack(f, axis=-1) for f in faces], axis=0)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


# ---- integer hash noise ----------------------------------------------------------------------


def _pcg_hash(x):
    x = x.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    x = (x * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(0xFFFFFFFF)
    sh = ((x >> np.uint64(28)) + np.uint64(4))
    w = (((x >> sh) ^ x) * np.uint64(277803737)) & np.uint64(0xFFFFFFFF)
    return ((w >> np.uint64(22)) ^ w) & np.uint64(0xFFFFFFFF)


def _lattice(ix, iy, iz, seed):
    h = _pcg_hash(ix.astype(np.int64) & 0xFFFFFFFF)
    h = _pcg_hash(h ^ (iy.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64))
    h = _pcg_hash(h ^ (iz.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64))
    h = _pcg_hash(h ^ np.uint64(seed & 0xFFFFFFFF))
    return h.astype(np.float64) / 4294967296.0


def value_noise3(p, seed):
    """Trilinear value noise on the integer lattice; p[...,3] float64 -> [0,1)."""
    pf = np.floor(p)
    f = p - pf
    f = f * f * (3 - 2 * f)
    ix, iy, iz = (pf[..., k].astype(np.int64) for k in range(3))
    out = 0
    for dx in (0, 1):
        wx = f[..., 0] if dx else 1 - f[..., 0]
        for dy in (0, 1):
            wy = f[..., 1] if dy else 1 - f[..., 1]
            for dz in (0, 1):
                wz = f[..., 2] if dz else 1 - f[..., 2]
                out = out + wx * wy * wz * _lattice(ix + dx, iy + dy, iz + dz, seed)
    return out


# ---- RGBE -------------------------------------------------------------------------------------


def rgbe_encode(rgb):
    """float [...,3] -> uint8 [...,4] (Radiance convention: value = mantissa * 2^(e-136))."""
    rgb = np.asarray(rgb, dtype=np.float64)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    ok = m > 1e-32
    mant, ex = np.frexp(np.where(ok, m, 1.0))        # m = mant * 2^ex, mant in [0.5,1)
    scale = np.where(ok, mant * 256.0 / np.where(ok, m, 1.0), 0.0)
    out[..., :3] = np.clip(np.floor(rgb * scale[..., None]), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(ok, ex + 128, 0).astype(np.uint8)
    out[~ok] = 0
    return out


def rgbe_decode(rgbe):
    """uint8 [...,4] -> float32 [...,4] with A = 1, exactly as stbi__hdr_convert does."""
    rgbe = np.asarray(rgbe, dtype=np.uint8)
    e = rgbe[..., 3].astype(np.int32)
    f1 = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = np.ones(rgbe.shape[:-1] + (4,), dtype=np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f1[..., None]
    out[..., :3][e == 0] = 0
    return out


def hdr_file_bytes(rgbe_rows, rle=True):
    """Encode uint8 [H][W][4] RGBE pixels as a Radiance .hdr file (new-style RLE or flat)."""
    H, W, _ = rgbe_rows.shape
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {H} +X {W}\n".encode()
    if not rle or W < 8 or W >= 32768:
        return head + rgbe_rows.tobytes()
    out = bytearray(head)
    for y in range(H):
        out += bytes([2, 2, (W >> 8) & 0xFF, W & 0xFF])
        for k in range(4):
            ch = rgbe_rows[y, :, k]
            i = 0
            while i < W:
                # find a run
                run = 1
                while i + run < W and run < 127 and ch[i + run] == ch[i]:
                    run += 1
                if run >= 4:
                    out += bytes([128 + run, int(ch[i])])
                    i += run
                else:
                    j = i
                    while j < W and j - i < 128:
                        r = 1
                        while j + r < W and r < 4 and ch[j + r] == ch[j]:
                            r += 1
                        if r >= 4:
                            break
                        j += 1
                    n = max(1, j - i)
                    out += bytes([n]) + ch[i:i + n].tobytes()
                    i += n
    return bytes(out)


# ---- environment ------------------------------------------------------------------------------

SUN_DIR = np.array([0.35, 0.55, 0.757], dtype=np.float64)
SUN_DIR /= np.linalg.norm(SUN_DIR)


def _env_face(args):
    W, seed, rgbe_roundtrip, f, chunk_rows = args
    out = np.empty((W, W, 4), dtype=np.float32)
    c = (np.arange(W, dtype=np.float64) + 0.5) / W
    for y0 in range(0, W, chunk_rows):
        y1 = min(W, y0 + chunk_rows)
        u, v = np.meshgrid(c, c[y0:y1], indexing="xy")
        sc, tc = 2 * (u - 0.5), 2 * (v - 0.5)
        one = np.ones_like(sc)
        comps = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)][f]
        d = np.stack(comps, axis=-1)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        up = d[..., 2]
        t = np.clip(up * 0.5 + 0.5, 0, 1)
        sky = np.stack([0.25 + 0.5 * t, 0.35 + 0.65 * t, 0.55 + 0.95 * t], axis=-1)
        ground = np.stack([0.18 + 0 * t, 0.15 + 0 * t, 0.12 + 0 * t], axis=-1)
        col = np.where((up > 0)[..., None], sky, ground)
        band = np.exp(-(up / 0.06) ** 2)
        col = col + band[..., None] * np.array([0.9, 0.7, 0.45])
        n1 = value_noise3(d * 6.0 + 17.0, seed)
        n2 = value_noise3(d * 23.0 + 5.0, seed ^ 0x9E3779B9)
        col = col * (0.6 + 0.5 * n1 + 0.3 * n2)[..., None]
        cs = d @ SUN_DIR
        ang = np.arccos(np.clip(cs, -1, 1))
        sun = 5.0e4 * np.exp(-(ang / 0.012) ** 4) + 40.0 * np.exp(-(ang / 0.08) ** 2)
        col = col + sun[..., None] * np.array([1.0, 0.92, 0.8])
        if rgbe_roundtrip:
            out[y0:y1] = rgbe_decode(rgbe_encode(col))
        else:
            out[y0:y1, :, :3] = col.astype(np.float32)
            out[y0:y1, :, 3] = 1.0
    return out


def synth_env(W, seed=0x5EED0001, rgbe_roundtrip=True, chunk_rows=256, workers=1):
    """Procedural HDR cube [6][W][W][4] float32: sky gradient + sun disc (peak 5e4) + horizon band
    + value noise, round-tripped through RGBE so the values are what stbi_loadf would return.
    workers > 1 computes the six faces in separate processes (same result)."""
    jobs = [(W, seed, rgbe_roundtrip, f, chunk_rows) for f in range(6)]
    if workers > 1:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(min(6, workers))
        try:
            faces = pool.map(_env_face, jobs)
        finally:
            pool.close(); pool.join()          # let workers exit on their own: Pool.terminate() signals them, which hangs under a preloaded profiler
    else:
        faces = [_env_face(j) for j in jobs]
    return np.stack(faces, axis=0)


def env_to_hdr_strip(env, rle=True):
    """[6][W][W][4] float cube -> bytes of a W x 6W vertical-strip .hdr (asset_import.cpp:17-27 layout)."""
    W = env.shape[1]
    rows = rgbe_encode(env[..., :3].reshape(6 * W, W, 3))
    return hdr_file_bytes(rows, rle=rle)


# ---- camera / Globals (numpy float64 model used only to build synthetic G-buffers) -------------


def perspective_lh_zo(fov_deg, aspect, near, far):
    """HMM_Perspective_LH_ZO as used by utils/camera.h:112 (column-major list of columns)."""
    cot = 1.0 / np.tan(np.deg2rad(fov_deg) / 2.0)
    m = np.zeros((4, 4))          # m[col][row]
    m[0][0] = cot / aspect
    m[1][1] = cot
    m[2][3] = 1.0
    m[2][2] = -(far / (near - far))
    m[3][2] = (near * far) / (near - far)
    return m


def quat_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = np.sin(angle / 2)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(angle / 2)])


def quat_to_mat3(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def camera_matrices(pos, ori_q, fov_deg=75.0, aspect=16.0 / 9.0, near=0.02, far=1.0e4):
    """Row-major 4x4 (math convention, M @ column-vector) float64 matrices of the camera model."""
    R = quat_to_mat3(ori_q)
    wfv = np.eye(4)
    wfv[:3, :3] = R
    wfv[:3, 3] = pos
    vfw = np.linalg.inv(wfv)
    cfv = perspective_lh_zo(fov_deg, aspect, near, far).T    # to row-major math convention
    cfw = cfv @ vfw
    return dict(world_from_view=wfv, view_from_world=vfw, clip_from_view=cfv, clip_from_world=cfw,
                world_from_clip=np.linalg.inv(cfw), view_from_clip=np.linalg.inv(cfv))


DEFAULT_ORI = quat_axis_angle((1, 0, 0), -np.pi / 2)     # utils/camera.h:45


def _encode_unorm8(x):
    return np.clip(np.floor(x * 255.0 + 0.5), 0, 255).astype(np.uint8)


PALETTE = np.array([[0.95, 0.64, 0.54], [0.91, 0.92, 0.92], [1.0, 0.77, 0.34], [0.56, 0.57, 0.58],
                    [0.8, 0.1, 0.1], [0.1, 0.5, 0.8], [0.2, 0.7, 0.3]])


def synth_gbuffer_spheres(width=1920, height=1080, cam_pos=(0.0, -9.0, 0.0), ori_q=None):
    """C3: 7x7 grid of unit-radius-0.55 spheres in the XZ plane at y=0, camera at (0,-9,0) facing +Y.
    metallic = row/6, roughness = col/6.  Returns dict of G-buffer arrays + camera matrices."""
    ori_q = DEFAULT_ORI if ori_q is None else ori_q
    cam = camera_matrices(np.asarray(cam_pos, dtype=np.float64), ori_q, aspect=width / height)
    xs = (np.arange(width) + 0.5) / width * 2 - 1
    ys = (np.arange(height) + 0.5) / height * 2 - 1
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    wfc = cam["world_from_clip"]

    def unproject(z):
        p = np.stack([X, Y, np.full_like(X, z), np.ones_like(X)], axis=-1) @ wfc.T
        return p[..., :3] / p[..., 3:4]

    o = np.asarray(cam_pos, dtype=np.float64)
    far_pt = unproject(0.5)
    rd = far_pt - o
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    base = np.zeros((height, width, 4), np.uint8)
    nrm = np.zeros((height, width, 4), np.uint8)
    orm = np.zeros((height, width, 4), np.uint8)
    emi = np.zeros((height, width, 4), np.uint8)
    depth = np.ones((height, width), np.float32)
    tbest = np.full((height, width), np.inf)
    rad = 0.55
    for row in range(7):
        for col in range(7):
            c = np.array([(col - 3) * 1.3, 0.0, (3 - row) * 1.3])
            oc = o - c
            b = rd @ oc
            disc = b * b - (oc @ oc - rad * rad)
            hit = disc > 0
            t = -b - np.sqrt(np.where(hit, disc, 0))
            hit &= (t > 0) & (t < tbest)
            if not hit.any():
                continue
            P = o + rd * t[..., None]
            N = (P - c) / rad
            tbest = np.where(hit, t, tbest)
            clip = np.concatenate([P, np.ones_like(P[..., :1])], axis=-1) @ cam["clip_from_world"].T
            z = (clip[..., 2] / clip[..., 3]).astype(np.float32)
            depth[hit] = z[hit]
            nrm[hit, :3] = _encode_unorm8(N * 0.5 + 0.5)[hit]
            nrm[hit, 3] = 255
            base[hit, :3] = _encode_unorm8(PALETTE[(row + col) % 7])
            base[hit, 3] = 255
            orm[hit] = _encode_unorm8(np.array([1.0, max(col / 6.0, 0.0), row / 6.0, 1.0]))
    return dict(base=base, normal=nrm, orm=orm, emissive=emi, depth=depth, camera=cam,
                cam_pos=np.asarray(cam_pos, np.float64), ori_q=ori_q)


def _temple_block(args):
    width, height, seed, cam_pos, ori_q, y0, y1 = args
    cam = camera_matrices(np.asarray(cam_pos, dtype=np.float64), ori_q, aspect=width / height)
    n = y1 - y0
    base = np.zeros((n, width, 4), np.uint8); nrm = np.zeros((n, width, 4), np.uint8)
    orm = np.zeros((n, width, 4), np.uint8); emi = np.zeros((n, width, 4), np.uint8)
    depth = np.ones((n, width), np.float32)
    o = np.asarray(cam_pos, dtype=np.float64)
    wfc = cam["world_from_clip"]
    xs = (np.arange(width) + 0.5) / width * 2 - 1
    col_ang = np.arange(32) * (2 * np.pi / 32)
    col_c = np.stack([18 * np.cos(col_ang), 18 * np.sin(col_ang)], axis=-1)
    ys = (np.arange(y0, y1) + 0.5) / height * 2 - 1
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    p = np.stack([X, Y, np.full_like(X, 0.5), np.ones_like(X)], axis=-1) @ wfc.T
    rd = p[..., :3] / p[..., 3:4] - o
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    tb = np.full(X.shape, np.inf)
    N = np.zeros(X.shape + (3,))
    # ground
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -o[2] / rd[..., 2]
    hit = (rd[..., 2] < 0) & (t > 0) & (t < 400)
    tb = np.where(hit, t, tb)
    N[hit] = (0, 0, 1)
    # dome (sphere radius 60 centred at origin, seen from inside, only z>0 part)
    b = rd @ o
    disc = b * b - (o @ o - 60.0 ** 2)
    t = -b + np.sqrt(np.maximum(disc, 0))
    P = o + rd * t[..., None]
    hit = (disc > 0) & (t > 0) & (P[..., 2] > 0) & (t < tb)
    tb = np.where(hit, t, tb)
    N = np.where(hit[..., None], -P / 60.0, N)
    # columns (vertical cylinders radius 1.2, height 14)
    for c in col_c:
        oc = o[:2] - c
        a = (rd[..., :2] ** 2).sum(-1)
        bb = rd[..., :2] @ oc
        cc = oc @ oc - 1.2 ** 2
        disc = bb * bb - a * cc
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (-bb - np.sqrt(np.maximum(disc, 0))) / a
        P = o + rd * t[..., None]
        hit = (disc > 0) & (t > 0) & (t < tb) & (P[..., 2] > 0) & (P[..., 2] < 14)
        tb = np.where(hit, t, tb)
        n = np.concatenate([(P[..., :2] - c) / 1.2, np.zeros_like(P[..., :1])], axis=-1)
        N = np.where(hit[..., None], n, N)
    hit = np.isfinite(tb)
    P = o + rd * np.where(hit, tb, 0)[..., None]
    clip = np.concatenate([P, np.ones_like(P[..., :1])], axis=-1) @ cam["clip_from_world"].T
    z = (clip[..., 2] / clip[..., 3]).astype(np.float32)
    depth[:] = np.where(hit, z, 1.0)
    n1 = value_noise3(P * 0.7 + 3.0, seed)
    n2 = value_noise3(P * 2.9 + 11.0, seed ^ 0xABCDEF)
    n3 = value_noise3(P * 0.23 + 7.0, seed ^ 0x13579B)
    bc = np.stack([0.35 + 0.6 * n1, 0.3 + 0.55 * n2, 0.25 + 0.5 * n3], axis=-1)
    base[:, :, :3] = np.where(hit[..., None], _encode_unorm8(bc), 0)
    base[:, :, 3] = np.where(hit, 255, 0)
    nrm[:, :, :3] = np.where(hit[..., None], _encode_unorm8(N * 0.5 + 0.5), 0)
    nrm[:, :, 3] = np.where(hit, 255, 0)
    o3 = np.stack([np.ones_like(n1), 0.08 + 0.9 * n2, (n3 > 0.6).astype(np.float64), np.ones_like(n1)], axis=-1)
    orm[:] = np.where(hit[..., None], _encode_unorm8(o3), 0)
    em = (n1 * 7919.0 % 1.0 < 0.02) & hit
    emi[:, :, :3] = np.where(em[..., None], _encode_unorm8(np.stack([n2, 0.5 * n3, 0.2 * n1], axis=-1)), 0)
    return y0, y1, base, nrm, orm, emi, depth


def synth_gbuffer_temple(width=7680, height=4320, seed=0x5EED0005, cam_pos=(0.0, -30.0, 6.0), ori_q=None, rows=None, workers=1):
    """C5: ground plane z=0 + ring of 32 columns + dome, per-pixel material from value noise,
    2% emissive pixels.  Analytic ray casting in float64, evaluated row-block-wise.
    rows=(y0, y1) fills only that band of the full-size planes (the rest stays sky: depth 1, zero planes) -- what one
    rank of a screen-band split needs; workers > 1 spreads the row blocks over forked processes (same result)."""
    ori_q = DEFAULT_ORI if ori_q is None else ori_q
    cam = camera_matrices(np.asarray(cam_pos, dtype=np.float64), ori_q, aspect=width / height)
    base = np.zeros((height, width, 4), np.uint8)
    nrm = np.zeros((height, width, 4), np.uint8)
    orm = np.zeros((height, width, 4), np.uint8)
    emi = np.zeros((height, width, 4), np.uint8)
    depth = np.ones((height, width), np.float32)
    r0, r1 = rows if rows is not None else (0, height)
    blk = 135
    jobs = [(width, height, seed, tuple(cam_pos), ori_q, y0, min(r1, y0 + blk)) for y0 in range(r0, r1, blk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(min(workers, len(jobs)))
        try:
            parts = pool.map(_temple_block, jobs)
        finally:
            pool.close(); pool.join()
    else:
        parts = [_temple_block(j) for j in jobs]
    for y0, y1, b, n, o_, e, d in parts:
        base[y0:y1], nrm[y0:y1], orm[y0:y1], emi[y0:y1], depth[y0:y1] = b, n, o_, e, d
    return dict(base=base, normal=nrm, orm=orm, emissive=emi, depth=depth, camera=cam,
                cam_pos=np.asarray(cam_pos, np.float64), ori_q=ori_q)


def synth_lightgrid(size=128, seed=0x5EED00B7, lit=True):
    """A voxelised-scene stand-in for the light grid (render.cpp:678, RGBA16F [z][y][x][4], returned as float16):
    a ground slab, a few hollow boxes and pillars are occupied (alpha 1, rgb = the lit surface colour the
    voxelize pass would write); empty voxels carry either nothing (lit=False: the frame-0 clear) or light from
    earlier sweeps; a sprinkle of alpha == 0.5 voxels exercises the shader's two different alpha tests."""
    rng = np.random.default_rng(seed)
    n = size
    g = np.zeros((n, n, n, 4), np.float16)
    occ = np.zeros((n, n, n), bool)
    occ[: max(2, n // 32)] = True                                     # ground slab (z is the first axis)
    for _ in range(10):
        lo = rng.integers(0, n - n // 4, 3)
        ext = rng.integers(n // 16, n // 4, 3)
        hi = np.minimum(lo + ext, n)
        box = np.zeros_like(occ)
        box[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
        inner = np.zeros_like(occ)
        inner[lo[0] + 1:hi[0] - 1, lo[1] + 1:hi[1] - 1, lo[2] + 1:hi[2] - 1] = True
        occ |= box & ~inner                                           # shells, like rasterised surfaces
    for _ in range(12):
        y, x = rng.integers(0, n, 2)
        occ[: rng.integers(n // 8, n // 2), y:y + 2, x:x + 2] = True   # pillars
    colour = (rng.random((n, n, n, 3)) * np.array([2.5, 2.0, 1.5])).astype(np.float16)
    g[..., :3] = np.where(occ[..., None], colour, 0)
    g[..., 3] = occ
    if lit:
        glow = (rng.random((n, n, n, 3)) ** 3 * 3.0).astype(np.float16)
        g[..., :3] = np.where(occ[..., None], g[..., :3], glow)
    half = (rng.random((n, n, n)) < 0.002) & ~occ
    g[half, 3] = 0.5
    return g


def synth_post_inputs(seed, W, H):
    """Seeded inputs of the post-process tail: an HDR frame with a few very bright texels, a depth plane, smooth
    sub-pixel velocities with a patch of large motion (history rejection) and a strip that reprojects off-screen."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    lighting = np.zeros((H, W, 4), np.float16)
    base = 0.4 + 0.3 * np.sin(xx * 0.37) * np.cos(yy * 0.23)
    lighting[..., :3] = (base[..., None] * np.array([1.0, 0.8, 0.6]) + rng.random((H, W, 3)) * 0.2).astype(np.float16)
    hot = rng.random((H, W)) < 0.01
    lighting[hot, :3] = (rng.random((int(hot.sum()), 3)) * 300.0).astype(np.float16)
    lighting[..., 3] = 1.0
    history = lighting.copy()
    history[..., :3] = (history[..., :3].astype(np.float32) * (0.7 + 0.6 * rng.random((H, W, 3)))).astype(np.float16)
    depth = (0.9980 + 0.0015 * rng.random((H, W))).astype(np.float32)
    vel = np.zeros((H, W, 2), np.float16)
    vel[..., 0] = (0.004 * np.sin(yy * 0.11) + 0.0007).astype(np.float16)
    vel[..., 1] = (0.003 * np.cos(xx * 0.07)).astype(np.float16)
    vel[:, : max(2, W // 16), 0] = 0.5                      # reprojects to uv.x < 0: history rejected (:272-275)
    vel_prev = vel.copy()
    patch = (slice(H // 3, H // 2), slice(W // 3, W // 2))
    vel_prev[patch] = (vel_prev[patch].astype(np.float32) + 0.01).astype(np.float16)   # velocity-based rejection (:269-270)
    return lighting, depth, vel, vel_prev, history


def synth_sun_depth(size=2048, seed=0x5EED00E0):
    """Stand-in for the sun depth map the shadow raster pass writes (render.cpp:676, D32F): a smooth height field around
    the sun-space depth of the scene (0.45) plus a few sharp occluder rectangles, float32 [size][size]."""
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:size, 0:size].astype(np.float64) / size
    d = 0.45 + 0.3 * np.sin(7.0 * u + 1.0) * np.cos(5.0 * v) + 0.02 * np.sin(61.0 * u) * np.sin(47.0 * v)
    for _ in range(6):
        x0, y0 = rng.random(2) * 0.8
        w, h = 0.05 + rng.random(2) * 0.15
        d[(u > x0) & (u < x0 + w) & (v > y0) & (v < y0 + h)] = 0.05 + 0.2 * rng.random()
    return d.astype(np.float32)


GI_SCENE_CAMERA = (0.0, -5.5, 0.0)   # close enough that the sphere grid overflows the screen (rays leave it: the off-screen exit)
GI_SCENE_EXTENT = 8.0        # half-size of the world cube the test scene's light grid covers (the reference uses 40: render.cpp:960)


def synth_gi_scene(width, height, seed=0x5EED00F0, grid_size=128):
    """Inputs of the complete live lighting shader around the metal-rough-spheres G-buffer: a light grid with the spheres, a back
    wall and a floor voxelised into it (occupied voxels: alpha 1, lit surface colour; empty voxels: a smooth glow, as after a
    few sweeps), the 'previous frame' mip chain (half resolution, like bloom_downscale_rt) and a sun depth map with fine ripples
    around the scene's depth as the sun sees it.  Returns (gbuffer dict, grid float16 [n][n][n][4], prev levels, sun float32)."""
    rng = np.random.default_rng(seed)
    gbd = synth_gbuffer_spheres(width, height, cam_pos=GI_SCENE_CAMERA)
    n = grid_size
    c = ((np.arange(n) + 0.5) / n * 2 - 1) * GI_SCENE_EXTENT
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")                     # grid[z][y][x]
    grid = np.zeros((n, n, n, 4), np.float16)
    glow = 0.35 + 0.25 * np.sin(0.9 * X + 0.4) * np.cos(0.7 * Z) + 0.15 * np.sin(1.3 * Y)
    grid[..., 0] = (glow * 1.0).astype(np.float16); grid[..., 1] = (glow * 0.9).astype(np.float16); grid[..., 2] = (glow * 1.2).astype(np.float16)
    occ = np.zeros((n, n, n), bool)
    col = np.zeros((n, n, n, 3))
    for row in range(7):
        for cl in range(7):
            ctr = np.array([(cl - 3) * 1.3, 0.0, (3 - row) * 1.3])
            inside = (X - ctr[0]) ** 2 + (Y - ctr[1]) ** 2 + (Z - ctr[2]) ** 2 < 0.55 ** 2
            occ |= inside
            col[inside] = PALETTE[(row + cl) % 7] * 1.5
    for (bx, bz, half) in ((-3.9, 3.9, 1.3), (2.6, -2.6, 0.9)):            # two spheres sit inside solid blocks: no open point in 4 steps
        blk = (np.abs(X - bx) < half) & (np.abs(Y) < half) & (np.abs(Z - bz) < half)
        occ |= blk
        col[blk] = (0.5, 0.45, 0.4)
    wall = (Y > 2.0) & (Y < 2.6)
    floor = Z < -4.9
    occ |= wall | floor
    col[wall] = (0.8, 0.75, 0.6); col[floor] = (0.3, 0.35, 0.3)
    grid[occ, :3] = col[occ].astype(np.float16)
    grid[..., 3] = occ
    soft = (rng.random((n, n, n)) < 0.003) & ~occ                     # a few half-dense voxels around the 0.3 alpha thresholds
    grid[soft, 3] = rng.choice([0.25, 0.35, 0.5], int(soft.sum())).astype(np.float16)
    # previous frame: half-resolution RGBA16F image + box-filtered mips
    pw, ph = max(1, width // 2), max(1, height // 2)
    yy, xx = np.mgrid[0:ph, 0:pw].astype(np.float64)
    img = np.stack([0.6 + 0.4 * np.sin(xx * 0.31), 0.5 + 0.4 * np.cos(yy * 0.27), 0.4 + 0.3 * np.sin((xx + yy) * 0.19), np.ones_like(xx)], -1)
    levels = [img.astype(np.float16)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        a = levels[-1].astype(np.float32)
        h2, w2 = max(1, a.shape[0] // 2), max(1, a.shape[1] // 2)
        a = a[: h2 * 2 if a.shape[0] > 1 else 1, : w2 * 2 if a.shape[1] > 1 else 1]
        if a.shape[0] > 1:
            a = 0.5 * (a[0::2] + a[1::2])
        if a.shape[1] > 1:
            a = 0.5 * (a[:, 0::2] + a[:, 1::2])
        levels.append(a.astype(np.float16))
        if len(levels) == 6:
            break
    v, u = np.mgrid[0:512, 0:512] / 512.0
    sun = (0.5 + 0.08 * np.sin(2 * np.pi * 40 * u) * np.cos(2 * np.pi * 36 * v)).astype(np.float32)
    return gbd, grid, levels, sun
