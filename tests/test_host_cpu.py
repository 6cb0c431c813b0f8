"""CPU tests (no GPU) of the product's host side: the C ABI exports every declared symbol, the host tables,
the RGBE decoder, the camera/Globals code, the work partitioner (incl. a 2-rank gloo run)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import pbrhip
    if not os.path.exists(pbrhip.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(pbrhip.PKG_ROOT, "csrc"), "-j", "8"], stdout=subprocess.DEVNULL)
    return pbrhip.lib()


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#\s*define.*$", "", text, flags=re.M)
    names = set(re.findall(r"\b((?:GPU|GPUX|PBR|pbrk)_[A-Za-z0-9_]+)\s*\(", text))
    return {n for n in names if not n.startswith(("GPU_STR", "GPU_LangAgnosticLiteral"))}


def test_library_exports_every_declared_symbol(L):
    """include/*.h <-> libgpu_hip.so <-> the ctypes prototype table agree (no compute calls)."""
    import pbrhip
    inline = {"GPU_Read", "GPU_Write", "GPU_ReadWrite", "GPU_GetFormatInfo"}
    declared = set()
    for h in ("gpu_hip.h", "gpux.h", "pbr_host.h", "pbr_kernels.h"):
        declared |= declared_functions(h)
    declared -= inline
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in include/ but not exported by libgpu_hip.so"
        assert name in pbrhip.PROTOTYPES, f"{name} has no ctypes prototype"
    assert L.GPUX_BackendName() == b"hip-gfx950"
    out = subprocess.check_output(["nm", "-D", "--defined-only", pbrhip.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported


def test_header_enums_keep_reference_numbering():
    """Enum values are ABI (callers pass them by value)."""
    src = "#include \"gpux.h\"\n#include <stdio.h>\nint main(){printf(\"%d %d %d %d %d %d %d %d %zu %zu %zu\\n\", GPU_Format_RGBA8UN, GPU_Format_RG16F," \
          " GPU_Format_RGBA16F, GPU_Format_RGBA32F, GPU_Format_D32F_Or_X8D24UN, GPU_Format_BC5_UN, GPU_TextureFlag_Cubemap, GPU_BufferFlag_StorageBuffer," \
          " sizeof(GPU_Texture), sizeof(GPU_Buffer), sizeof(GPU_ShaderDesc)); return 0;}\n"
    exe = "/tmp/pbr_enum_check"
    subprocess.run(["gcc", "-std=c11", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    vals = subprocess.check_output([exe], text=True).split()
    assert vals == ["3", "6", "8", "12", "23", "29", "8", "4", "28", "16", "80"]


def test_host_tables_match_oracle(L):
    import pbr_oracle as O
    n = 8192
    ang = np.zeros((n, 4), np.float32)
    L.pbrk_host_sample_angles(n, ang.ctypes.data_as(C.c_void_p))
    ref = np.zeros((n, 4), np.float32)
    O.lib().orc_sample_angles(n, ref.reshape(-1))
    assert np.array_equal(ang, ref)
    for rough, nz_expected in ((0.03, 1389), (0.15, 8192), (0.4, 8192), (0.6, 8192)):
        tab = np.zeros((n, 4), np.float32)
        alpha = C.c_float()
        cnt = L.pbrk_host_prefilter_table(n, C.c_float(rough), tab.ctypes.data_as(C.c_void_p), C.byref(alpha))
        assert cnt == nz_expected                         # only i <= 1388 non-zero at m = .03
        D = np.zeros(n, np.float32)
        O.lib().orc_prefilter_D(n, rough, D)
        dw = np.float32(2.0) * np.float32(3.14159265358979323846) / np.float32(n)
        w = (D * ref[:, 0]) * dw
        nz = np.nonzero(w)[0]
        assert len(nz) == cnt and np.array_equal(tab[:cnt, 3], w[nz])
        assert np.allclose(np.linalg.norm(tab[:cnt, :3], axis=1), 1.0, atol=1e-6)
        # alpha = what the shader accumulates in its alpha lane
        want_alpha = O.prefilter_mip(None, 1, 16, 1, roughness=rough, faces=(0, 1), rows=(0, 1))[0, 0, 0, 3]
        assert alpha.value == want_alpha
    irr = np.zeros((1024, 4), np.float32)
    assert L.pbrk_host_irradiance_table(1024, irr.ctypes.data_as(C.c_void_p)) == 1024
    assert np.allclose(irr[:, 3], 1 - np.arange(1024) / 1024.0, atol=3e-7) and np.array_equal(irr[:, 2], irr[:, 3])


def test_pyramid_layout_helpers(L):
    import pbr_oracle as O
    for W in (1, 2, 64, 2048):
        levels = L.pbrk_mip_count(W, W)
        assert levels == O.mip_count(W)
        for l in range(levels + 1):
            assert L.pbrk_level_offset(W, l) * 4 == O.level_offset(W, l)
        assert L.pbrk_bordered_pyramid_texels(W, levels) == sum(6 * (max(1, W >> l) + 2) ** 2 for l in range(levels))


def test_product_rgbe_decoder(L, golden_dir):
    """PBR_DecodeHDR (host/pbr_rgbe.c) against the reference's stb_image output and hand-made edge cases."""
    from pbrhip import synth
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    def decode(data):
        w, h, err = C.c_int(), C.c_int(), C.c_char_p()
        p = L.PBR_DecodeHDR(data, len(data), C.byref(w), C.byref(h), C.byref(err))
        if not p:
            return None, err.value
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h.value, w.value, 4)).copy()
        libc.free(p)
        return arr, None

    z = np.load(os.path.join(golden_dir, "ref_stb_hdr_decode.npz"))
    for key in ("file_rle", "file_flat"):
        got, err = decode(z[key].tobytes())
        assert err is None and np.array_equal(got, z["decoded"])
    rng = np.random.default_rng(11)
    px = rng.integers(0, 256, (5, 4, 4), dtype=np.uint8)          # width < 8: flat only
    got, _ = decode(synth.hdr_file_bytes(px, rle=True))
    assert np.array_equal(got, synth.rgbe_decode(px))
    assert decode(b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 1 +X 1\n\0\0\0\0")[0] is None       # unsupported format
    assert decode(b"P6\n1 1\n255\n\0\0\0")[0] is None                                           # not an HDR
    assert decode(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 1 +X 1\n\0\0\0\0")[0] is None       # unsupported orientation
    bad = bytearray(synth.hdr_file_bytes(rng.integers(0, 256, (2, 16, 4), dtype=np.uint8), rle=True))
    bad[-40] = 0xFF                                              # corrupt a run length
    arr, err = decode(bytes(bad))
    assert arr is None or arr.shape == (2, 16, 4)                # must not crash; error or garbage-but-bounded


def test_fill_globals_matches_reference_handmade_math(L, golden_dir):
    import pbrhip
    z = np.load(os.path.join(golden_dir, "ref_globals_poses.npz"))
    for k in range(len(z["globals"])):
        fov, aspect, near, far, sx, sy, frame = z["params"][k]
        ori = None if z["default_ori"][k] else z["ori"][k]
        g = pbrhip.fill_globals(z["pos"][k], ori, fov, aspect, near, far, (sx, sy), int(frame))
        got = np.frombuffer(bytes(g), np.float32)[:137]
        want = z["globals"][k][:137]
        # bit for bit: same operation order as the reference's math library (world_space_from_clip feeds the sky test)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0])
    g0 = pbrhip.fill_globals((0, 0, 5))
    assert np.allclose(list(g0.sun_direction)[:3], (-0.827670276, -0.101625085, -0.551936984), atol=1e-6)
    wfc = np.array(list(g0.world_space_from_clip)).reshape(4, 4)
    assert np.allclose(wfc[0], (1.36413682, 0, 0, 0), atol=1e-6) and np.allclose(wfc[3], (0, 0.999999881, 250.000015, 50.0000038), rtol=1e-6)


def units_of(specular, min_size, irr, env, world, rank):
    import pbrhip
    arr, n = pbrhip.partition(specular, min_size, irr, env, world, rank)
    return [(arr[i].kind, arr[i].mip, arr[i].face0, arr[i].face1, arr[i].row0, arr[i].row1, arr[i].cost) for i in range(n)]


def test_partition_covers_every_texel_once_and_balances(L):
    for (spec, irr, world) in ((4096, 128, 8), (512, 32, 2), (64, 16, 3), (256, 32, 1)):
        mips = int(np.log2(spec)) + 1
        cover = {("p", m): np.zeros((6, max(1, spec >> m)), np.int32) for m in range(mips)}
        cover[("i", 0)] = np.zeros((6, irr), np.int32)
        loads = []
        for r in range(world):
            us = units_of(spec, 1, irr, 2048, world, r)
            loads.append(sum(u[6] for u in us))
            for (kind, mip, f0, f1, r0, r1, cost) in us:
                key = ("i", 0) if kind == 1 else ("p", mip)
                cover[key][f0:f1, r0:r1] += 1
        for k, c in cover.items():
            assert np.all(c == 1), (spec, world, k)
        # the partition balances measured TIME (per-level and per-face weights of up to 1.6x around the plain sample count and a
        # copy level worth ~6 evaluations per texel), so the plain counts agree only roughly
        assert max(loads) <= 1.3 * (sum(loads) / world), (spec, world, loads)
        # rank < 0 lists everything; partition is deterministic
        assert len(units_of(spec, 1, irr, 2048, world, -1)) == sum(len(units_of(spec, 1, irr, 2048, world, r)) for r in range(world))
        assert units_of(spec, 1, irr, 2048, world, 0) == units_of(spec, 1, irr, 2048, world, 0)
    # communication-aware: the copy level stays on rank 0 (where results are gathered), rank 0 holds most of the byte-heavy mip 1, the
    # other ranks send about the same number of bytes each, and nobody gets more than a handful of dispatches
    sent = []
    for r in range(8):
        us = units_of(4096, 1, 128, 2048, 8, r)
        assert len(us) <= 10
        assert all(r == 0 for u in us if u[0] == 0 and u[1] == 0) and (r != 0 or any(u[0] == 0 and u[1] == 0 for u in us))
        sent.append(sum((u[3] - u[2]) * (u[5] - u[4]) * (128 if u[0] == 1 else 4096 >> u[1]) * 16 for u in us if not (u[0] == 0 and u[1] == 0)))
    assert max(sent[1:]) <= 1.25 * min(sent[1:]) and sent[0] > 2 * max(sent[1:])
    # reference stop rule: `if (size < 16) break;` (render.cpp:566)
    assert {u[1] for u in units_of(256, 16, 0, 256, 1, 0)} == {0, 1, 2, 3, 4}


def test_partition_property_sweep(L):
    """Every texel of every level exactly once for many (size, irradiance, stop size, world) combinations; multi-face units are
    whole faces (so that each unit is one contiguous byte range of the [mip][face][y][x] layout, which the gather relies on)."""
    import pbrhip
    for spec in (2048, 512, 64, 16, 8):
        for irr in (0, 16, 128):
            for min_size in (1, 16):
                for world in (1, 2, 3, 5, 7, 8, 12, 16):
                    mips = int(np.log2(spec)) + 1
                    cover = {("p", m): np.zeros((6, max(1, spec >> m)), np.int32) for m in range(mips) if (spec >> m) >= min_size}
                    if irr:
                        cover[("i", 0)] = np.zeros((6, irr), np.int32)
                    loads = []
                    for r in range(world):
                        u, n = pbrhip.partition(spec, min_size, irr, 2048, world, r)
                        loads.append(sum(x.cost for x in u[:n]))
                        for x in u[:n]:
                            key = ("i", 0) if x.kind == 1 else ("p", x.mip)
                            size = cover[key].shape[1]
                            assert x.face0 < x.face1 <= 6 and x.row0 < x.row1 <= size
                            assert x.face1 - x.face0 == 1 or (x.row0 == 0 and x.row1 == size)
                            cover[key][x.face0:x.face1, x.row0:x.row1] += 1
                    assert all(np.all(c == 1) for c in cover.values()), (spec, irr, min_size, world)
                    if spec >= 512 and world <= 8:
                        assert max(loads) <= 1.3 * sum(loads) / world, (spec, irr, min_size, world)    # plain counts; the balance is in measured time


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "vulkan-pbr-renderer_amd", "python")); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import numpy as np, torch, torch.distributed as dist
import pbrhip, pbr_oracle as O
from pbrhip import synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, spec, irr = 32, 32, 8
L = pbrhip.lib()
O.set_threads(1)
env = synth.synth_env(W, seed=0x5EED00AA)
pyr = O.build_pyramid(env)
mips = L.pbrk_mip_count(spec, spec)
total = L.pbrk_pyramid_texels(spec, mips) * 4
mem = torch.zeros(total, dtype=torch.float32)                       # the specular pyramid, [mip][face][y][x][4]
imem = torch.zeros(6 * irr * irr * 4, dtype=torch.float32)          # the irradiance cube
def sl(u):                                                          # the byte range PBR_UnitByteRange yields, as a float slice
    if u.kind == pbrhip.Unit_Irradiance:
        size, base, m = irr, 0, imem
    else:
        size, base, m = max(1, spec >> u.mip), L.pbrk_level_offset(spec, u.mip) * 4, mem
    return m[base + ((u.face0 * size + u.row0) * size) * 4: base + (((u.face1 - 1) * size + u.row1) * size) * 4]
def compute(u):                                                     # stand-in for the kernels: the oracle on exactly this unit
    if u.kind == pbrhip.Unit_Irradiance:
        full = O.irradiance(pyr, W, irr, faces=(u.face0, u.face1), rows=(u.row0, u.row1))
    else:
        full = O.prefilter_mip(pyr, W, spec, u.mip, faces=(u.face0, u.face1), rows=(u.row0, u.row1))
    if u.face1 == u.face0 + 1:
        return np.ascontiguousarray(full[u.face0, u.row0:u.row1]).ravel()
    return np.ascontiguousarray(full[u.face0:u.face1]).ravel()
mine, n = pbrhip.partition(spec, 1, irr, W, world, rank)
for i in range(n):
    sl(mine[i]).copy_(torch.from_numpy(compute(mine[i])))
ops = []
if rank == 0:
    for r in range(1, world):
        us, m = pbrhip.partition(spec, 1, irr, W, world, r)
        ops += [dist.P2POp(dist.irecv, sl(us[i]), r) for i in range(m)]
else:
    ops = [dist.P2POp(dist.isend, sl(mine[i]), 0) for i in range(n)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
if rank == 0:                                                       # gathered == the whole job computed in one piece, bit for bit
    want = np.concatenate([O.prefilter_mip(pyr, W, spec, m).ravel() for m in range(mips)])
    assert np.array_equal(mem.numpy().view(np.uint32), want.view(np.uint32)), "gathered specular pyramid differs from the single-rank job"
    assert np.array_equal(imem.numpy().view(np.uint32), O.irradiance(pyr, W, irr).ravel().view(np.uint32)), "gathered irradiance differs"
    print("GATHER_OK", int(total))
dist.barrier()
# --- the overlapped form (PBR_RunPartitionedIBL's schedule): the units of the early levels (PBR_SelectUnits, mask = mip 1) are
# computed first and are on the wire while the rest is computed; two grouped exchanges, every rank issues them in the same order
import ctypes as C
mem.zero_(); imem.zero_()
EARLY = 0x2
def select(us, m, mask):
    out = (pbrhip.PBR_WorkUnit * max(1, m))()
    k = L.PBR_SelectUnits(us, m, mask & 0xFFFFFFFF, out)
    return [out[i] for i in range(k)]
early, late = select(mine, n, EARLY), select(mine, n, ~EARLY)
assert len(early) + len(late) == n and all(u.kind == pbrhip.Unit_Prefilter and u.mip == 1 for u in early)
def exchange(mask, my_units):
    if rank == 0:
        ops = []
        for r in range(1, world):
            us, m = pbrhip.partition(spec, 1, irr, W, world, r)
            ops += [dist.P2POp(dist.irecv, sl(u), r) for u in select(us, m, mask)]
    else:
        ops = [dist.P2POp(dist.isend, sl(u), 0) for u in my_units]
    return dist.batch_isend_irecv(ops) if ops else []
for u in early:
    sl(u).copy_(torch.from_numpy(compute(u)))
pending = exchange(EARLY, early)                                    # in flight ...
for u in late:                                                      # ... while the rest is computed
    sl(u).copy_(torch.from_numpy(compute(u)))
pending += exchange(~EARLY, late)
for w in pending:
    w.wait()
if rank == 0:
    assert np.array_equal(mem.numpy().view(np.uint32), want.view(np.uint32)), "two-phase gather: specular pyramid differs"
    assert np.array_equal(imem.numpy().view(np.uint32), O.irradiance(pyr, W, irr).ravel().view(np.uint32)), "two-phase gather: irradiance differs"
    print("OVERLAP_OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_gloo_gather(L, tmp_path):
    """The N>1 data path (PBR_PartitionIBL -> per-rank units -> one grouped send/recv gather of the units' byte ranges to rank 0)
    on CPU with gloo standing in for RCCL and the oracle standing in for the kernels: the gathered maps equal the whole job
    computed in one piece, bit for bit -- once as one exchange after all units, once in the two overlapped phases of
    PBR_RunPartitionedIBL (early levels on the wire while the late ones are computed)."""
    script = tmp_path / "gloo_worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GATHER_OK" in out.stdout and "OVERLAP_OK" in out.stdout


def test_unorm8_decode_trick():
    """k_shade.hip decodes unorm8 as fma(fma(-255, q, b), rc, q) with q = b*rc, rc = fl(1/255): equal to the correctly
    rounded b/255.0f (what the shader's texture fetch returns) for all 256 inputs, while b*rc alone is not."""
    b = np.arange(256, dtype=np.float32)
    want = b / np.float32(255.0)
    rc = np.float32(1.0) / np.float32(255.0)
    q = b * rc
    r = (b.astype(np.float64) - 255.0 * q.astype(np.float64)).astype(np.float32)          # fma(-255, q, b): exact
    q2 = (r.astype(np.float64) * np.float64(rc) + q.astype(np.float64)).astype(np.float32)  # fma(r, rc, q)
    assert np.array_equal(q2, want)
    assert (q != want).sum() > 0


def test_hdr_writer_round_trip(L):
    """PBR_EncodeHDR (extension): what it writes is decoded identically by the product decoder and by the oracle, for planar
    (8 <= w < 32768) and flat (w < 8) containers, including a scanline that starts with the bytes (2, 2, x<128)."""
    import pbr_oracle as O
    from pbrhip import synth
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(21)
    for (h, w) in ((5, 4), (3, 8), (4, 200)):
        rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        rgbe[..., 3] = rng.integers(100, 160, (h, w))
        rgbe[0, 0] = (2, 2, 5, 130)                       # would read as an RLE marker in a flat file
        img = synth.rgbe_decode(rgbe)                     # exactly representable values
        n = C.c_size_t()
        p = L.PBR_EncodeHDR(img.ctypes.data_as(C.c_void_p), w, h, C.byref(n))
        assert p and n.value > 0
        data = C.string_at(p, n.value)
        libc.free(p)
        back = O.rgbe_decode(data)
        # the writer normalises the shared exponent (largest mantissa in [128,255]); values survive exactly when the
        # original mantissas had their top bit set, and within one mantissa step otherwise
        assert back.shape == img.shape
        assert np.allclose(back[..., :3], img[..., :3], rtol=2.0 ** -7, atol=0)
        again = L.PBR_EncodeHDR(back.ctypes.data_as(C.c_void_p), w, h, C.byref(n))
        data2 = C.string_at(again, n.value); libc.free(again)
        assert np.array_equal(O.rgbe_decode(data2), back)  # idempotent on its own output


def test_hdr_codec_under_sanitizers(tmp_path):
    """The file-facing host code (Radiance .hdr decoder / encoder, host/pbr_rgbe.c + pbr_hdrio.c) built with ASan + UBSan on the CPU
    (GPU sanitizers are not available): 20 000 encode -> decode round trips and 20 000 truncated / bit-flipped / spliced files."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = str(tmp_path / "fuzz_hdr")
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-w",
           "-I" + os.path.join(root, "include"), os.path.join(here, "sanitize", "fuzz_hdr.c"), os.path.join(here, "sanitize", "gpu_stubs.c"),
           os.path.join(root, "vulkan-pbr-renderer_amd", "host", "pbr_rgbe.c"), os.path.join(root, "vulkan-pbr-renderer_amd", "host", "pbr_hdrio.c"),
           "-o", exe, "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout[-500:], r.stderr[-3000:])


def test_partition_and_exchange_plan_under_sanitizers(tmp_path):
    """The multi-GPU host logic (host/pbr_ibl.c partitioner, host/pbr_gather.c byte ranges / phase selection / exchange plans) built
    with ASan + UBSan on the CPU: 1500 random (sizes, world) cases -- exact ownership of every row of every level, unit ranges inside
    their textures, root's receives == the peers' sends, the two phases of the overlapped exchange a partition of the one-shot plan.
    Backend entry points the logic never reaches are linked as aborting stubs generated from the objects' undefined symbols."""
    import shutil
    import subprocess
    if not shutil.which("gcc") or not shutil.which("nm"):
        pytest.skip("no gcc / nm")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    flags = ["-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-w",
             "-I" + os.path.join(root, "include")]
    srcs = [os.path.join(here, "sanitize", "fuzz_partition.c"), os.path.join(root, "vulkan-pbr-renderer_amd", "host", "pbr_gather.c"),
            os.path.join(root, "vulkan-pbr-renderer_amd", "host", "pbr_ibl.c")]
    objs = []
    for src in srcs:
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        b = subprocess.run(["gcc"] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if b.returncode != 0 and "sanitize" in b.stderr:
            pytest.skip("toolchain without sanitizer runtimes")
        assert b.returncode == 0, b.stderr[-2000:]
        objs.append(obj)
    defined, undefined = set(), set()
    for obj in objs:
        for line in subprocess.run(["nm", obj], capture_output=True, text=True, check=True).stdout.splitlines():
            parts = line.split()
            if len(parts) == 2 and parts[0] == "U":
                undefined.add(parts[1])
            elif len(parts) == 3 and parts[1] in "TD":
                defined.add(parts[2])
    need = sorted(x for x in undefined - defined if x.startswith(("GPU_", "GPUX_", "pbrk_", "PBR_", "nccl")))
    stubs = tmp_path / "stubs.c"
    stubs.write_text("#include <stdlib.h>\n" + "".join(f"void {x}(void) {{ abort(); }}\n" for x in need))
    exe = str(tmp_path / "fuzz_partition")
    b = subprocess.run(["gcc"] + flags + objs + [str(stubs), "-o", exe, "-lm"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout[-500:], r.stderr[-3000:])


def test_c_gather_runs_at_world_2_3_5_8_against_stub_rccl(tmp_path):
    """the REAL C exchange code (host/pbr_gather.c: PBR_GatherUnits, the two overlapped phases of
    PBR_RunPartitionedIBL, PBR_GatherBands; host/pbr_ibl.c: PBR_PartitionIBL + PBR_RecordUnits) executed by N = 2, 3, 5, 8
    PROCESSES on the CPU, under ASan + UBSan.  RCCL is replaced at its run-time binding point (PBR_SetRcclLibrary) by
    tests/sanitize/nccl_stub.c (grouped send / recv over socket pairs, RCCL's matching rules), the kernels by a fake backend that
    writes value(level, face, y, x) into exactly the rows a recorded dispatch covers.  Root: gathered maps == that function on every
    texel (bit for bit), byte counts == the plan; peers: nothing but their own rows changed.  `failsend`: an ncclSend that fails
    inside the group leaves GroupStart / GroupEnd balanced and the thread usable."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    flags = ["-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-w",
             "-I" + os.path.join(root, "include")]
    stub = str(tmp_path / "libnccl_stub.so")
    exe = str(tmp_path / "gather_world")
    b = subprocess.run(["gcc"] + flags + ["-shared", "-fPIC", os.path.join(here, "sanitize", "nccl_stub.c"), "-o", stub], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-2000:]
    host = os.path.join(root, "vulkan-pbr-renderer_amd", "host")
    b = subprocess.run(["gcc"] + flags + [os.path.join(here, "sanitize", "gather_world.c"), os.path.join(host, "pbr_gather.c"),
                                         os.path.join(host, "pbr_ibl.c"), "-o", exe, "-ldl", "-lm"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    cases = [(w, 256, 32, 1, m) for w in (2, 3, 5, 8) for m in ("units", "overlap", "bands")]
    cases += [(1, 256, 32, 1, "units"), (1, 256, 32, 1, "overlap"), (3, 512, 0, 16, "overlap"), (2, 1024, 128, 1, "units"),
              (3, 1024, 128, 1, "overlap"), (1, 64, 0, 1, "failsend")]
    for (w, spec, irr, mn, mode) in cases:
        r = subprocess.run([exe, stub, str(w), str(spec), str(irr), str(mn), mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), ((w, spec, irr, mn, mode), r.stdout[-500:], r.stderr[-3000:])


def test_rccl_is_bound_at_first_use_not_at_link_time():
    """libgpu_hip.so carries no librccl dependency (a single-GPU consumer neither loads nor needs it); the exchange
    binds the library at first use and reports which file it took."""
    import subprocess
    import pbrhip
    out = subprocess.run(["readelf", "-d", pbrhip.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0:
        assert "rccl" not in out.stdout and "nccl" not in out.stdout, out.stdout
    L = pbrhip.lib()
    assert L.PBR_SetRcclLibrary(b"/nonexistent/librccl.so") == -2            # PBR_E_COMM, loudly
    assert L.PBR_ExchangeRanges(C.c_void_p(1), None, (pbrhip.PBR_XferRange * 1)(pbrhip.PBR_XferRange(1, 16, 0)), 1, None, 0) == -2
    assert L.PBR_SetRcclLibrary(None) == 0                                    # back to the default search
    if os.path.exists("/opt/rocm/lib/librccl.so.1"):
        ver, path = pbrhip.rccl_info()
        assert ver >= 20000 and "rccl" in path, (ver, path)


def test_bench_dry_launch_prints_worker_command():
    """`bench.py --gpus N --dry-launch` (no GPU, no torch import): the command the self-launch would start, rc 0."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--dry-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["cmd"]
    assert d["n_workers"] == 8 and "torch.distributed.run" in cmd and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and "--dry-launch" not in cmd and cmd[-5].endswith("bench.py")
    # a launcher world that contradicts --gpus is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env2)
    assert r.returncode == 2 and "must agree" in r.stderr


def test_reference_callers_compile_against_the_boundary(tmp_path):
    """VERDICT r2 item 6a (container only: skipped where /root/reference is absent; nothing of the reference is copied or shipped):
    the reference's own callers of the path -- src/demo_pbr_renderer/render.cpp with HotreloadShaders :505-619, InitRenderer
    :794-871 and BuildRenderCommands :1117-1127 -- must keep compiling (g++ -fsyntax-only) against include/gpu.h put in the place of
    src/gpu/gpu.h.  The include tree is a scratch directory of symlinks ("gpu/gpu.h" -> OUR header, "fire/" -> the reference's
    src/Fire, whose file names differ in case from its #includes on a case-sensitive file system); the only additions are
    declarations of the three MSVC-CRT functions that Fire's headers call (_aligned_free, _aligned_realloc, strcpy_s)."""
    import shutil
    ref = "/root/reference"
    src = os.path.join(ref, "src", "demo_pbr_renderer", "render.cpp")
    if not os.path.exists(src) or not shutil.which("g++"):
        pytest.skip("reference tree (or g++) not present: container-only test")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = tmp_path / "inc"
    (inc / "gpu").mkdir(parents=True)
    os.symlink(os.path.join(root, "include", "gpu.h"), inc / "gpu" / "gpu.h")
    os.symlink(os.path.join(ref, "src", "Fire"), inc / "fire")
    shim = tmp_path / "msvc_crt_decls.h"
    shim.write_text("#include <stddef.h>\nextern \"C\" { void _aligned_free(void*); void* _aligned_realloc(void*, size_t, size_t); "
                    "int strcpy_s(char*, size_t, const char*); }\n")
    # -fpermissive: Fire's fire_ds.h:144 relies on MSVC's lenient two-phase lookup (a reference-side header, not the boundary)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-fpermissive", "-I" + str(inc), "-I" + os.path.join(root, "include"), "-I" + os.path.join(ref, "src"),
           "-I" + os.path.join(ref, "third_party"), "-include", str(shim), src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    # the check can fail: the same command against a header that lacks one of the entry points render.cpp calls
    broken = tmp_path / "broken"
    (broken / "gpu").mkdir(parents=True)
    text = open(os.path.join(root, "include", "gpu_hip.h")).read()
    assert "GPU_OpDispatch(" in text
    (broken / "gpu" / "gpu.h").write_text(text.replace("GPU_OpDispatch(", "GPU_OpDispatch_REMOVED("))
    os.symlink(os.path.join(ref, "src", "Fire"), broken / "fire")
    cmd2 = [c if c != "-I" + str(inc) else "-I" + str(broken) for c in cmd]
    cmd2 = [c for c in cmd2 if c != "-I" + os.path.join(root, "include")] + ["-I" + os.path.join(root, "include")]
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "GPU_OpDispatch" in r2.stderr


def test_public_headers_are_self_contained(tmp_path):
    """Every header under include/ compiles on its own as C11 (pedantic) and as C++17: what a reference-side build would include."""
    import shutil
    import subprocess
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    for h in sorted(os.listdir(inc)):
        for cc, std, ext in (("gcc", "-std=c11", ".c"), ("g++", "-std=c++17", ".cpp")):
            src = tmp_path / ("t" + ext)
            src.write_text(f'#include "{h}"\nint main(void) {{ return 0; }}\n')
            r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + inc, str(src)], capture_output=True, text=True)
            assert r.returncode == 0, (h, cc, r.stderr[-1500:])
