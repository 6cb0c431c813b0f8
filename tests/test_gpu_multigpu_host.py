"""The multi-GPU host layer on a one-GPU lease: the C gather (host/pbr_gather.c) on a 1-rank RCCL communicator, submission
order between graphs in flight, and invalidation of sampler twins after writes the backend cannot see."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _maps_32(L, pbrhip):
    from pbrhip import synth
    env = synth.synth_env(32, seed=0x5EED00AA)
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 32, 32, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 8, 64, 32)
    L.PBR_GenPrefilteredEnvMap(tex, maps.tex_specular_env_map, 1)
    L.PBR_GenIrradianceMap(tex, maps.irradiance_map)
    return tex, maps


def test_rccl_gather_on_single_rank_communicator(gpu):
    """PBR_ExchangeRanges (ncclGroupStart / ncclRecv / ncclSend / ncclGroupEnd) on a 1-rank communicator: every unit a rank of a
    2- and 3-way split would send travels to self into a second buffer and arrives bit-identical; PBR_UnitByteRange addresses
    exactly the unit's texels; the world == 1 gather is a no-op."""
    import pbrhip
    L = gpu
    tex, maps = _maps_32(L, pbrhip)
    comm = pbrhip.rccl_comm_init(1, 0, pbrhip.rccl_unique_id())
    try:
        g = L.GPU_MakeGraph()
        stream = L.GPUX_GraphStream(g)
        assert L.PBR_GatherUnits(comm, stream, 0, 1, 0, C.byref(maps), 1, 32) == 0
        for (t, name) in ((maps.tex_specular_env_map, "spec"), (maps.irradiance_map, "irr")):
            nbytes = L.GPUX_TextureTotalBytes(t)
            dst = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_GPU, None)
            host = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_CPU, None)
            zero = np.zeros(nbytes, np.uint8)
            zbuf = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_CPU, zero.ctypes.data_as(C.c_void_p))
            L.GPU_OpCopyBufferToBuffer(g, zbuf, dst, 0, 0, nbytes); L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
            src_ptr = L.GPUX_TextureDevicePtr(t, 0)
            dst_ptr = L.GPUX_BufferDevicePtr(dst)
            sends, recvs, covered = [], [], np.zeros(nbytes, bool)
            for world in (2, 3):
                for rank in range(1, world):
                    units, n = pbrhip.partition(32, 1, 8, 32, world, rank)
                    for i in range(n):
                        u = units[i]
                        if (u.kind == pbrhip.Unit_Irradiance) != (name == "irr"):
                            continue
                        tp = pbrhip.TexP(); off = C.c_uint64(); nb = C.c_uint64()
                        assert L.PBR_UnitByteRange(C.byref(maps), C.byref(u), C.byref(tp), C.byref(off), C.byref(nb)) == 0
                        # the same range from the layout definition ([mip][face][y][x], 16 B texels)
                        size = 8 if name == "irr" else max(1, 32 >> u.mip)
                        base = 0 if name == "irr" else L.pbrk_level_offset(32, u.mip) * 16
                        a = base + ((u.face0 * size + u.row0) * size) * 16
                        b = base + (((u.face1 - 1) * size + u.row1) * size) * 16
                        assert (off.value, nb.value) == (a, b - a)
                        if covered[a:b].any():
                            continue                           # ranges of the 2-way and the 3-way split overlap: send each byte once
                        covered[a:b] = True
                        sends.append(pbrhip.PBR_XferRange(src_ptr + a, b - a, 0)); recvs.append(pbrhip.PBR_XferRange(dst_ptr + a, b - a, 0))
            assert sends
            S = (pbrhip.PBR_XferRange * len(sends))(*sends); R = (pbrhip.PBR_XferRange * len(recvs))(*recvs)
            assert L.PBR_ExchangeRanges(comm, stream, S, len(sends), R, len(recvs)) == 0
            L.GPU_OpCopyBufferToBuffer(g, dst, host, 0, 0, nbytes)           # same stream: follows the exchange
            L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
            got = np.frombuffer((C.c_char * nbytes).from_address(host.contents.data), np.uint8)
            want = np.concatenate([pbrhip.read_mip(t, m).view(np.uint8).ravel() for m in range(t.contents.mip_level_count)])
            assert np.array_equal(got[covered], want[covered]) and not got[~covered].any() and covered.any()
            for b_ in (dst, host, zbuf):
                L.GPU_DestroyBuffer(b_)
        # argument checking happens before anything is enqueued
        bad = (pbrhip.PBR_XferRange * 1)(pbrhip.PBR_XferRange(None, 16, 0))
        assert L.PBR_ExchangeRanges(comm, stream, bad, 1, None, 0) == -1
        assert L.PBR_GatherUnits(None, stream, 0, 2, 1, C.byref(maps), 1, 32) == -1
        L.GPU_DestroyGraph(g)
    finally:
        pbrhip.rccl_comm_destroy(comm)
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(tex)


def test_two_graphs_in_flight_keep_submission_order(gpu):
    """The reference keeps two graphs in flight on one queue (main.cpp:49-51, 91-99): graph B, submitted while graph A still
    runs, must see what A wrote.  A = a 256^2 prefilter chain (milliseconds), B = read-back of its mips; no wait in between."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    env = synth.synth_env(128, seed=0x5EED00AB)
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 128, 128, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 8, 64, 256)
    spec = maps.tex_specular_env_map
    L.PBR_GenPrefilteredEnvMap(tex, spec, 16)                       # reference result, sequential
    want = [pbrhip.read_mip(spec, m).copy() for m in range(5)]
    gA, gB = L.GPU_MakeGraph(), L.GPU_MakeGraph()
    for m in range(5):
        L.GPU_OpClearColorF(gA, spec, m, 0.0, 0.0, 0.0, 0.0)
    L.GPU_GraphSubmit(gA); L.GPU_GraphWait(gA)
    pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena()
    units, n = pbrhip.partition(256, 16, 0, 128, 1, 0)
    bufs = [L.GPU_MakeBuffer(L.GPUX_TextureMipBytes(spec, m), pbrhip.BufferFlag_CPU, None) for m in range(5)]
    L.PBR_RecordUnits(pipes, gA, arena, tex, C.byref(maps), units, n)
    for m in range(5):
        L.GPUX_OpCopyTextureMipToBuffer(gB, spec, m, bufs[m], 0)
    L.GPU_GraphSubmit(gA)
    L.GPU_GraphSubmit(gB)                                           # A is still running
    L.GPU_GraphWait(gB); L.GPU_GraphWait(gA)
    for m in range(5):
        nb = L.GPUX_TextureMipBytes(spec, m)
        got = np.frombuffer((C.c_char * nb).from_address(bufs[m].contents.data), np.float32).reshape(want[m].shape)
        assert np.array_equal(got.view(np.uint32), want[m].view(np.uint32)), m
    for b in bufs:
        L.GPU_DestroyBuffer(b)
    L.GPU_DestroyGraph(gA); L.GPU_DestroyGraph(gB); L.GPU_DestroyDescriptorArena(arena); L.PBR_DestroyIBLPipelines(pipes)
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(tex)


def test_invalidate_texture_after_external_write(gpu):
    """A texture over caller-owned memory (GPUX_MakeTextureExternal: where RCCL receives into) is sampled through lazily built
    twins; after the memory changes behind the backend's back GPUX_InvalidateTexture makes the next use rebuild them."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    envA = synth.synth_env(32, seed=0x5EED00AA)
    envB = np.ascontiguousarray(envA[[1, 0, 3, 2, 5, 4]] * np.float32(0.5)); envB[..., 3] = 1.0
    nbytes = envA.nbytes
    store = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_GPU, None)
    ext = L.GPUX_MakeTextureExternal(pbrhip.Format_RGBA32F, 32, 32, 1, pbrhip.TextureFlag_Cubemap, L.GPUX_BufferDevicePtr(store), nbytes)
    out = pbrhip.make_texture(pbrhip.Format_RGBA32F, 8, 8, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage)
    ref = {}
    g = L.GPU_MakeGraph()
    for name, env in (("A", envA), ("B", envB)):                    # what the irradiance of each environment is
        t = pbrhip.make_texture(pbrhip.Format_RGBA32F, 32, 32, pbrhip.TextureFlag_Cubemap, env)
        L.PBR_GenIrradianceMap(t, out); ref[name] = pbrhip.read_mip(out, 0).copy(); L.GPU_DestroyTexture(t)
    assert not np.array_equal(ref["A"], ref["B"])

    def write(env):                                                 # writes the buffer, not the texture: invisible to the backend
        up = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_CPU, np.ascontiguousarray(env).ctypes.data_as(C.c_void_p))
        L.GPU_OpCopyBufferToBuffer(g, up, store, 0, 0, nbytes); L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_DestroyBuffer(up)

    write(envA); L.GPUX_InvalidateTexture(ext)
    L.PBR_GenIrradianceMap(ext, out)
    assert np.array_equal(pbrhip.read_mip(out, 0), ref["A"])
    write(envB)
    L.PBR_GenIrradianceMap(ext, out)                                # stale twin: still A's values (this is the hazard)
    assert np.array_equal(pbrhip.read_mip(out, 0), ref["A"])
    L.GPUX_InvalidateTexture(ext)
    L.PBR_GenIrradianceMap(ext, out)
    assert np.array_equal(pbrhip.read_mip(out, 0), ref["B"])
    L.GPU_DestroyGraph(g); L.GPU_DestroyTexture(out); L.GPU_DestroyTexture(ext); L.GPU_DestroyBuffer(store)


def test_overlapped_partitioned_run_equals_sequential(gpu):
    """PBR_RunPartitionedIBL (two graphs: the units of the early levels, then the rest; each followed by its exchange) on world = 1:
    both graphs run, nothing moves, and the maps equal the sequential PBR_Gen* result bit for bit -- for the default early level
    (mip 1), for an early set that holds most of the job, and for an empty one."""
    import pbrhip
    from pbrhip import synth
    L = gpu
    env = synth.synth_env(64, seed=0x5EED00AC)
    tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, 64, 64, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 16, 64, 128)
    spec, irr = maps.tex_specular_env_map, maps.irradiance_map
    n_mips = spec.contents.mip_level_count
    L.PBR_GenPrefilteredEnvMap(tex, spec, 1); L.PBR_GenIrradianceMap(tex, irr)
    want = [pbrhip.read_mip(spec, m).copy() for m in range(n_mips)] + [pbrhip.read_mip(irr, 0).copy()]
    gA, gB, gC = L.GPU_MakeGraph(), L.GPU_MakeGraph(), L.GPU_MakeGraph()
    pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena()
    for early in (0x2, 0x80000006, 0x0):
        for m in range(n_mips):
            L.GPU_OpClearColorF(gC, spec, m, 0.0, 0.0, 0.0, 0.0)
        L.GPU_OpClearColorF(gC, irr, 0, 0.0, 0.0, 0.0, 0.0)
        L.GPU_GraphSubmit(gC); L.GPU_GraphWait(gC)
        L.GPU_OpGenerateMipmaps(gA, tex)                              # the caller's prologue rides in the early graph
        moved = L.PBR_RunPartitionedIBL(pipes, gA, gB, arena, tex, C.byref(maps), None, 0, 1, 0, 1, early)
        assert moved == 0
        L.GPU_GraphWait(gB); L.GPU_GraphWait(gA); L.GPU_ResetDescriptorArena(arena)
        got = [pbrhip.read_mip(spec, m) for m in range(n_mips)] + [pbrhip.read_mip(irr, 0)]
        for m, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (hex(early), m)
    assert L.PBR_RunPartitionedIBL(pipes, gA, gA, arena, tex, C.byref(maps), None, 0, 1, 0, 1, 2) == -1      # one graph twice
    assert L.PBR_RunPartitionedIBL(pipes, gA, gB, arena, tex, C.byref(maps), None, 0, 2, 1, 1, 2) == -1      # world 2 without a communicator
    for g in (gA, gB, gC):
        L.GPU_DestroyGraph(g)
    L.GPU_DestroyDescriptorArena(arena); L.PBR_DestroyIBLPipelines(pipes)
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(tex)


def test_gather_phases_cover_the_exchange_once_and_run_on_two_streams(gpu):
    """The two phases of the overlapped exchange (PBR_GatherPlan with the early mask and with its complement) are disjoint and
    together equal the one-shot plan, on the sending ranks and on the root (whose receives mirror the peers' sends); then the
    phases of every rank of a 3-way split run as two grouped RCCL exchanges on two different streams of a 1-rank communicator
    (to self, into a second buffer) and every unit arrives bit-identical."""
    import pbrhip
    L = gpu
    tex, maps = _maps_32(L, pbrhip)
    spec = maps.tex_specular_env_map
    ALL, EARLY = 0xFFFFFFFF, 0x2

    def plan(world, rank, mask):
        cap = 4096
        xs = (pbrhip.PBR_XferRange * cap)()
        n = L.PBR_GatherPlan(0, world, rank, C.byref(maps), 1, 32, mask, xs, cap)
        assert n >= 0 and n == L.PBR_GatherPlan(0, world, rank, C.byref(maps), 1, 32, mask, None, 0)
        return [(xs[i].ptr, xs[i].bytes, xs[i].peer) for i in range(n)]

    for world in (2, 3, 8):
        sends = {}
        for r in range(1, world):
            a, e, l = plan(world, r, ALL), plan(world, r, EARLY), plan(world, r, ~EARLY & ALL)
            assert sorted(e + l) == sorted(a) and not set(e) & set(l) and all(p == 0 for (_, _, p) in a)
            ivs = sorted((p_, p_ + b) for (p_, b, _) in a)
            assert all(x[1] <= y[0] for x, y in zip(ivs, ivs[1:]))            # ranges of one rank do not overlap
            sends[r] = a
        for mask in (ALL, EARLY, ~EARLY & ALL):
            root = plan(world, 0, mask)
            peers = sorted((p_, b, r) for r in range(1, world) for (p_, b, _) in plan(world, r, mask))
            assert sorted(root) == peers                                       # root receives exactly what the peers send, in place
        units0, n0 = pbrhip.partition(32, 1, 8, 32, world, 0)
        assert n0 > 0 and plan(1, 0, ALL) == []
    # run the phases of ranks 1 and 2 of the 3-way split: early on stream A, late on stream B, one communicator
    comm = pbrhip.rccl_comm_init(1, 0, pbrhip.rccl_unique_id())
    try:
        gA, gB = L.GPU_MakeGraph(), L.GPU_MakeGraph()
        nbytes = L.GPUX_TextureTotalBytes(spec)
        dst = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_GPU, None); host = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_CPU, None)
        zbuf = L.GPU_MakeBuffer(nbytes, pbrhip.BufferFlag_CPU, np.zeros(nbytes, np.uint8).ctypes.data_as(C.c_void_p))
        L.GPU_OpCopyBufferToBuffer(gA, zbuf, dst, 0, 0, nbytes); L.GPU_GraphSubmit(gA); L.GPU_GraphWait(gA)
        src0, dst0 = L.GPUX_TextureDevicePtr(spec, 0), L.GPUX_BufferDevicePtr(dst)
        covered = np.zeros(nbytes, bool)
        for g, mask in ((gA, EARLY), (gB, ~EARLY & ALL)):
            S, R = [], []
            for r in (1, 2):
                for (p_, b, _) in plan(3, r, mask):
                    off = p_ - src0
                    if not (0 <= off < nbytes):
                        continue                                             # irradiance units live in another texture
                    S.append(pbrhip.PBR_XferRange(p_, b, 0)); R.append(pbrhip.PBR_XferRange(dst0 + off, b, 0)); covered[off:off + b] = True
            assert S
            Sa, Ra = (pbrhip.PBR_XferRange * len(S))(*S), (pbrhip.PBR_XferRange * len(R))(*R)
            assert L.PBR_ExchangeRanges(comm, L.GPUX_GraphStream(g), Sa, len(S), Ra, len(R)) == 0
        L.GPU_GraphSubmit(gA); L.GPU_GraphSubmit(gB)                           # empty graphs: B is ordered after A's stream, exchanges included
        L.GPU_GraphWait(gB); L.GPU_GraphWait(gA)
        L.GPU_OpCopyBufferToBuffer(gA, dst, host, 0, 0, nbytes); L.GPU_GraphSubmit(gA); L.GPU_GraphWait(gA)
        got = np.frombuffer((C.c_char * nbytes).from_address(host.contents.data), np.uint8)
        want = np.concatenate([pbrhip.read_mip(spec, m).view(np.uint8).ravel() for m in range(spec.contents.mip_level_count)])
        assert covered.any() and np.array_equal(got[covered], want[covered]) and not got[~covered].any()
        for b_ in (dst, host, zbuf):
            L.GPU_DestroyBuffer(b_)
        L.GPU_DestroyGraph(gA); L.GPU_DestroyGraph(gB)
    finally:
        pbrhip.rccl_comm_destroy(comm)
    L.PBR_DestroyIBLMaps(C.byref(maps)); L.GPU_DestroyTexture(tex)


_TORCH_FIRST = r"""
import os, sys, ctypes as C
import torch                                        # FIRST: maps torch's bundled librccl.so (SONAME librccl.so.1), as in bench.py
sys.path.insert(0, os.path.join(sys.argv[1], "vulkan-pbr-renderer_amd", "python"))
import pbrhip
torch.cuda.set_device(0)
L = pbrhip.init(device=0)
ver, path = pbrhip.rccl_info()                      # what host/pbr_gather.c bound at first use
print("RCCL", ver, path)
comm = pbrhip.rccl_comm_init(1, 0, pbrhip.rccl_unique_id())
assert pbrhip.comm_info(comm) == (1, 0)
src = torch.arange(1 << 20, dtype=torch.float32, device="cuda") * 0.5 + 1.0
dst = torch.zeros_like(src)
g = L.GPU_MakeGraph()
n = 7                                               # ragged pieces, each to self
cuts = [0] + sorted(int(x) for x in torch.randint(1, src.numel() - 1, (n - 1,), generator=torch.Generator().manual_seed(5))) + [src.numel()]
S = (pbrhip.PBR_XferRange * n)(*[pbrhip.PBR_XferRange(src.data_ptr() + 4 * a, 4 * (b - a), 0) for a, b in zip(cuts, cuts[1:])])
R = (pbrhip.PBR_XferRange * n)(*[pbrhip.PBR_XferRange(dst.data_ptr() + 4 * a, 4 * (b - a), 0) for a, b in zip(cuts, cuts[1:])])
torch.cuda.synchronize()
assert L.PBR_ExchangeRanges(comm, L.GPUX_GraphStream(g), S, n, R, n) == 0
L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)            # waits for the graph's stream, hence for the exchange
torch.cuda.synchronize()
assert torch.equal(src, dst)
pbrhip.rccl_comm_destroy(comm)
L.GPU_DestroyGraph(g); L.GPU_WaitUntilIdle(); L.GPU_Deinit()
print("TORCH_FIRST_OK")
"""


def test_exchange_with_torch_imported_first(gpu, tmp_path):
    """the library load order of bench.py -- torch first, so that the process's librccl.so.1 is torch's bundled
    copy -- then a 1-rank communicator from that same copy and PBR_ExchangeRanges through the C host layer's run-time binding.
    Runs in a child process (the test session itself has ROCm's librccl mapped by earlier tests)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "torch_first.py"
    script.write_text(_TORCH_FIRST)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "TORCH_FIRST_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL ")][0]
    assert "torch" in line.split(" ", 2)[2], f"expected the exchange to bind torch's bundled librccl, got: {line}"


def test_bench_launches_its_own_ranks(gpu):
    """`python3 bench.py --gpus 2` with no launcher starts its two workers itself (child torch.distributed.run)
    and returns their code; here as the gloo rehearsal on one GPU (RCCL refuses two ranks on one device), small workload, with
    --check: the gathered result equals what one GPU computes alone, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PBR_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "ref", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-shade", "--check"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["extra"]["check_gather_equals_single_gpu"] is True
    assert out["extra"]["check_max_rel_err_vs_oracle"] < 1e-4
    assert out["rccl"]["backend"] == "gloo" and out["step_split"]["bytes_sent_by_rank"][1] > 0
