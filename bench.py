#!/usr/bin/env python3
"""bench.py -- headline benchmark of the IBL-precompute hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|ref]
  N > 1: either the launcher form (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
         or plain `python bench.py --gpus N`: without WORLD_SIZE in the environment the script starts its N workers itself
         (a child `torch.distributed.run`, decided before anything touches the GPU) and exits with their code.

One "step" = one full IBL precompute of the workload with the environment cube's level 0 already
resident in HBM: mip-chain build (K2) + apron build + specular prefilter of every mip (K4a copy,
K4b Monte-Carlo) + diffuse irradiance (K3), driven through the GPU_* C ABI exactly as the reference's
HotreloadShaders does (render.cpp:505-589).  With N ranks the (mip, face, row-tile) work units are
cost-partitioned over the ranks (strong scaling: the job is fixed) and the output tiles are gathered
to rank 0 with one grouped RCCL send/recv batch per step.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the stream it runs on) and `cpu_baseline` (the scalar C oracle timed on the host).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

WORKLOADS = {
    # name: (env W, specular size, irradiance size, seed, description)
    "c4": (2048, 4096, 128, 0x5EED0004, "C4: 4096x4096 specular prefilter (6 faces, 13 mips) + 128x128 irradiance from a 2048^2 HDR cube"),
    "c2": (1024, 512, 32, 0x5EED0001, "C2: 512x512 specular prefilter (6 faces, 10 mips) + 32x32 irradiance from a 1024^2 HDR cube"),
    "ref": (256, 256, 32, 0x5EED00AB, "reference sizes: 256x256 specular (mips 0-4) + 32x32 irradiance from a 256^2 HDR cube"),
}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector
FLOP_PER_SAMPLE = 65.0         # frame transform + face select + projection + 4-tap RGB lerp + accumulate


def load_env(W, seed, workers):
    from pbrhip import synth
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pbr_env_{W}_{seed:08x}.npy")
    if os.path.exists(cache):
        try:
            a = np.load(cache)
            if a.shape == (6, W, W, 4):
                return a
        except Exception:
            pass
    a = synth.synth_env(W, seed=seed, workers=workers)
    try:
        tmp = cache + f".{os.getpid()}.tmp.npy"
        np.save(tmp, a)
        os.replace(tmp, cache)
    except Exception:
        pass
    return a


def sha256_of(paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
    return h.hexdigest()


def inspect_hdr(path):
    """Resolution line of a Radiance file -> layout (the decode itself is the C host layer's: PBR_DecodeHDR)."""
    with open(path, "rb") as f:
        head = f.read(4096)
    if not head.startswith(b"#?"):
        raise SystemExit(f"{path}: not a Radiance .hdr file")
    import re as _re
    m = _re.search(rb"\n\n-Y (\d+) \+X (\d+)\n", head)
    if not m:
        raise SystemExit(f"{path}: only the standard -Y H +X W orientation is supported (as stb_image.h:7196)")
    h, w = int(m.group(1)), int(m.group(2))
    if h == 6 * w:
        layout = "cube strip"
    elif w == 2 * h:
        layout = "equirectangular"
    else:
        raise SystemExit(f"{path}: {w}x{h} is neither a 6-face strip (height == 6 x width) nor a 2:1 panorama")
    return {"path": os.path.abspath(path), "width": w, "height": h, "layout": layout, "sha256": sha256_of([path])}


def load_gbuffer_dir(d):
    names = ("base_color", "normal", "orm", "emissive", "depth")
    paths = [os.path.join(d, n + ".npy") for n in names]
    arrs = [np.load(p, allow_pickle=False) for p in paths]
    H, W = arrs[4].shape
    for a in arrs[:4]:
        if a.dtype != np.uint8 or a.shape != (H, W, 4):
            raise SystemExit(f"{d}: colour planes must be uint8 [H][W][4] matching depth.npy ({W}x{H})")
    if arrs[4].dtype != np.float32:
        raise SystemExit(f"{d}: depth.npy must be float32 [H][W]")
    cam = {"pos": [0.0, -9.0, 0.0], "ori_xyzw": None, "fov": 75.0}
    cj = os.path.join(d, "camera.json")
    if os.path.exists(cj):
        cam.update(json.load(open(cj)))
        paths.append(cj)
    return {"base": arrs[0], "normal": arrs[1], "orm": arrs[2], "emissive": arrs[3], "depth": arrs[4], "camera": cam,
            "info": {"dir": os.path.abspath(d), "width": int(W), "height": int(H), "sha256": sha256_of(paths)}}


def nonzero_weight_count(L, nsamples, roughness):
    tab = np.zeros((nsamples, 4), np.float32)
    alpha = C.c_float()
    return L.pbrk_host_prefilter_table(nsamples, C.c_float(roughness), tab.ctypes.data_as(C.c_void_p), C.byref(alpha))


def ref_roughness(mip):
    if mip < 5:
        return [0.0, 0.03, 0.15, 0.4, 0.6][mip]
    return min(1.0, float(np.float32(0.6) + np.float32(0.08) * np.float32(mip - 4)))


def cpu_baseline(env, W, spec_size, irr_size, budget_s=15.0):
    """Scalar C oracle (oracle/pbr_oracle.c, 'port') timed on the host cores over a bounded sample of the same job."""
    import pbr_oracle as O
    threads = O.get_threads()
    pyr = O.build_pyramid(env)
    nm = O.mip_count(spec_size)
    t_mc = 0.0; n_mc = 0
    t_cp = 0.0; n_cp = 0
    # copy mip: a few rows of every face
    size0 = spec_size
    rows = max(1, min(size0, 16))
    t = time.perf_counter()
    O.prefilter_mip(pyr, W, spec_size, 0, rows=(0, rows))
    t_cp += time.perf_counter() - t; n_cp += 6 * rows * size0
    # Monte-Carlo mips: row slices of increasing size until the budget is spent
    deadline = time.perf_counter() + budget_s
    sample_desc = []
    for mip in (2, 1, 3, 4):
        if mip >= nm:
            continue
        size = max(1, spec_size >> mip)
        r = max(1, min(size, (threads * 2 + 5) // 6))
        while time.perf_counter() < deadline:
            t = time.perf_counter()
            O.prefilter_mip(pyr, W, spec_size, mip, rows=(0, r))
            dt = time.perf_counter() - t
            t_mc += dt; n_mc += 6 * r * size
            sample_desc.append(f"mip{mip}:{6 * r * size}tx")
            if dt > budget_s / 8 or r >= size:
                break
            r = min(size, r * 2)
    rate_mc = n_mc / t_mc if t_mc > 0 else float("nan")       # MC texels / s (8192 samples each, no zero-weight skipping)
    rate_cp = n_cp / t_cp
    total_mc = sum(6 * max(1, spec_size >> m) ** 2 for m in range(1, nm)) + 6 * irr_size * irr_size / 8.0
    total_cp = 6 * spec_size * spec_size
    est_time = total_mc / rate_mc + total_cp / rate_cp
    total_units = total_cp + sum(6 * max(1, spec_size >> m) ** 2 for m in range(1, nm)) + 6 * irr_size * irr_size
    # the same loops on ONE thread, in the hoisted-table form and in the literal form (the shader's own operation
    # sequence per sample: two Rotate() calls, acos, tan, exp), on a few rows of mip 2
    extra_rates = {}
    try:
        O.set_threads(1)
        size2 = max(1, spec_size >> 2)
        for key, literal, rows1 in (("single_thread", False, 2), ("single_thread_literal", True, 1)):
            t = time.perf_counter()
            O.prefilter_mip(pyr, W, spec_size, 2, faces=(0, 1), rows=(0, rows1), literal=literal)
            dt = time.perf_counter() - t
            tx = rows1 * size2
            extra_rates[key] = {"msamples_per_s": tx * 8192 / dt / 1e6, "mc_texels_per_s": tx / dt, "cores": 1,
                                "sample": f"mip2: {tx} texels x 8192 samples in {dt:.2f} s"}
    finally:
        O.set_threads(threads)
    return {
        "value": total_units / est_time / 1e6, "unit": "Mtexels/s", "cores": threads, "kind": "port", **extra_rates,
        "sample": f"oracle/pbr_oracle.c (OpenMP, {threads} threads): {n_cp} copy texels + {n_mc} Monte-Carlo texels x 8192 samples "
                  f"({', '.join(sample_desc[:6])}) in {t_mc + t_cp:.1f} s, extrapolated to the whole job "
                  f"({rate_mc * 8192 / 1e6:.1f} Msamples/s, est. {est_time:.0f} s per job)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shade", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the 7680x4320 screen-band shade leg (extra.shade_c5)")
    ap.add_argument("--c5-frames", type=int, default=10)
    ap.add_argument("--overlap", action="store_true", default=os.environ.get("PBR_BENCH_OVERLAP", "0") == "1",
                    help="record the job as two graphs (mip-1 units first) and send each part while the next computes (PBR_RunPartitionedIBL; "
                         "needs the C gather for N > 1; opt-in: never run on more than one GPU)")
    ap.add_argument("--check", action="store_true", help="after the run, spot-check output texels against the oracle")
    ap.add_argument("--bounded-cut", action="store_true",
                    help="also time the job with the opt-in tolerance-budgeted sample cut (GPUX_SetPrefilterTolerance(1e-7)) -> extra.c4_bounded_cut; "
                         "off by default so that a kernel trace of the default command holds the exact kernels only")
    ap.add_argument("--hdr", default=None, metavar="FILE.hdr",
                    help="real environment instead of the synthetic one: a Radiance .hdr cube strip (height == 6 x width, the reference's "
                         "layout, asset_import.cpp:17-27 -> PBR_MakeTextureFromHDRIFile) or an equirectangular panorama (width == 2 x "
                         "height -> PBR_MakeTextureFromEquirectHDRIFile at the workload's cube size); `data` becomes \"file\" with its SHA-256")
    ap.add_argument("--gbuffer", default=None, metavar="DIR",
                    help="real G-buffer for the 1920x1080-class shade leg (extra.shade): DIR holds base_color.npy, normal.npy, orm.npy, "
                         "emissive.npy (uint8 [H][W][4], the RGBA8 attachments of render.cpp:680-687), depth.npy (float32 [H][W]) and "
                         "optionally camera.json {\"pos\": [x,y,z], \"ori_xyzw\": [..], \"fov\": deg}")
    ap.add_argument("--dry-launch", action="store_true", help="print the worker command `--gpus N` would start and exit 0 (no GPU, no torch)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: become the launcher.  Decided here, before torch / HIP are touched: the workers are CHILD
    # processes (never an exec from a process that has initialised the GPU), rank 0 of them prints the one JSON line.
    if args.dry_launch or (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        worker_args = [a for a in sys.argv[1:] if a != "--dry-launch"]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + worker_args
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
        if args.dry_launch:
            print(json.dumps({"dry_launch": True, "n_workers": args.gpus, "cmd": cmd,
                              "env": {k: env[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")}}))
            sys.exit(0)
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree", file=sys.stderr)
        sys.exit(2)

    W, spec_size, irr_size, seed, desc = WORKLOADS[args.workload]
    # inputs first (fork-based workers), GPU afterwards
    hdr_info = None
    if args.hdr:
        hdr_info = inspect_hdr(args.hdr)                                          # dims + layout + SHA-256; the file itself is decoded by the C host layer
        env = None
    else:
        env = load_env(W, seed, workers=max(1, min(6, (os.cpu_count() or 8) // max(1, world))))
    gb_file = load_gbuffer_dir(args.gbuffer) if args.gbuffer else None
    c5_gbd = None
    if not args.no_shade and not args.no_c5:                      # this rank's band of the 7680x4320 G-buffer (worker processes: before the GPU is touched)
        from pbrhip import synth
        c5_gbd = synth.synth_gbuffer_temple(7680, 4320, rows=(4320 * rank // world, 4320 * (rank + 1) // world),
                                            workers=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))

    import torch
    import torch.distributed as dist
    import pbrhip
    # PBR_BENCH_BACKEND=gloo is a functional rehearsal of the N>1 path on a box with fewer GPUs than ranks (tiles are
    # staged through host memory; timings are meaningless); the real path is RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("PBR_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    L = pbrhip.init(device=dev_index)
    L.GPUX_EnableOpTiming(1)
    # The exchange step runs in the C host layer (host/pbr_gather.c: PBR_GatherUnits / PBR_GatherBands on grouped ncclSend /
    # ncclRecv); this script only bootstraps the communicator (the library never owns one).  PBR_BENCH_GATHER=torch selects
    # the same exchange through torch.distributed P2P ops instead; the gloo rehearsal always does.
    comm, gather_impl = None, "none"
    if world > 1:
        gather_impl = "torch.distributed"
        if backend == "nccl" and os.environ.get("PBR_BENCH_GATHER", "c") != "torch":
            ok = 1
            try:
                uid = [pbrhip.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = pbrhip.rccl_comm_init(world, rank, uid[0])
            except Exception as e:                                  # pragma: no cover (reported in the JSON line, never silent)
                ok = 0
                print(f"bench.py: rank {rank}: RCCL communicator for the C gather failed ({e!r}); using torch.distributed P2P", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                gather_impl = "PBR_GatherUnits (C host layer, RCCL)"
            else:
                comm = None
                gather_impl = "torch.distributed (C gather unavailable)"
    # which RCCL the C host layer bound (it binds at first use: here torch was imported first, so it is torch's bundled copy --
    # the same copy that made `comm`), and what the communicator itself says about its size
    rccl_report = None
    if world > 1:
        rccl_report = {"backend": backend, "gather": gather_impl, "version": None, "library": None, "comm_ranks": None, "comm_rank_of_rank0": None}
        try:
            ver, path = pbrhip.rccl_info()
            rccl_report.update(version=ver, library=path)
            if comm is not None:
                n_c, r_c = pbrhip.comm_info(comm)
                rccl_report.update(comm_ranks=n_c, comm_rank_of_rank0=r_c, comm_matches_launcher=bool(n_c == world and r_c == rank))
        except Exception as e:                                  # reported, never fatal: the exchange itself fails loudly if the binding is wrong
            rccl_report["error"] = repr(e)

    # ---- resources: env cube (level 0 resident), output maps over torch-owned HBM (so RCCL can move them)
    if hdr_info is not None:
        if hdr_info["layout"] == "cube strip":
            env_tex = L.PBR_MakeTextureFromHDRIFile(args.hdr.encode())            # asset_import.cpp:17-27: decode, upload, mip chain
        else:
            env_tex = L.PBR_MakeTextureFromEquirectHDRIFile(args.hdr.encode(), W)  # N1: panorama -> W^2 cube (K6) + mip chain
        if not env_tex:
            raise RuntimeError(f"{args.hdr}: the host layer could not load it")
        W = int(env_tex.contents.width)
        env = pbrhip.read_mip(env_tex, 0)                                          # what the oracle legs (cpu_baseline, --check) are fed
        desc += f" [environment from file: {hdr_info['layout']}, cube {W}^2]"
    else:
        env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
    spec_flags = pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps | pbrhip.TextureFlag_StorageImage
    n_mips = L.pbrk_mip_count(spec_size, spec_size)
    spec_floats = L.pbrk_pyramid_texels(spec_size, n_mips) * 4
    spec_mem = torch.zeros(spec_floats, dtype=torch.float32, device="cuda")
    irr_mem = torch.zeros(6 * irr_size * irr_size * 4, dtype=torch.float32, device="cuda")
    maps = pbrhip.PBR_IBLMaps()
    maps.tex_specular_env_map = L.GPUX_MakeTextureExternal(pbrhip.Format_RGBA32F, spec_size, spec_size, 1, spec_flags,
                                                           spec_mem.data_ptr(), spec_mem.numel() * 4)
    maps.irradiance_map = L.GPUX_MakeTextureExternal(pbrhip.Format_RGBA32F, irr_size, irr_size, 1,
                                                     pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_StorageImage,
                                                     irr_mem.data_ptr(), irr_mem.numel() * 4)
    maps.brdf_lut = pbrhip.make_texture(pbrhip.Format_RG16F, 256, 256, pbrhip.TextureFlag_StorageImage)
    L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)          # 256 KB: computed redundantly on every rank, not timed

    my_units, n_my = pbrhip.partition(spec_size, 1, irr_size, W, world, rank)
    all_units = []
    if world > 1:
        for r in range(world):
            u, n = pbrhip.partition(spec_size, 1, irr_size, W, world, r)
            all_units.append([(u[i].kind, u[i].mip, u[i].face0, u[i].face1, u[i].row0, u[i].row1) for i in range(n)])

    def unit_slice(kind, mip, f0, f1, r0, r1):
        """Flat float range of one work unit inside spec_mem / irr_mem (faces are consecutive: one unit = one face)."""
        if kind == pbrhip.Unit_Irradiance:
            size, base, mem = irr_size, 0, irr_mem
        else:
            size, base, mem = max(1, spec_size >> mip), L.pbrk_level_offset(spec_size, mip) * 4, spec_mem
        assert f1 == f0 + 1 or (r0 == 0 and r1 == size)
        a = base + ((f0 * size + r0) * size) * 4
        b = base + (((f1 - 1) * size + r1) * size) * 4
        return mem[a:b]

    pipes = L.PBR_MakeIBLPipelines()
    arena = L.GPU_MakeDescriptorArena()
    graph = L.GPU_MakeGraph()
    graph2 = L.GPU_MakeGraph() if args.overlap else None
    overlap = bool(args.overlap and (world == 1 or comm is not None))

    # the exchange is the same every step: the send/recv descriptors (views of the output memory) are built once
    gather_ops, staged = [], []
    if world > 1:
        if comm is not None:
            pass                                                    # PBR_GatherUnits derives the ranges itself
        elif rank == 0:
            for r in range(1, world):
                for u in all_units[r]:
                    dst = unit_slice(*u)
                    if backend == "nccl":
                        gather_ops.append(dist.P2POp(dist.irecv, dst, r))
                    else:
                        staged.append((r, dst, torch.empty(dst.shape, dtype=dst.dtype)))
        elif backend == "nccl":
            gather_ops = [dist.P2POp(dist.isend, unit_slice(*u), 0) for u in all_units[rank]]

    phase = {"compute": 0.0, "exchange": 0.0}           # host-clock split of a step on this rank (reported for N > 1)

    def step():
        t_a = time.perf_counter()
        L.GPU_OpGenerateMipmaps(graph, env_tex)                                   # K2 (+ apron rebuild on first sample)
        if overlap:                                                               # two graphs, each followed by its share of the exchange
            moved = L.PBR_RunPartitionedIBL(pipes, graph, graph2, arena, env_tex, C.byref(maps), comm, 0, world, rank, 1, 0x2)
            if moved < 0:
                raise RuntimeError(f"PBR_RunPartitionedIBL failed ({moved})")
            L.GPU_GraphWait(graph2); L.GPU_GraphWait(graph)
            L.GPU_ResetDescriptorArena(arena)
            t_b = time.perf_counter()
            # compute = the two graphs' busy spans on their own streams (one event pair each: dispatches overlapping on side
            # streams are not counted twice); exchange = what the step took beyond that
            k_s = (L.GPUX_GraphSpanMs(graph) + L.GPUX_GraphSpanMs(graph2)) * 1e-3
            phase["compute"] += min(k_s, t_b - t_a); phase["exchange"] += max(0.0, t_b - t_a - k_s)
            return
        L.PBR_RecordUnits(pipes, graph, arena, env_tex, C.byref(maps), my_units, n_my)
        L.GPU_GraphSubmit(graph)
        if comm is not None:                                                      # the exchange follows the kernels on the graph's own stream
            moved = L.PBR_GatherUnits(comm, L.GPUX_GraphStream(graph), 0, world, rank, C.byref(maps), 1, W)
            if moved < 0:
                raise RuntimeError(f"PBR_GatherUnits failed ({moved})")
        L.GPU_GraphWait(graph)                                                    # ... and this waits for both
        L.GPU_ResetDescriptorArena(arena)
        t_b = time.perf_counter()
        if comm is not None:                                                      # split by the graph's busy span (first op .. last join, one event pair)
            k_s = L.GPUX_GraphSpanMs(graph) * 1e-3
            phase["compute"] += min(k_s, t_b - t_a); phase["exchange"] += max(0.0, t_b - t_a - k_s)
            return
        phase["compute"] += t_b - t_a
        if world > 1:                                                             # one grouped RCCL exchange: tiles -> rank 0
            if backend == "nccl":
                for w in (dist.batch_isend_irecv(gather_ops) if gather_ops else []):
                    w.wait()
                torch.cuda.current_stream().synchronize()      # wait() only orders streams: the step ends when its exchange has landed
            else:
                ops = [dist.P2POp(dist.irecv, buf, r) for (r, dst, buf) in staged] if rank == 0 else \
                      [dist.P2POp(dist.isend, unit_slice(*u).cpu(), 0) for u in all_units[rank]]
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()
                for (r, dst, buf) in staged:
                    dst.copy_(buf)
            phase["exchange"] += time.perf_counter() - t_b

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        L.GPU_WaitUntilIdle()

    for _ in range(args.warmup):
        step()
    op_ms = {}
    phase["compute"] = phase["exchange"] = 0.0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for g_ in ((graph, graph2) if overlap else (graph,)):
            for i in range(L.GPUX_GraphTimedOpCount(g_)):
                nm = L.GPUX_GraphTimedOpName(g_, i).decode()
                op_ms.setdefault(nm, []).append(L.GPUX_GraphTimedOpMs(g_, i))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    step_split = None
    if world > 1:       # per-rank compute time and (on rank 0: waiting for the slowest sender + the transfer itself) exchange time
        pt = torch.tensor([phase["compute"], phase["exchange"]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(pt) for _ in range(world)]
        dist.all_gather(gathered, pt)
        step_split = {"compute_ms_per_step_by_rank": [float(g[0]) / args.steps * 1e3 for g in gathered],
                      "exchange_ms_per_step_by_rank": [float(g[1]) / args.steps * 1e3 for g in gathered],
                      "bytes_sent_by_rank": [0] + [int(sum(unit_slice(*u).numel() * 4 for u in all_units[r])) for r in range(1, world)],
                      "gather": gather_impl + (" in two overlapped phases (PBR_RunPartitionedIBL, mip 1 first)" if overlap else "")}

    total_texels = sum(6 * max(1, spec_size >> m) ** 2 for m in range(n_mips)) + 6 * irr_size * irr_size
    ms_per_step = elapsed / args.steps * 1e3
    value = total_texels * args.steps / elapsed / 1e6

    # ---- per-kernel accounting (this rank's launches; at N=1 one launch per output level)
    kernels = []
    for nm, v in op_ms.items():
        ms = float(np.mean(v))
        launches_per_step = len(v) / args.steps
        ent = {"kernel": nm, "avg_ms": ms, "launches_per_step": launches_per_step}
        if world == 1 and nm.startswith("K4b.prefilter_mc.mip"):
            mip = int(nm.rsplit("mip", 1)[1])
            size = max(1, spec_size >> mip)
            nz = nonzero_weight_count(L, 8192, ref_roughness(mip))
            samples = 6.0 * size * size * nz
            ent.update(bound="valu", samples=samples, flop=samples * FLOP_PER_SAMPLE,
                       achieved_tflops=samples * FLOP_PER_SAMPLE / (ms * 1e-3) / 1e12,
                       alg_bytes=6.0 * size * size * 16, msamples_per_s=samples / (ms * 1e-3) / 1e6)
            ent["frac"] = ent["achieved_tflops"] / PEAK_FP32_TFLOPS
        elif world == 1 and nm == "K3.irradiance":
            samples = 6.0 * irr_size * irr_size * 1024
            ent.update(bound="valu", samples=samples, flop=samples * FLOP_PER_SAMPLE,
                       achieved_tflops=samples * FLOP_PER_SAMPLE / (ms * 1e-3) / 1e12)
            ent["frac"] = ent["achieved_tflops"] / PEAK_FP32_TFLOPS
        elif world == 1 and nm.startswith("K4a."):
            src = 6.0 * (W // 2) ** 2 * 16
            byt = 6.0 * spec_size * spec_size * 16 + src
            ent.update(bound="hbm", alg_bytes=byt, achieved_gbs=byt / (ms * 1e-3) / 1e9)
            ent["frac"] = ent["achieved_gbs"] / PEAK_HBM_GBS
        elif world == 1 and nm == "K2.mip_chain":
            byt = 80.0 * 6 * W * W / 3.0
            ent.update(bound="hbm", alg_bytes=byt, achieved_gbs=byt / (ms * 1e-3) / 1e9)
            ent["frac"] = ent["achieved_gbs"] / PEAK_HBM_GBS
        kernels.append(ent)
    kernels.sort(key=lambda e: -e["avg_ms"] * e["launches_per_step"])

    # HBM traffic per launch from the committed rocprofv3 PMC summary of this same command (profiles/, separate
    # --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE exact)
    pmc, pmc_file = {}, None
    for rnd in ("r03", "r02"):                                  # the newest committed summary of this workload
        cand = f"profiles/{rnd}_{args.workload}_pmc.json"
        try:
            with open(os.path.join(ROOT, cand)) as f:
                pmc = json.load(f).get("kernels", {})
            pmc_file = cand
            break
        except Exception:
            continue
    traffic_keys = {}                                   # which entry of the profile file each traffic figure came from (staleness is visible)

    def traffic_for(kernel_prefix, grid_threads, who=None):
        for name, e in pmc.items():
            if name.startswith(kernel_prefix) and name.endswith(f"grid={grid_threads}"):
                if "fetch_bytes_corrected_max" in e and "write_bytes_max" in e:
                    if who:
                        traffic_keys[who] = f"{pmc_file}: {name}"
                    return e["fetch_bytes_corrected_max"] + e["write_bytes_max"]
        if who:
            traffic_keys[who] = f"{pmc_file}: no entry for '{kernel_prefix}' grid={grid_threads} (launch shape changed since the profile was taken)"
        return None

    for ent in kernels:
        nm = ent["kernel"]
        if nm.startswith("K4b.prefilter_mc.mip") and world == 1:
            size = max(1, spec_size >> int(nm.rsplit("mip", 1)[1]))
            if size >= 512:            # region kernel: 1024 threads per 16x16 tile (4 sample slices per texel)
                ent["traffic"] = traffic_for("void k_mc_region<", 6 * size * size * 4, who=nm)
        elif nm.startswith("K4a.") and world == 1:
            ent["traffic"] = traffic_for("k_prefilter_copy", 6 * spec_size * spec_size, who=nm)
        elif nm == "K2.mip_chain" and world == 1:
            ent["traffic"] = None

    roofline = None
    if world == 1 and kernels:
        dom = next((k for k in kernels if "frac" in k), None)
        if dom is not None:
            if dom["bound"] == "valu":
                # the contract's two values are "hbm" | "mfma": the compute-side bound is priced against the dense fp32 peak, which on gfx950 is
                # the same 157.3 TFLOP/s for the matrix and the vector pipe; `pipe` says which one the kernel actually runs on
                roofline = {"kernel": dom["kernel"], "bound": "mfma", "pipe": "fp32 VALU", "achieved": dom["achieved_tflops"],
                            "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"], "traffic": dom.get("traffic"),
                            "traffic_source": traffic_keys.get(dom["kernel"]),
                            "note": "compute-bound Monte-Carlo kernel (SURVEY S9): priced against the dense fp32 vector peak (157.3 TFLOP/s at "
                                    "2.4 GHz); 65 algorithmic flop per non-zero-weight sample (SURVEY 8d) x 6 x size^2 x samples per launch. "
                                    "Taps come from LDS-staged regions of the source level (k_mc_region); measured limiter: VALU issue "
                                    "(~39 instructions per sample and lane at ~3 clk each). HBM-shaped kernels: roofline_hbm"}
            else:
                roofline = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["achieved_gbs"], "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": dom["frac"], "traffic": dom.get("traffic"), "traffic_source": traffic_keys.get(dom["kernel"])}
    roofline_hbm = [{"kernel": k["kernel"], "bound": "hbm", "achieved": k["achieved_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": k["frac"], "avg_ms": k["avg_ms"]} for k in kernels if k.get("bound") == "hbm"]

    extra = {}
    if rank == 0 and not args.no_shade:
        try:
            extra["shade"] = shade_bench(L, pbrhip, env_tex, world, gb_file=gb_file)
        except Exception as e:      # the headline number must not depend on the extra
            extra["shade_error"] = repr(e)
        try:
            extra["shade_live"] = live_shade_bench(L, pbrhip, maps)
        except Exception as e:
            extra["shade_live_error"] = repr(e)
        try:
            extra["post_process"] = post_bench(L, pbrhip)
        except Exception as e:
            extra["post_process_error"] = repr(e)
        try:
            extra["frame_loop"] = frame_loop_bench(L, pbrhip, env_tex)
        except Exception as e:
            extra["frame_loop_error"] = repr(e)
        try:
            extra["lightgrid_sweep"] = sweep_bench(L, pbrhip)
        except Exception as e:
            extra["lightgrid_sweep_error"] = repr(e)
        try:
            extra["lut_c1"] = lut_bench(L, pbrhip)
        except Exception as e:
            extra["lut_c1_error"] = repr(e)

    if not args.no_shade and not args.no_c5:                      # every rank takes part (screen bands + gather)
        c5 = shade_c5_bench(L, pbrhip, env_tex, rank, world, backend, torch, dist, c5_gbd, frames=args.c5_frames, comm=comm)
        if rank == 0:
            extra["shade_c5"] = c5

    if rank == 0 and world == 1 and args.bounded_cut:
        # NOT the headline and never the default (the reference sums every sample): the same job with the tolerance-budgeted sample cut
        # (GPUX_SetPrefilterTolerance(1e-7): a rigorous per-texel bound from the weight table and each source level's measured range)
        try:
            exact_copy = spec_mem.clone()
            L.GPUX_SetPrefilterTolerance(1e-7)
            step(); sync()
            t0c = time.perf_counter()
            for _ in range(3):
                step()
            sync()
            cut_ms = (time.perf_counter() - t0c) / 3 * 1e3
            kept = {f"mip{m}": int(L.GPUX_PrefilterKeptSamples(m)) for m in range(1, min(n_mips, 6))}
            denom = torch.clamp(exact_copy.abs(), min=1e-3)
            dmax = float(((spec_mem - exact_copy).abs() / denom).max().item())
            extra["c4_bounded_cut"] = {"note": "opt-in GPUX_SetPrefilterTolerance(1e-7): samples whose total weight x max(level) is below 1e-7 x kept weight x min(level) are "
                                               "dropped (rigorous per-texel bound); not the headline, not the reference's arithmetic",
                                       "ms_per_step": cut_ms, "mtexels_per_s": total_texels / cut_ms / 1e3, "kept_samples": kept,
                                       "max_rel_diff_vs_exact_sum": dmax}
            del exact_copy
        except Exception as e:
            extra["c4_bounded_cut_error"] = repr(e)
        finally:
            L.GPUX_SetPrefilterTolerance(0.0)
            step(); sync()                                                        # leave the exact maps in place for what follows

    if os.environ.get("PBR_MC_STATS") == "1":         # self-check of the region kernel: wave-slices recomputed with direct loads (must be 0)
        st = (C.c_uint64 * 2)()
        if L.pbrk_mc_region_stats(st, 0) == 0:
            extra["mc_region_recomputed_wave_slices"] = [int(st[0]), int(st[1])]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(env, W, spec_size, irr_size)

    if args.check and rank == 0:
        import pbr_oracle as O
        pyr = O.build_pyramid(env)
        worst = 0.0
        for mip in range(min(n_mips, 6)):
            size = max(1, spec_size >> mip)
            off = L.pbrk_level_offset(spec_size, mip) * 4
            got = spec_mem[off: off + 6 * size * size * 4].cpu().numpy().reshape(6, size, size, 4)
            for (f, y) in ((0, 0), (3, size // 2), (5, size - 1)):
                want = O.prefilter_mip(pyr, W, spec_size, mip, faces=(f, f + 1), rows=(y, y + 1))[f, y]
                err = np.abs(got[f, y].astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
                worst = max(worst, float(err.max()))
        extra["check_max_rel_err_vs_oracle"] = worst
        if world > 1:      # the gathered result must equal what one GPU computes alone, bit for bit
            gathered = spec_mem.clone()
            one, n_one = pbrhip.partition(spec_size, 1, irr_size, W, 1, 0)
            L.PBR_RecordUnits(pipes, graph, arena, env_tex, C.byref(maps), one, n_one)
            L.GPU_GraphSubmit(graph); L.GPU_GraphWait(graph); L.GPU_ResetDescriptorArena(arena)
            extra["check_gather_equals_single_gpu"] = bool(torch.equal(gathered, spec_mem))
        if world == 1:
            # How far does the job move between the two cube-sampler conventions the reference permits?  The same job once more
            # with tap coordinates snapped to 1/256 texel (general kernels), against the exact-weight result above; the exact maps are restored.
            try:
                exact_spec, exact_irr = spec_mem.clone(), irr_mem.clone()
                L.pbrk_set_cube_sampler_snap(1)
                step(); sync()
                delta = {}
                for mip in range(min(n_mips, 5)):
                    size = max(1, spec_size >> mip)
                    off = L.pbrk_level_offset(spec_size, mip) * 4
                    a = spec_mem[off: off + 6 * size * size * 4].view(-1, 4)[:, :3].double(); b = exact_spec[off: off + 6 * size * size * 4].view(-1, 4)[:, :3].double()
                    e = (a - b).abs() / torch.clamp(b.abs(), min=1e-3)
                    delta[f"prefilter_mip{mip}"] = {"max_rel": float(e.max().item()), "rms_rel": float(e.pow(2).mean().sqrt().item())}
                a = irr_mem.view(-1, 4)[:, :3].double(); b = exact_irr.view(-1, 4)[:, :3].double()
                e = (a - b).abs() / torch.clamp(b.abs(), min=1e-3)
                delta["irradiance"] = {"max_rel": float(e.max().item()), "rms_rel": float(e.pow(2).mean().sqrt().item())}
                extra["check_sampler_convention_delta"] = delta
            except Exception as e:
                extra["check_sampler_convention_delta_error"] = repr(e)
            finally:
                L.pbrk_set_cube_sampler_snap(0)
                step(); sync()

    # SURVEY 8(d) headline rates of the whole job (all ranks): Monte-Carlo texels only, sample evaluations, algorithmic bytes
    mc_texels = sum(6 * max(1, spec_size >> m) ** 2 for m in range(1, n_mips)) + 6 * irr_size * irr_size
    sample_evals = sum(6.0 * max(1, spec_size >> m) ** 2 * nonzero_weight_count(L, 8192, ref_roughness(m)) for m in range(1, n_mips)) \
        + 6.0 * irr_size * irr_size * 1024
    alg_bytes = 16.0 * total_texels + 16.0 * 6 * (W // 2) ** 2 + 160.0 * W * W        # outputs + copy-level source + mip chain
    rates = {"mtexels_per_s_all_mips": value, "mtexels_per_s_mc_mips": mc_texels * args.steps / elapsed / 1e6,
             "msamples_per_s": sample_evals * args.steps / elapsed / 1e6,
             "valu_fraction_of_fp32_peak": sample_evals * FLOP_PER_SAMPLE * args.steps / elapsed / 1e12 / (PEAK_FP32_TFLOPS * world),
             "hbm_gbs_by_algorithmic_bytes": alg_bytes * args.steps / elapsed / 1e9,
             "hbm_fraction_by_algorithmic_bytes": alg_bytes * args.steps / elapsed / 1e9 / (PEAK_HBM_GBS * world),
             "note": "the job is compute-bound by construction (up to 8192 sample evaluations per 16-byte texel): the HBM fraction is tiny"}

    if rank == 0:
        out = {
            "metric": "IBL-prefilter Mtexels/s (specular prefilter all mips + irradiance; PBR-shaded Mpixels/s under extra.shade_c5 / extra.shade)",
            "value": value, "unit": "Mtexels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "file" if (hdr_info or gb_file) else "synthetic",
            "inputs": ({"hdr": hdr_info} if hdr_info else {}) | ({"gbuffer": gb_file["info"]} if gb_file else {}) or None,
            "config": {"workload": desc, "env": (f"{os.path.basename(args.hdr)} sha256 {hdr_info['sha256']}" if hdr_info else
                                                 f"procedural HDR cube {W}^2 x6 RGBA32F (seed {seed:#x}, RGBE round-tripped)"),
                       "texels_per_step": total_texels, "sample_evaluations_per_step": sample_evals,
                       "overlap": overlap, "parallelism": "single GPU" if world == 1 else f"{world} ranks, weighted linear partition of output rows (3-8 dispatches per rank), 1 grouped RCCL send/recv gather per step"},
            "roofline": roofline, "roofline_hbm": roofline_hbm, "rates": rates, "step_split": step_split, "rccl": rccl_report, "kernels": kernels[:12],
            "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(out))

    L.GPU_DestroyGraph(graph)
    L.GPU_DestroyDescriptorArena(arena)
    L.PBR_DestroyIBLPipelines(pipes)
    L.GPU_DestroyTexture(maps.tex_specular_env_map); L.GPU_DestroyTexture(maps.irradiance_map); L.GPU_DestroyTexture(maps.brdf_lut)
    L.GPU_DestroyTexture(env_tex)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def shade_bench(L, pbrhip, env_tex, world, frames=20, gb_file=None):
    """C3: 1920x1080 synthetic metal-rough-spheres G-buffer through the lighting pass (K5); IBL maps at the reference's sizes
    (render.cpp:794-796: 32^2 irradiance, 256^2 LUT, 256^2 prefiltered cube with mips down to 16^2) from this run's environment."""
    from pbrhip import synth
    if gb_file is not None:                                                # --gbuffer DIR: the planes as the raster passes would have left them
        gbd = gb_file
        H, W = gbd["depth"].shape
        cam = gbd["camera"]
    else:
        W, H = 1920, 1080
        gbd = synth.synth_gbuffer_spheres(W, H)
        cam = {"pos": gbd["cam_pos"], "ori_xyzw": None, "fov": 75.0}
    maps = pbrhip.PBR_IBLMaps()
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
    L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16)
    L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    for name, arr in (("base_color", gbd["base"]), ("normal", gbd["normal"]), ("orm", gbd["orm"]),
                      ("emissive", gbd["emissive"]), ("depth", gbd["depth"])):
        pbrhip.upload_mip(getattr(gb, name), 0, arr)
    lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    glob = pbrhip.fill_globals(cam["pos"], ori=cam.get("ori_xyzw"), fov=float(cam.get("fov", 75.0)), aspect=W / H)
    g = L.GPU_MakeGraph()
    L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)                      # warm-up (builds the aprons)
    for _ in range(frames):
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    t0 = time.perf_counter()
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    wall = time.perf_counter() - t0
    ms = [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g))
          if L.GPUX_GraphTimedOpName(g, i).decode() == "K5.shade"]
    k_ms = float(np.mean(ms)) if ms else float("nan")
    byt = 28.0 * W * H
    res = {"workload": (f"G-buffer from {gb_file['info']['dir']} ({W}x{H}), Cook-Torrance shade pass (IBL mode), RGBA16F target" if gb_file is not None
                        else "C3: 1920x1080 G-buffer Cook-Torrance shade pass (IBL mode), RGBA16F target"), "frames": frames,
           "kernel_avg_ms": k_ms, "mpixels_per_s_kernel": W * H / (k_ms * 1e-3) / 1e6,
           "mpixels_per_s_wall": W * H * frames / wall / 1e6,
           "roofline": {"kernel": "K5.shade", "bound": "hbm", "achieved": byt / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": byt / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb)); L.PBR_DestroyIBLMaps(C.byref(maps))
    return res


def shade_c5_bench(L, pbrhip, env_tex, rank, world, backend, torch, dist, gbd, frames=10, comm=None):
    """C5: 7680x4320 synthetic 'temple' G-buffer, deferred shade split into horizontal screen bands over the ranks, bands
    gathered to rank 0 every frame (one grouped exchange).  IBL maps at the reference's sizes are computed redundantly on
    every rank (9 MB: cheaper than communicating).  Collective-safe: ranks agree on success before the timed loop."""
    from pbrhip import synth
    W, H = 7680, 4320
    r0, r1 = H * rank // world, H * (rank + 1) // world
    ok, err, res = 1, None, None
    try:
        maps = pbrhip.PBR_IBLMaps()
        L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)                         # render.cpp:794-796
        L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
        L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16)
        L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
        out_mem = torch.zeros(W * H * 4, dtype=torch.float16, device="cuda")   # RGBA16F frame over torch-owned HBM (RCCL moves bands)
        gb = pbrhip.PBR_GBuffer()
        rt = pbrhip.TextureFlag_RenderTarget
        gb.base_color = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt); gb.normal = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt)
        gb.orm = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt); gb.emissive = pbrhip.make_texture(pbrhip.Format_RGBA8UN, W, H, rt)
        gb.depth = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, W, H, rt)
        gb.lighting_result = L.GPUX_MakeTextureExternal(pbrhip.Format_RGBA16F, W, H, 1, rt, out_mem.data_ptr(), out_mem.numel() * 2)
        for name, arr in (("base_color", gbd["base"]), ("normal", gbd["normal"]), ("orm", gbd["orm"]), ("emissive", gbd["emissive"]), ("depth", gbd["depth"])):
            pbrhip.upload_mip(getattr(gb, name), 0, arr)
        lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
        glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H)
        g = L.GPU_MakeGraph()
    except Exception as e:                                                      # pragma: no cover (reported, never raised)
        ok, err = 0, repr(e)
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item()) if ok else 0
    if not ok:
        return {"error": err or "another rank failed during setup"}

    band = lambda r: out_mem[(H * r // world) * W * 4: (H * (r + 1) // world) * W * 4]

    def frame(gather=True):
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), r0, r1)
        L.GPU_GraphSubmit(g)
        if world > 1 and gather and comm is not None:                             # C host layer: bands -> rank 0 behind the kernel, same stream
            if L.PBR_GatherBands(comm, L.GPUX_GraphStream(g), 0, world, rank, gb.lighting_result) < 0:
                raise RuntimeError("PBR_GatherBands failed")
        L.GPU_GraphWait(g)
        if world > 1 and gather and comm is None:
            ops, staged = [], []
            if rank == 0:
                for r in range(1, world):
                    dst = band(r)
                    buf = dst if backend == "nccl" else torch.empty(dst.shape, dtype=dst.dtype)
                    staged.append((dst, buf)); ops.append(dist.P2POp(dist.irecv, buf, r))
            else:
                src = band(rank)
                ops.append(dist.P2POp(dist.isend, src if backend == "nccl" else src.cpu(), 0))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if backend == "nccl":
                torch.cuda.current_stream().synchronize()
            for dst, buf in staged:
                if backend != "nccl":
                    dst.copy_(buf)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); L.GPU_WaitUntilIdle()

    frame()                                                                     # warm-up: aprons / cells of the maps, RCCL channels
    sync()
    k_ms = []
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
        k_ms += [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g)) if L.GPUX_GraphTimedOpName(g, i).decode() == "K5.shade"]
    sync()
    elapsed = time.perf_counter() - t0
    t0 = time.perf_counter()                                                    # the same frames without the exchange (bands stay where they are shaded)
    for _ in range(frames):
        frame(gather=False)
    sync()
    elapsed_local = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed, elapsed_local], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_local = float(tt[0].item()), float(tt[1].item())
    if rank == 0:
        k = float(np.mean(k_ms)) if k_ms else float("nan")
        byt = 28.0 * W * (r1 - r0)
        frame_sum = float(torch.nan_to_num(out_mem.float(), posinf=65504.0).sum(dtype=torch.float64).item())
        res = {"workload": "C5: 7680x4320 'temple' G-buffer, Cook-Torrance + IBL shade, RGBA16F target, horizontal bands per rank, "
                           "bands gathered to rank 0 every frame", "n_gpus": world, "frames": frames, "scaling": "strong",
               "ms_per_frame": elapsed / frames * 1e3, "mpixels_per_s": W * H * frames / elapsed / 1e6,
               "mpixels_per_s_without_gather": W * H * frames / elapsed_local / 1e6,
               "gather_bytes_per_frame": 8.0 * W * (H - (r1 - r0)),
               "rank0_band_rows": r1 - r0, "rank0_kernel_avg_ms": k,
               "roofline": {"kernel": "K5.shade", "bound": "hbm", "achieved": byt / (k * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": byt / (k * 1e-3) / 1e9 / PEAK_HBM_GBS},
               "frame_checksum": frame_sum}
    L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp)
    for name in ("base_color", "normal", "orm", "emissive", "depth", "lighting_result"):
        L.GPU_DestroyTexture(getattr(gb, name))
    L.PBR_DestroyIBLMaps(C.byref(maps))
    return res


def live_shade_bench(L, pbrhip, maps, frames=10):
    """N4: the reference's complete live lighting shader (light shafts + sun shadows + voxel-GI ambient / specular with its
    screen-space trace) on the 1920x1080 spheres scene with a voxelised light grid, previous-frame pyramid and sun depth map."""
    from pbrhip import synth
    W, H = 1920, 1080
    gbd, grid, levels, sun = synth.synth_gi_scene(W, H)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    for name, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, name), 0, gbd[key])
    n = grid.shape[0]
    grid_tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, n, n, pbrhip.TextureFlag_StorageImage, depth=n)
    pbrhip.upload_mip(grid_tex, 0, grid)
    prev_tex = pbrhip.make_texture(pbrhip.Format_RGBA16F, levels[0].shape[1], levels[0].shape[0], pbrhip.TextureFlag_RenderTarget | pbrhip.TextureFlag_HasMipmaps)
    for m in range(min(prev_tex.contents.mip_level_count, len(levels))):
        pbrhip.upload_mip(prev_tex, m, levels[m])
    sun_tex = pbrhip.make_texture(pbrhip.Format_D32F_Or_X8D24UN, sun.shape[1], sun.shape[0], pbrhip.TextureFlag_RenderTarget)
    pbrhip.upload_mip(sun_tex, 0, sun)
    lp = L.PBR_MakeLightingPassLive(C.byref(gb), C.byref(maps), W, H, sun_tex, grid_tex, prev_tex)
    L.GPUX_SetShadeFlags(L.PBR_LightingPipeline(lp), pbrhip.Shade_LightShafts | pbrhip.Shade_SunShadows | pbrhip.Shade_VoxelGI)
    glob = pbrhip.fill_globals(synth.GI_SCENE_CAMERA, aspect=W / H, frame_idx=3)
    glob.lightgrid_scale = 1.0 / synth.GI_SCENE_EXTENT
    g = L.GPU_MakeGraph()
    L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    for _ in range(frames):
        L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    ms = [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g)) if L.GPUX_GraphTimedOpName(g, i).decode() == "K5.shade"]
    k_ms = float(np.mean(ms))
    res = {"workload": "N4: 1920x1080 complete live lighting shader (shafts + sun shadows + voxel GI with screen-space trace), RGBA16F target",
           "frames": frames, "kernel_avg_ms": k_ms, "mpixels_per_s_kernel": W * H / (k_ms * 1e-3) / 1e6,
           "surface_pixel_fraction": float((gbd["depth"] < 1).mean()),
           "note": "data-dependent ray marching (2 traces per surface pixel); not an HBM-shaped kernel"}
    L.GPU_DestroyGraph(g); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb))
    for t in (grid_tex, prev_tex, sun_tex):
        L.GPU_DestroyTexture(t)
    return res


def post_bench(L, pbrhip, frames=20):
    """N3: 1920x1080 TAA resolve (K8) + tone-map pass (K9) of render.cpp:1131-1137, 1181-1187 on a synthetic HDR frame."""
    from pbrhip import synth
    W, H = 1920, 1080
    lighting, depth, vel, vel_prev, history = synth.synth_post_inputs(0x5EED00D0, W, H)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    pbrhip.upload_mip(gb.lighting_result, 0, lighting); pbrhip.upload_mip(gb.depth, 0, depth)
    pbrhip.upload_mip(L.PBR_PostVelocity(pp, 0), 0, vel); pbrhip.upload_mip(L.PBR_PostVelocity(pp, 1), 0, vel_prev)
    pbrhip.upload_mip(L.PBR_PostTaaOutput(pp, 1), 0, history)
    g = L.GPU_MakeGraph()

    def record(f):                                                # render.cpp:1131-1187: TAA -> bloom (6 + 6 passes) -> final
        L.PBR_RecordTaaResolve(pp, g, f); L.PBR_RecordBloom(pp, g, f); L.PBR_RecordFinalPostProcessBloom(pp, g, f)
    record(0)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    for f in range(frames):
        record(f)
    t0 = time.perf_counter()
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    wall = time.perf_counter() - t0
    ms = {}
    for i in range(L.GPUX_GraphTimedOpCount(g)):
        ms.setdefault(L.GPUX_GraphTimedOpName(g, i).decode(), []).append(L.GPUX_GraphTimedOpMs(g, i))
    # the same frames once more without the per-op HIP events (each event pair costs ~2.5 us of stream time on ~20 small ops per frame)
    L.GPUX_EnableOpTiming(0)
    for f in range(frames):
        record(f)
    t0 = time.perf_counter()
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    wall_untimed = time.perf_counter() - t0
    L.GPUX_EnableOpTiming(1)
    res = {"workload": "N3: 1920x1080 TAA resolve + bloom chain (6 down, 6 up) + tone map (RGBA16F in, BGRA8 out)", "frames": frames,
           "mpixels_per_s_wall": W * H * frames / wall_untimed / 1e6, "us_per_frame_wall": wall_untimed / frames * 1e6,
           "us_per_frame_wall_with_op_events": wall / frames * 1e6,
           "us_per_frame_kernels": float(sum(sum(v) for v in ms.values()) / frames * 1e3),
           "note": "bloom entries: mean over the 6 passes of a frame (avg_ms and alg_bytes per pass); us_per_frame_kernels is the sum of the per-op HIP-event "
                   "timings of the frame's ~21 ops (each pair of events adds ~2.5 us of stream time to its op): rocprofv3's kernel trace of the same "
                   "frames without events is profiles/r03_post_kernel_trace.txt (tools/post_prof.sh)", "kernels": []}
    # bloom bytes: every pass reads its source level once and writes (upsample: reads + writes) its target, 8 B per texel
    lv = lambda w, h, m: max(1, w >> m) * max(1, h >> m)
    down_b = sum(8.0 * ((W * H if m == 0 else lv(W // 2, H // 2, m - 1)) + lv(W // 2, H // 2, m)) for m in range(6)) / 6
    up_b = sum(8.0 * (lv(W // 2, H // 2, 5) if m == 5 else lv(W, H, m + 1)) + 16.0 * lv(W, H, m) for m in range(6)) / 6
    per_pass = lambda name: [float(np.mean(ms[name][k::6]) * 1e3) for k in range(6)]       # the six passes repeat frame after frame
    res["bloom_downsample_us_by_pass"] = per_pass("K10.bloom_downsample")
    res["bloom_upsample_us_by_pass"] = per_pass("K11.bloom_upsample")
    for name, byt in (("K8.taa_resolve", 36.0 * W * H), ("K10.bloom_downsample", down_b), ("K11.bloom_upsample", up_b), ("K9.final_post_process", 12.0 * W * H)):
        k_ms = float(np.mean(ms[name]))
        res["kernels"].append({"kernel": name, "avg_ms": k_ms, "bound": "hbm", "alg_bytes": byt,
                               "achieved": byt / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": byt / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS})
    L.GPU_DestroyGraph(g); L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))
    return res


def frame_loop_bench(L, pbrhip, env_tex, frames=100):
    """The demo's frame loop at 1920x1080, default (IBL) mode: light-grid sweep (K7) + shade (K5) + TAA resolve (K8) + bloom chain (K10 / K11)
    + tone map (K9), frame f recorded and submitted while frame f - 1 runs (two graphs in flight, main.cpp:49-51, 91-99); wall time per
    frame without per-op events.  tools/frame_chain_time.py is the same loop as a stand-alone tool (with the hipGraph replay variant)."""
    from pbrhip import synth
    W, H = 1920, 1080
    gbd = synth.synth_gbuffer_spheres(W, H)
    _, _, vel, vel_prev, history = synth.synth_post_inputs(0x5EED00D0, W, H)
    scene = synth.synth_lightgrid(128, lit=False).view(np.uint16)
    maps = pbrhip.PBR_IBLMaps()                                   # the reference's map sizes (render.cpp:794-796), as in shade_bench
    L.PBR_MakeIBLMaps(C.byref(maps), 32, 256, 256)
    L.PBR_GenIrradianceMap(env_tex, maps.irradiance_map)
    L.PBR_GenPrefilteredEnvMap(env_tex, maps.tex_specular_env_map, 16)
    L.PBR_GenBRDFIntegrationMap(maps.brdf_lut)
    gb = pbrhip.PBR_GBuffer()
    L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
    for nm, key in (("base_color", "base"), ("normal", "normal"), ("orm", "orm"), ("emissive", "emissive"), ("depth", "depth")):
        pbrhip.upload_mip(getattr(gb, nm), 0, gbd[key])
    lp = L.PBR_MakeLightingPass(C.byref(gb), C.byref(maps), W, H)
    pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
    pbrhip.upload_mip(L.PBR_PostVelocity(pp, 0), 0, vel); pbrhip.upload_mip(L.PBR_PostVelocity(pp, 1), 0, vel_prev)
    pbrhip.upload_mip(L.PBR_PostTaaOutput(pp, 1), 0, history)
    lg = L.PBR_MakeLightgrid(128)
    pbrhip.upload_mip(L.PBR_LightgridTexture(lg), 0, scene)
    graphs = [L.GPU_MakeGraph(), L.GPU_MakeGraph()]
    L.GPUX_EnableOpTiming(0)
    try:
        t0 = None
        for f in range(frames + 8):
            if f == 8:
                L.GPU_WaitUntilIdle(); t0 = time.perf_counter()
            g = graphs[f % 2]
            if f >= 2:
                L.GPU_GraphWait(g)                               # the frame before last
            glob = pbrhip.fill_globals(gbd["cam_pos"], aspect=W / H, frame_idx=f % 59)
            L.PBR_RecordLightgridSweep(lg, g)
            L.PBR_RecordLightingPass(lp, g, C.byref(glob), 0, 0)
            L.PBR_RecordTaaResolve(pp, g, f); L.PBR_RecordBloom(pp, g, f); L.PBR_RecordFinalPostProcessBloom(pp, g, f)
            L.GPU_GraphSubmit(g)
        for g in graphs:
            L.GPU_GraphWait(g)
        per = (time.perf_counter() - t0) / frames
    finally:
        L.GPUX_EnableOpTiming(1)
    bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
    res = {"workload": "1920x1080 frame loop, IBL mode: light-grid sweep + shade + TAA + bloom (6 + 6) + tone map, two graphs in flight", "frames": frames,
           "us_per_frame": per * 1e6, "frames_per_s": 1.0 / per, "mpixels_per_s": W * H / per / 1e6, "backbuffer_checksum": int(bb.astype(np.uint64).sum())}
    for g in graphs:
        L.GPU_DestroyGraph(g)
    L.PBR_DestroyLightgrid(lg); L.PBR_DestroyPostProcess(pp); L.PBR_DestroyLightingPass(lp); L.PBR_DestroyGBuffer(C.byref(gb))
    L.PBR_DestroyIBLMaps(C.byref(maps))
    return res


def lut_bench(L, pbrhip, runs=10):
    """C1: the 256^2 split-sum BRDF LUT (K1, gen_brdf_integration_map.glsl:142-210; render.cpp:591-619), 4096 samples per texel."""
    t = pbrhip.make_texture(pbrhip.Format_RG16F, 256, 256, pbrhip.TextureFlag_StorageImage)
    pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena(); g = L.GPU_MakeGraph()
    maps = pbrhip.PBR_IBLMaps(); maps.brdf_lut = t
    u = (pbrhip.PBR_WorkUnit * 1)(); u[0].kind = pbrhip.Unit_BrdfLut; u[0].row0 = 0; u[0].row1 = 256; u[0].face0 = 0; u[0].face1 = 1
    ms = []
    for _ in range(runs + 1):
        L.PBR_RecordUnits(pipes, g, arena, None, C.byref(maps), u, 1)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_ResetDescriptorArena(arena)
        ms += [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g))]
    k_ms = float(np.median(ms[1:]))
    evals = 256.0 * 256.0 * 4096.0
    res = {"workload": "C1: 256x256 split-sum BRDF LUT, 4096 samples per texel, RG16F target", "runs": runs, "kernel_ms": k_ms,
           "mtexels_per_s": 65536.0 / (k_ms * 1e-3) / 1e6, "msamples_per_s": evals / (k_ms * 1e-3) / 1e6,
           # executed work per (sample, texel): the row part (6 mul + 2 FMA = 10 flop, one v_exp_f32) + 1/16 of the column part (~60 flop, 3 transcendentals)
           "roofline": {"kernel": "K1.brdf_lut", "bound": "mfma", "pipe": "fp32 VALU", "achieved": evals * 13.75 / (k_ms * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                        "frac": evals * 13.75 / (k_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                        "shader_equivalent_tflops": evals * 60.0 / (k_ms * 1e-3) / 1e12,
                        "note": "priced by the work the kernel executes (13.75 flop + 1.2 transcendentals per sample and texel: the roughness-independent part of a "
                                "sample is evaluated once per column x 16 rows); the shader's own 60 flop per sample (SURVEY 8d) would read shader_equivalent_tflops"}}
    L.GPU_DestroyGraph(g); L.GPU_DestroyDescriptorArena(arena); L.PBR_DestroyIBLPipelines(pipes); L.GPU_DestroyTexture(t)
    return res


def sweep_bench(L, pbrhip, frames=30):
    """N2: 128^3 RGBA16F light grid, the per-frame sweep of render.cpp:1061-1072 (K7), directions cycling y, z, x."""
    from pbrhip import synth
    n = 128
    scene = synth.synth_lightgrid(n, lit=False).view(np.uint16)
    lg = L.PBR_MakeLightgrid(n)
    tex = L.PBR_LightgridTexture(lg)
    pbrhip.upload_mip(tex, 0, scene)
    g = L.GPU_MakeGraph()
    for _ in range(3):
        L.PBR_RecordLightgridSweep(lg, g)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)                      # warm-up
    for _ in range(frames):
        L.PBR_RecordLightgridSweep(lg, g)
    t0 = time.perf_counter()
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    wall = time.perf_counter() - t0
    per_dir = {}
    for i in range(L.GPUX_GraphTimedOpCount(g)):
        per_dir.setdefault(L.GPUX_GraphTimedOpName(g, i).decode(), []).append(L.GPUX_GraphTimedOpMs(g, i))
    empty = float((scene[..., 3].view(np.float16) < 0.5).mean())
    byt = n ** 3 * 8.0 * (1.0 + empty)                            # 8 B read per voxel + 8 B written per empty voxel
    k_ms = float(np.mean([v for vs in per_dir.values() for v in vs]))
    res = {"workload": "N2: 128^3 RGBA16F light-grid sweep, one direction per frame", "frames": frames,
           "kernel_avg_ms": k_ms, "kernel_avg_ms_by_direction": {k: float(np.mean(v)) for k, v in sorted(per_dir.items())},
           "mvoxels_per_s_kernel": n ** 3 / (k_ms * 1e-3) / 1e6, "mvoxels_per_s_wall": n ** 3 * frames / wall / 1e6,
           "empty_voxel_fraction": empty,
           "roofline": {"kernel": "K7.sweep", "bound": "hbm", "achieved": byt / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": byt / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    L.GPU_DestroyGraph(g); L.PBR_DestroyLightgrid(lg)
    return res


if __name__ == "__main__":
    main()
