#!/usr/bin/env python3
"""Compute time of one rank's share of the C4 job on one GPU (no communication): how well the partition's tiles fill the chip.
   python3 tools/rank_time.py <world> [rank ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python"))
import bench  # noqa: E402
import pbrhip  # noqa: E402

world = int(sys.argv[1])
ranks = [int(a) for a in sys.argv[2:]] or list(range(world))
W, spec_size, irr_size, seed, _ = bench.WORKLOADS["c4"]
env = bench.load_env(W, seed, workers=6)
L = pbrhip.init()
env_tex = pbrhip.make_texture(pbrhip.Format_RGBA32F, W, W, pbrhip.TextureFlag_Cubemap | pbrhip.TextureFlag_HasMipmaps, env)
maps = pbrhip.PBR_IBLMaps()
L.PBR_MakeIBLMaps(C.byref(maps), irr_size, 256, spec_size)
pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena(); graph = L.GPU_MakeGraph()
for rank in ranks:
    units, n = pbrhip.partition(spec_size, 1, irr_size, W, world, rank)
    ts = []
    for it in range(3):
        L.GPU_OpGenerateMipmaps(graph, env_tex)
        L.PBR_RecordUnits(pipes, graph, arena, env_tex, C.byref(maps), units, n)
        L.GPU_WaitUntilIdle()
        t0 = time.perf_counter()
        L.GPU_GraphSubmit(graph); L.GPU_GraphWait(graph)
        ts.append((time.perf_counter() - t0) * 1e3)
        L.GPU_ResetDescriptorArena(arena)
    print(f"world {world} rank {rank}: {n} units, {min(ts[1:]):.2f} ms (ideal {137.5 / world:.2f} ms)", flush=True)
    if os.environ.get("RANK_TIME_UNITS") == "1":
        from collections import Counter
        c = Counter((u.kind, u.mip) for u in units[:n])
        cost = Counter()
        for u in units[:n]:
            cost[(u.kind, u.mip)] += u.cost
        print("   ", {k: (c[k], round(cost[k] / 1e9, 2)) for k in sorted(c)}, flush=True)
