#!/usr/bin/env python3
"""The post-process tail alone at 1920x1080 -- TAA resolve (K8), bloom chain (K10 x 6, clear, blit, K11 x 6), tone map (K9) --
recorded N frames into one graph, plain stream launches (for rocprofv3 --kernel-trace --stats: per-kernel durations without the
per-op event overhead of bench.py's extra.post_process).   python3 tools/post_time.py [frames]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python"))
from pbrhip import synth  # noqa: E402

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = 1920, 1080
lighting, depth, vel, vel_prev, history = synth.synth_post_inputs(0x5EED00D0, W, H)
import pbrhip  # noqa: E402
if os.environ.get("PBRHIP_LIB"):
    pbrhip.LIB_PATH = os.environ["PBRHIP_LIB"]

L = pbrhip.init(0)
gb = pbrhip.PBR_GBuffer()
L.PBR_MakeGBuffer(C.byref(gb), W, H, pbrhip.Format_RGBA16F)
pp = L.PBR_MakePostProcess(C.byref(gb), W, H, pbrhip.Format_BGRA8UN)
pbrhip.upload_mip(gb.lighting_result, 0, lighting); pbrhip.upload_mip(gb.depth, 0, depth)
pbrhip.upload_mip(L.PBR_PostVelocity(pp, 0), 0, vel); pbrhip.upload_mip(L.PBR_PostVelocity(pp, 1), 0, vel_prev)
pbrhip.upload_mip(L.PBR_PostTaaOutput(pp, 1), 0, history)
g = L.GPU_MakeGraph()
for rep in range(3):
    for f in range(FRAMES):
        L.PBR_RecordTaaResolve(pp, g, f); L.PBR_RecordBloom(pp, g, f); L.PBR_RecordFinalPostProcessBloom(pp, g, f)
    t0 = time.perf_counter()
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g)
    dt = time.perf_counter() - t0
    print(f"post tail: {dt / FRAMES * 1e6:.1f} us per 1080p frame (wall, {FRAMES} frames back to back)", flush=True)
bb = pbrhip.read_mip(L.PBR_PostBackbuffer(pp), 0)
print("backbuffer checksum", int(bb.astype(np.uint64).sum()))
L.GPU_DestroyGraph(g); L.PBR_DestroyPostProcess(pp); L.PBR_DestroyGBuffer(C.byref(gb))
L.GPU_WaitUntilIdle(); L.GPU_Deinit()
