#!/usr/bin/env python3
"""K1 alone: time of the 256^2 BRDF LUT and a hash of its fp32 result (LUT_LIB=<other libgpu_hip.so> runs another build: the hash
shows whether two builds agree bit for bit).   python3 tools/lut_time.py"""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vulkan-pbr-renderer_amd", "python"))
import pbrhip, numpy as np
if os.environ.get("LUT_LIB"):
    pbrhip.LIB_PATH = os.environ["LUT_LIB"]
L = pbrhip.init(0); L.GPUX_EnableOpTiming(1)
pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena(); g = L.GPU_MakeGraph()
for fmt, name in ((pbrhip.Format_RG16F, "RG16F"), (pbrhip.Format_RG32F, "RG32F")):
    t = pbrhip.make_texture(fmt, 256, 256, pbrhip.TextureFlag_StorageImage)
    maps = pbrhip.PBR_IBLMaps(); maps.brdf_lut = t
    u = (pbrhip.PBR_WorkUnit * 1)(); u[0].kind = 2; u[0].row0 = 0; u[0].row1 = 256; u[0].face0 = 0; u[0].face1 = 1
    ms = []
    for it in range(6):
        L.PBR_RecordUnits(pipes, g, arena, None, C.byref(maps), u, 1)
        L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_ResetDescriptorArena(arena)
        ms += [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g))]
    full = pbrhip.read_mip(t, 0).copy()
    # ragged row shards == full
    u3 = (pbrhip.PBR_WorkUnit * 3)()
    for k, (r0, r1) in enumerate(((0, 7), (7, 200), (200, 256))):
        u3[k].kind = 2; u3[k].row0 = r0; u3[k].row1 = r1; u3[k].face0 = 0; u3[k].face1 = 1
    L.GPU_OpClearColorF(g, t, 0, 0.0, 0.0, 0.0, 0.0)
    L.PBR_RecordUnits(pipes, g, arena, None, C.byref(maps), u3, 3)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_ResetDescriptorArena(arena)
    shard = pbrhip.read_mip(t, 0)
    print(f"K1 256^2 LUT {name}: ms {[round(m, 4) for m in ms[1:]]}  sha256 {hashlib.sha256(full.tobytes()).hexdigest()[:16]}  ragged shards == full: {np.array_equal(full.view(np.uint8), shard.view(np.uint8))}")
    L.GPU_DestroyTexture(t)
L.GPU_WaitUntilIdle(); L.GPU_Deinit()
