import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "vulkan-pbr-renderer_amd", "python"))
import pbrhip, numpy as np
L = pbrhip.init(0); L.GPUX_EnableOpTiming(1)
t = pbrhip.make_texture(pbrhip.Format_RG16F, 256, 256, pbrhip.TextureFlag_StorageImage)
pipes = L.PBR_MakeIBLPipelines(); arena = L.GPU_MakeDescriptorArena(); g = L.GPU_MakeGraph()
maps = pbrhip.PBR_IBLMaps(); maps.brdf_lut = t
u = (pbrhip.PBR_WorkUnit * 1)(); u[0].kind = 2; u[0].row0 = 0; u[0].row1 = 256; u[0].face0 = 0; u[0].face1 = 1
ms = []
for it in range(6):
    L.PBR_RecordUnits(pipes, g, arena, None, C.byref(maps), u, 1)
    L.GPU_GraphSubmit(g); L.GPU_GraphWait(g); L.GPU_ResetDescriptorArena(arena)
    ms += [L.GPUX_GraphTimedOpMs(g, i) for i in range(L.GPUX_GraphTimedOpCount(g))]
print("K1 256^2 LUT ms:", [round(m, 4) for m in ms])
