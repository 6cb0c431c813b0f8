#!/usr/bin/env python3
"""Time pbrk_mc_filter (K4b) on synthetic levels: picoseconds per sample evaluation and clocks per wave-sample per CU for a
given (n_src, out_size, roughness).  Kernel choice follows the library (env PBR_MC_LDS / PBR_MC_REGION / PBR_MC_BINNED).
   python3 tools/mc_probe.py n_src out_size [roughness] [rows] [face0 face1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python"))
import pbrhip  # noqa: E402
if os.environ.get("PBRHIP_LIB"):
    pbrhip.LIB_PATH = os.environ["PBRHIP_LIB"]


def main():
    n_src, out = int(sys.argv[1]), int(sys.argv[2])
    rough = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else out
    f0, f1 = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 6)
    L = pbrhip.init(0)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    lvl = rng.random((6, n_src, n_src, 4), dtype=np.float32) + 0.1
    pyr = torch.from_numpy(lvl).to(dev)
    bord = torch.zeros((6, n_src + 2, n_src + 2, 4), dtype=torch.float32, device=dev)
    assert L.pbrk_border_build(pyr.data_ptr(), bord.data_ptr(), n_src, 1, None) == 0
    cells = torch.zeros(L.pbrk_cells_bytes(n_src) // 4, dtype=torch.float32, device=dev)
    use_cells = n_src <= 512
    if use_cells:
        assert L.pbrk_cells_build(bord.data_ptr(), n_src, cells.data_ptr(), None) == 0
    tab = np.zeros((8192, 4), dtype=np.float32)
    alpha = C.c_float()
    n_tab = L.pbrk_host_prefilter_table(8192, rough, tab.ctypes.data_as(C.c_void_p), C.byref(alpha))
    dtab = torch.from_numpy(tab).to(dev)
    outt = torch.zeros((6, out, out, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def run():
        rc = L.pbrk_mc_filter(bord.data_ptr(), cells.data_ptr() if use_cells else None, n_src, dtab.data_ptr(), n_tab,
                              float(np.pi), alpha.value, outt.data_ptr(), out, f0, f1, 0, rows, None)
        assert rc == 0, rc

    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = min(ts)
    evals = float(f1 - f0) * out * rows * n_tab
    ps = ms * 1e-3 / evals * 1e12
    clk = ps * 1e-12 * 256 * 2.4e9 * 64
    chk = float(outt[:, :rows].double().sum().item())
    print(f"n_src {n_src} out {out} rows {rows} faces {f0}-{f1} n_tab {n_tab}: {ms:.3f} ms  {ps:.3f} ps/eval  {clk:.1f} clk per wave-sample per CU @2.4GHz  "
          f"{65 * evals / ms * 1e-9:.1f} TFLOP/s alg  checksum {chk:.6e}", flush=True)
    if os.environ.get("PBR_MC_STATS") == "1":
        st = (C.c_uint64 * 2)()
        fl = (C.c_uint64 * 3)()
        if L.pbrk_mc_region_flag_stats(fl) == 0 and fl[1]:
            tiles = fl[1] // n_tab
            wn = (C.c_uint64 * 1)()
            L.pbrk_mc_region_window_stats(wn)
            print(f"   binning: {fl[0] / fl[1]:.3f} regions flagged per sample, {fl[2] / tiles:.2f} regions visited per tile ({tiles} tile launches), "
                  f"{wn[0] / fl[1]:.3f} of the samples proved in-region for the whole tile (test-free body)", flush=True)
        if L.pbrk_mc_region_stats(st, 1) == 0:
            print(f"   region kernel: {st[0]} of {st[1]} wave-slices recomputed with direct loads", flush=True)
    L.GPU_WaitUntilIdle(); L.GPU_Deinit()


if __name__ == "__main__":
    main()
