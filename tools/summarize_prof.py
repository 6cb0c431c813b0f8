#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel trace + stats, FETCH_SIZE pass, WRITE_SIZE pass) into the
small files committed under profiles/.  usage: summarize_prof.py <tag> <kt_dir> [<fetch_dir> <write_dir>]"""
import collections, csv, glob, json, os, shutil, sys

def find(d, suffix):
    m = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return m[0] if m else None

def main():
    tag, kt = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    stats = find(kt, "_kernel_stats.csv")
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
    res = {"note": "rocprofv3 --kernel-trace --pmc <counter> (separate passes). FETCH_SIZE/WRITE_SIZE are in KiB as reported; "
                   "per MI355X_MICROARCH.md (HBM) FETCH_SIZE reads 1/2 of a wide coalesced 16 B/lane stream on gfx950: "
                   "fetch_bytes_corrected = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16 B/lane stores.", "kernels": {}}
    if len(sys.argv) >= 5:
        for counter, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
                if r["Counter_Name"] == counter:
                    # one entry per (kernel, grid size): the same kernel serves several output levels
                    agg[r["Kernel_Name"].split("(")[0] + " grid=" + r["Grid_Size"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                e = res["kernels"].setdefault(k, {})
                e[counter + "_KiB_mean"] = sum(v) / len(v)
                e[counter + "_KiB_max"] = max(v)
                e["launches"] = len(v)
        for k, e in res["kernels"].items():
            if "FETCH_SIZE_KiB_max" in e:
                e["fetch_bytes_corrected_max"] = 2 * e["FETCH_SIZE_KiB_max"] * 1024
            if "WRITE_SIZE_KiB_max" in e:
                e["write_bytes_max"] = e["WRITE_SIZE_KiB_max"] * 1024
        json.dump(res, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
    print("wrote", out)

if __name__ == "__main__":
    main()
