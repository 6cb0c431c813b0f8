#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc SQ_* pass (counter_collection.csv) into profiles/<tag>_sq_counters.json: per (kernel, grid size)
the summed counters of the largest launch, plus VALU instructions per wave.   usage: summarize_sq.py <tag> <dir> [<dir2> ...]"""
import collections, csv, glob, json, os, sys


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    res = {"note": "rocprofv3 --kernel-trace --pmc SQ_* (one pass per counter set); values are sums over the chip for ONE launch (the "
                   "longest of that kernel / grid size). SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md).", "kernels": {}}
    for d in dirs:
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        if not f:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(dict))         # key -> dispatch id -> counter -> value
        for r in csv.DictReader(open(f[0])):
            key = r["Kernel_Name"].split("(")[0] + " grid=" + r["Grid_Size"]
            per[key][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        for key, disp in per.items():
            best = max(disp.values(), key=lambda c: c.get("SQ_WAVE_CYCLES", c.get("SQ_BUSY_CYCLES", 0.0)))
            e = res["kernels"].setdefault(key, {})
            e.update(best); e["launches"] = len(disp)
    for e in res["kernels"].values():
        if e.get("SQ_WAVES"):
            if "SQ_INSTS_VALU" in e:
                e["valu_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
            if "SQ_INSTS_SALU" in e:
                e["salu_per_wave"] = e["SQ_INSTS_SALU"] / e["SQ_WAVES"]
    json.dump(res, open(os.path.join(out, f"{tag}_sq_counters.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(out, f"{tag}_sq_counters.json"))


if __name__ == "__main__":
    main()
