# K5 on the 8K temple frame: SQ / TA / TCP counters of the shipped fast kernel and of the opt-in tiled kernel (round 3)
#   bash tools/k5_pmc_r3.sh   -> gpurun_out/k5pmc_r3.txt
set -e
export TMPDIR=/tmp PROBE_WORKERS=1
R=$PWD
cd /tmp
P="python3 $R/tools/shade_probe.py 7680 4320"
: > $R/gpurun_out/k5pmc_r3.txt
for mode in fast tile; do
  if [ $mode = tile ]; then export PBR_SHADE_TILE_MIN_PIXELS=3000000; else export PBR_SHADE_TILE_MIN_PIXELS=100000000000; fi
  rm -rf /tmp/k5a /tmp/k5b /tmp/k5c
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/k5a -- $P > $R/gpurun_out/k5pmc_r3_a.log 2>&1
  rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/k5b -- $P > $R/gpurun_out/k5pmc_r3_b.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/k5c -- $P > $R/gpurun_out/k5pmc_r3_c.log 2>&1
  echo "== $mode" >> $R/gpurun_out/k5pmc_r3.txt
  grep temple $R/gpurun_out/k5pmc_r3_a.log >> $R/gpurun_out/k5pmc_r3.txt || true
  python3 - >> $R/gpurun_out/k5pmc_r3.txt <<'PY'
import csv, glob, collections
agg = {}
for d in "abc":
    f = glob.glob(f"/tmp/k5{d}/**/*_counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "k_shade" in r["Kernel_Name"] and r["Grid_Size"] in ("33177600", "34078720", "33423360"):
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if per:
        agg.update(per[sorted(per, key=int)[-3]])
w = agg.get("SQ_WAVES", 1.0)
for k in sorted(agg):
    print(f"  {k:32s} {agg[k]:16.0f}   per wave {agg[k] / w:10.1f}")
PY
done
cat $R/gpurun_out/k5pmc_r3.txt
