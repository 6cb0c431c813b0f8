# SQ counters of the post-process tail's kernels at 1920x1080 (per kernel and grid size = per pass):  bash tools/post_pmc.sh
set -e
export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/postpmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/postpmc -- python3 $R/tools/post_time.py 4 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("/tmp/postpmc/**/*_counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
    if not m: continue
    per[(m.group(1), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(per.items()):
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    w = max(med.get("SQ_WAVES", 1.0), 1.0)
    print(f"{k[0][:34]:34s} grid {k[1]:>8}  waves {w:7.0f}  VALU/wave {med['SQ_INSTS_VALU']/w:7.0f}  SALU {med['SQ_INSTS_SALU']/w:5.0f}  loads {med['SQ_INSTS_VMEM_RD']/w:5.1f}  "
          f"wave_cycles/wave {med['SQ_WAVE_CYCLES']/w:8.0f}  wait_any {med['SQ_WAIT_ANY']/w:8.0f}  wait_inst {med['SQ_WAIT_INST_ANY']/w:8.0f}  busy_cycles {med['SQ_BUSY_CYCLES']:10.0f}")
PY
