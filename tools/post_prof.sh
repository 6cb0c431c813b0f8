# kernel durations of the post-process tail at 1920x1080 from rocprofv3 --kernel-trace (per kernel and grid size = per pass)
#   bash tools/post_prof.sh [libgpu_hip variant ...]
set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/post
for lib in "$@"; do
  export PBRHIP_LIB=$R/vulkan-pbr-renderer_amd/$lib
  python3 $R/tools/post_time.py 100 > $R/gpurun_out/post/wall_$lib.log 2>&1
  cd /tmp; rm -rf /tmp/postkt
  rocprofv3 --kernel-trace --output-format csv -d /tmp/postkt -- python3 $R/tools/post_time.py 50 > $R/gpurun_out/post/chain_$lib.log 2>&1
  f=$(find /tmp/postkt -name "*kernel_trace.csv" | head -1)
  echo "== $lib"; tail -2 $R/gpurun_out/post/wall_$lib.log
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    import re
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    gx = r.get("Grid_Size_X", r.get("Grid_Size", "")); gy = r.get("Grid_Size_Y", "")
    agg[(name, gx, gy)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
per_frame = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v); med = v[len(v) // 2]
    print(f"  {k[0][:40]:40s} grid {k[1]:>6}x{k[2]:<5} n {len(v):4d}  median {med:7.2f} us")
    if len(v) >= 100: per_frame += med * (len(v) / 150.0)
print(f"  sum of medians per frame (150 frames traced): {per_frame:.1f} us")
PY
  cd $R
done
