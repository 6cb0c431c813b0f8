#!/bin/bash
# Collects everything profiles/<tag>_* is made of, on the GPU box, in one go:
#   tools/collect_profiles.sh <tag>        (e.g. r02_c4; run from the repo root: `gpurun -- 'bash tools/collect_profiles.sh r02_c4'`)
# Outputs land in gpurun_out/profiles_<tag>/ (gpurun merges that directory back); copy them into profiles/ afterwards.
# Each rocprofv3 pass runs `python3 $B ...` directly after `--` (no wrapper), counters in passes of their own.
set -e -o pipefail
TAG=${1:-r02_c4}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/profiles_$TAG
SCR=/tmp/prof_$TAG
rm -rf "$SCR"; mkdir -p "$OUT" "$SCR"
export TMPDIR=/tmp
cd /tmp                                        # rocprofv3 scratch files go to the working directory
B=$REPO/bench.py

python3 $B --bounded-cut > "$OUT/${TAG}_bench.json"
echo "[1/6] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/kt" -- python3 $B --no-cpu-baseline > "$OUT/${TAG}_bench_under_rocprof.json"
echo "[2/6] kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$SCR/fetch" -- python3 $B --no-cpu-baseline --steps 1 --warmup 0 --no-shade > /dev/null
echo "[3/6] FETCH_SIZE done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$SCR/write" -- python3 $B --no-cpu-baseline --steps 1 --warmup 0 --no-shade > /dev/null
echo "[4/6] WRITE_SIZE done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
    --output-format csv -d "$SCR/sq1" -- python3 $B --no-cpu-baseline --steps 1 --warmup 0 --c5-frames 2 > /dev/null
echo "[5/6] SQ pass 1 done"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$SCR/sq2" -- python3 $B --no-cpu-baseline --steps 1 --warmup 0 --c5-frames 2 > /dev/null
echo "[6/6] SQ pass 2 done"
python3 $REPO/tools/summarize_prof.py "$TAG" "$SCR/kt" "$SCR/fetch" "$SCR/write"
python3 $REPO/tools/summarize_sq.py "$TAG" "$SCR/sq1" "$SCR/sq2"
cd "$REPO"; cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/${TAG}_sq_counters.json "$OUT/"
ls -la "$OUT"
