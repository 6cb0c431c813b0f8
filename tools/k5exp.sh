set -e
mkdir -p gpurun_out/k5exp
: > gpurun_out/k5exp/log.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_post.py -x -q -k "shade or replay" >> gpurun_out/k5exp/log.txt 2>&1 || { tail -40 gpurun_out/k5exp/log.txt; exit 1; }
for t in 0 1 0 1; do
  echo "== PBR_SHADE_TABLES=$t" >> gpurun_out/k5exp/log.txt
  PBR_SHADE_TABLES=$t python3 tools/shade_probe.py 7680 4320 >> gpurun_out/k5exp/log.txt 2>&1 || true
done
tail -30 gpurun_out/k5exp/log.txt
