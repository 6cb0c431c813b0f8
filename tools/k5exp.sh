set -e
mkdir -p gpurun_out/k5exp
: > gpurun_out/k5exp/log.txt
for rep in 1 2; do for lib in libgpu_hip.so libgpu_hip_prio1.so libgpu_hip_prio2.so; do
  echo "== $lib" >> gpurun_out/k5exp/log.txt
  PBRHIP_LIB=$PWD/vulkan-pbr-renderer_amd/$lib python3 tools/shade_probe.py 7680 4320 >> gpurun_out/k5exp/log.txt 2>&1 || true
done; done
cat gpurun_out/k5exp/log.txt
