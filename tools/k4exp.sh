set -e
mkdir -p gpurun_out/k4exp
: > gpurun_out/k4exp/log.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q -k "prefilter or sharded or c2 or c4 or ragged or units" >> gpurun_out/k4exp/log.txt 2>&1 || { tail -40 gpurun_out/k4exp/log.txt; exit 1; }
for rep in 1 2; do for lib in libgpu_hip_r2mc.so libgpu_hip.so; do
  echo "== $lib" >> gpurun_out/k4exp/log.txt
  PBRHIP_LIB=$PWD/vulkan-pbr-renderer_amd/$lib python3 tools/mc_probe.py 128 2048 0.03 >> gpurun_out/k4exp/log.txt 2>&1
  PBRHIP_LIB=$PWD/vulkan-pbr-renderer_amd/$lib python3 tools/mc_probe.py 64 1024 0.15 >> gpurun_out/k4exp/log.txt 2>&1
  PBRHIP_LIB=$PWD/vulkan-pbr-renderer_amd/$lib python3 tools/mc_probe.py 32 512 0.4 >> gpurun_out/k4exp/log.txt 2>&1
  PBRHIP_LIB=$PWD/vulkan-pbr-renderer_amd/$lib python3 tools/mc_probe.py 16 256 0.6 >> gpurun_out/k4exp/log.txt 2>&1
done; done
grep -v amdgpu.ids gpurun_out/k4exp/log.txt
