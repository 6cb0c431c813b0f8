#!/bin/bash
# A/B of two builds of the library on the C4 prefilter shapes (same box, alternating): tools/k4_ab.sh LIB_A LIB_B [rounds]
# Prints tools/mc_probe.py's line per shape and library; checksums must agree between the two.
A=${1:-vulkan-pbr-renderer_amd/libgpu_hip_prevmc.so}; B=${2:-vulkan-pbr-renderer_amd/libgpu_hip.so}; N=${3:-2}
for r in $(seq $N); do
  for lib in $A $B; do
    echo "== $lib"
    for shape in "128 2048 0.03" "64 1024" "32 512" "16 256"; do
      PBRHIP_LIB=$PWD/$lib python3 tools/mc_probe.py $shape 2>&1 | grep -v amdgpu.ids
    done
  done
done
