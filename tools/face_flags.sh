# single-face dispatches of the region kernel, one line per face: time and (PBR_MC_STATS=1) the binning's yield
export PBR_MC_STATS=1
for cfg in "32 512 0.4" "64 1024 0.15" "128 2048 0.03"; do
  set -- $cfg
  for f in 0 1 2; do
    python3 tools/mc_probe.py $1 $2 $3 $2 $f $((f+1)) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tr '\n' ' '; echo
  done
  python3 tools/mc_probe.py $1 $2 $3 2>&1 | grep "n_src"
done
