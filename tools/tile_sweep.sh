for t in 1024 2048 4096; do
  MC_TILE=$t python -m mcaller_amd.build --force 2>/dev/null
  MC_TILE=$t MCALLER_VERBOSE=1 python tools/k1_experiments.py 1e8 2>&1 | grep -E "occupancy|debug=0" | tail -2 | cut -c1-230
done
