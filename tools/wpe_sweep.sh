for w in 2 3; do
  MC_SCAN_WPE=$w python -m mcaller_amd.build --force 2>/dev/null
  MCALLER_VERBOSE=1 python tools/k1_experiments.py 1e8 2>&1 | grep -E "occupancy|debug=0" | tail -2 | cut -c1-200 | sed "s/^/WPE=$w /"
done
