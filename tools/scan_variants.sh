#!/bin/bash
for w in 8 9 10 12 16; do
  echo "== MCALLER_SCAN_WGS=$w"
  MCALLER_SCAN_WGS=$w python tools/k1_experiments.py 1e8 | tail -1 | cut -c1-120
done
unset MCALLER_SCAN_WGS; MCALLER_VERBOSE=1 python tools/k1_experiments.py 1e8 2>&1 | grep -E "occupancy|debug" | tail -2 | cut -c1-120
