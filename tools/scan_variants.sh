#!/bin/bash
# Timing of k1_scan for a few build variants (run on the GPU box).
for cfg in "4096 3" "4096 4" "2048 3"; do
  set -- $cfg
  MC_TILE=$1 MC_SCAN_WGS=$2 python -c "from mcaller_amd.build import build_lib; build_lib(force=True, verbose=False)" 2>/dev/null || exit 1
  echo "== TILE=$1 WGS=$2"
  MC_TILE=$1 python tools/k1_experiments.py 1e8 | tail -1
done
