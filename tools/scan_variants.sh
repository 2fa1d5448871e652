#!/bin/bash
# Timing of k1_scan for a few build variants (run on the GPU box).
for cfg in "4096 1" "4096 6" "4096 8" "2048 6" "2048 8"; do
  set -- $cfg
  MC_TILE=$1 MC_SCAN_WAVES=$2 python -c "from mcaller_amd.build import build_lib; build_lib(force=True, verbose=False)" || exit 1
  echo "== TILE=$1 WAVES=$2"
  MC_TILE=$1 python tools/k1_experiments.py 1e8 | tail -5 | head -1
done
