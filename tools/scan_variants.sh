#!/bin/bash
# Timing of k1_scan for a few build variants (run on the GPU box): TILE, threads per workgroup, resident WGs per CU
for cfg in "2048 64 8" "2048 64 12" "2048 64 16" "1024 64 16" "4096 128 6"; do
  set -- $cfg
  MC_TILE=$1 MC_NTHREADS=$2 MC_SCAN_WGS=$3 python -c "from mcaller_amd.build import build_lib; build_lib(force=True, verbose=False)" 2>/dev/null || exit 1
  echo "== TILE=$1 NTHREADS=$2 WGS=$3"
  MC_TILE=$1 python tools/k1_experiments.py 1e8 | tail -1 | cut -c1-150
done
