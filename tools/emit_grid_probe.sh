#!/bin/bash
# k1_emit's time by the number of its workgroups per CU (MCALLER_EMIT_WGS): time = fixed + per-round * rounds.  tools/emit_grid_probe.sh [events]
for w in 1 2 3 4 5 8; do
  echo "== EMIT_WGS $w"
  MCALLER_EMIT_WGS=$w tools/sparse_stats.sh eg$w --events ${1:-1e8} 2>&1 | grep "k1_emit" | tail -1
done
