for t in ${TILES:-2048 2560 3072}; do
  MC_TILE=$t python -m mcaller_amd.build --force 2>/dev/null
  K1_DEBUGS=${K1_DEBUGS:-0,3} MC_TILE=$t MCALLER_VERBOSE=1 python tools/k1_experiments.py 1e8 2>&1 | grep -E "occupancy|debug=" | cut -c1-150
done
