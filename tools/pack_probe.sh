# The pipelined sparse step under other geometries of the packing kernel (variants pk*: tools/variants.sh pk128 "MC_PACK_WGS=128 MC_PACK_THREADS=256" ...): bash tools/pack_probe.sh
for v in default pk128 pk256 pk64; do
  if [ $v != default ] && [ ! -f mcaller_amd/variants/$v.so ]; then continue; fi
  if [ $v = default ]; then unset MCALLER_LIB; else export MCALLER_LIB=$PWD/mcaller_amd/variants/$v.so; fi
  for rep in 1 2; do
  python3 bench.py --kernels-only ${@:---steps 200} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g'%d['value'], '%.4f'%d['ms_per_step'], '%.4f'%d['ms_per_step_steady'])"
  done
done
