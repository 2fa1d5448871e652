#!/bin/bash
# What every phase of k1_fused issues: SQ counters of the stop-after-phase variants (tools/variants.sh fd_stop1 "MC_FD_STOP=1" ... fd_stop5)
# and of the default build, one rocprofv3 --pmc pass per counter set and library (variants that are not built are skipped).
# tools/fused_phase_counters.sh [rows]   Output: gpurun_out/fdc/summary.txt
rows=${1:-1e8}
export TMPDIR=/tmp
rm -rf gpurun_out/fdc; mkdir -p gpurun_out/fdc
for v in fd_stop1 fd_stop2 fd_stop3 fd_stop4 fd_stop5 default; do
  if [ $v != default ] && [ ! -f mcaller_amd/variants/$v.so ]; then continue; fi
  if [ $v = default ]; then unset MCALLER_LIB; else export MCALLER_LIB=$PWD/mcaller_amd/variants/$v.so; fi
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/fdc/$v.p$i -- python3 tools/fused_probe.py --one $rows > gpurun_out/fdc/$v.p$i.log 2>&1
  done
done
python3 - <<'P'
import csv, glob, collections, os
rows = {}
for v in ('fd_stop1', 'fd_stop2', 'fd_stop3', 'fd_stop4', 'fd_stop5', 'default'):
    agg = collections.defaultdict(list)
    for path in glob.glob('gpurun_out/fdc/%s.p*/**/*counter_collection.csv' % v, recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(path)):
            if 'k1_fused<true>' not in row['Kernel_Name']: continue
            per[(row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
        for (d, c), val in per.items():
            agg[c].append(val)
    if agg: rows[v] = {c: sum(x) / len(x) for c, x in agg.items()}
names = sorted({c for r in rows.values() for c in r})
with open('gpurun_out/fdc/summary.txt', 'w') as out:
    out.write('%-10s ' % 'variant' + ' '.join('%20s' % c for c in names) + '\n')
    for v, r in rows.items():
        out.write('%-10s ' % v + ' '.join('%20.0f' % r.get(c, 0) for c in names) + '\n')
print(open('gpurun_out/fdc/summary.txt').read())
P
find gpurun_out/fdc -name "*kernel_trace.csv" -delete; find gpurun_out/fdc -name "*counter_collection.csv" -delete
