#!/bin/bash
# SQ-side counters of the bench kernels (separate rocprofv3 passes; --kernel-trace only).  Output: gpurun_out/sq/*.csv
# usage: tools/sq_counters.sh [extra bench.py arguments, e.g. "--motif A"]
extra="$1"
export TMPDIR=/tmp
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/sq/p$i -- python3 bench.py --steps 2 --warmup 1 --kernels-only $extra > gpurun_out/sq/p$i.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/sq/p*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace('k1_dense<true, false>', 'k1_dense_emit').replace('k1_dense<false, false>', 'k1_dense_count').replace('k1_dense<false, true>', 'k1_dense_countv').split('<')[0]
        per[(k, row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
    for (k, d, c), v in per.items():
        agg[k][c].append(v)
with open('gpurun_out/sq/summary.txt', 'w') as out:
    for k in sorted(agg):
        if not k.startswith('k'): continue
        out.write(k + '\n')
        for c in sorted(agg[k]):
            vs = agg[k][c]
            out.write('   %-24s %14.0f  (n=%d)\n' % (c, sum(vs) / len(vs), len(vs)))
print(open('gpurun_out/sq/summary.txt').read())
P
find gpurun_out/sq -name "*kernel_trace.csv" -delete
