#!/bin/bash
# rocprofv3 kernel averages of the bench kernels, one pass at a time and pipelined: tools/kstats.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for mode in sync pipe; do
  extra=""; [ $mode = sync ] && extra="--no-pipeline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/$mode -- python3 bench.py --steps 50 --warmup 5 --kernels-only $extra "$@" > gpurun_out/$tag/$mode.log 2>&1
  find gpurun_out/$tag/$mode -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/${mode}_stats.csv \;
  rm -rf gpurun_out/$tag/$mode
done
python3 - <<P
import csv
for mode in ('sync', 'pipe'):
    print(mode)
    for r in csv.DictReader(open('gpurun_out/$tag/%s_stats.csv' % mode)):
        print('  %-44s %6s %9.1f' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44], r['Calls'], float(r['AverageNs']) / 1e3))
P
