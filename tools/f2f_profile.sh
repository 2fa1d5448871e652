#!/bin/bash
# kernel averages of a file-to-file run (CLI, streaming): tools/f2f_profile.sh <rows>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/f2f; mkdir -p gpurun_out/f2f
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/f2f/p -- python3 tools/file_to_file.py ${1:-1e7} > gpurun_out/f2f/log.txt 2>&1
find gpurun_out/f2f/p -name "*kernel_stats.csv" -exec cp {} gpurun_out/f2f/kernel_stats.csv \;
python3 - <<P
import csv
for r in csv.DictReader(open('gpurun_out/f2f/kernel_stats.csv')):
    print('  %-44s %6s %9.1f us  total %8.2f ms' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
P
python3 tools/trace_timeline.py gpurun_out/f2f/p 140 | head -100 > gpurun_out/f2f/timeline.txt
rm -rf gpurun_out/f2f/p
grep -E "run|timing" gpurun_out/f2f/log.txt | tail -4
