#!/bin/bash
# kernel averages of the dense (-m A) bench, one pass at a time through the pipelined interface (the fused pass) and through the
# synchronous one (the scan + emit pair): tools/dense_stats.sh <tag> [bench args]
tag=${1:-dense}; shift
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
for mode in depth1 sync; do
  extra="--depth 1"; [ $mode = sync ] && extra="--no-pipeline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --kernels-only --motif A --steps 20 --warmup 3 $extra "$@" > $out/bench_$mode.json 2> $out/err_$mode.log
  cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_$mode.csv; rm -rf $out/stats
  echo "== $mode"
  python3 - <<P
import csv
for row in csv.DictReader(open("$out/kernel_stats_$mode.csv")):
    name=row["Name"].replace("(anonymous namespace)::","")[:52]
    if float(row["AverageNs"]) > 3000 and int(row["Calls"]) > 5: print("%-54s %5s %9.1f us" % (name, row["Calls"], float(row["AverageNs"])/1e3))
P
done
