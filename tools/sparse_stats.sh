#!/bin/bash
# kernel averages of the headline (GATC) bench: pipelined, and one pass at a time through the pipelined interface.  tools/sparse_stats.sh <tag>
tag=${1:-sparse}; shift
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
for mode in pipelined depth1; do
  extra=""; [ $mode = depth1 ] && extra="--depth 1"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --kernels-only $extra "$@" > $out/bench_$mode.json 2> $out/err_$mode.log
  cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_$mode.csv; rm -rf $out/stats
  echo "== $mode"
  python3 - <<P
import csv, json
for row in csv.DictReader(open("$out/kernel_stats_$mode.csv")):
    name=row["Name"].replace("(anonymous namespace)::","")[:48]
    if float(row["AverageNs"]) > 3000 and int(row["Calls"]) > 20: print("%-50s %5s %9.1f us" % (name, row["Calls"], float(row["AverageNs"])/1e3))
try:
    d=json.load(open("$out/bench_$mode.json")); print(d["value"], d["ms_per_step"], d["ms_per_step_steady"], d["config"]["kernel_ms"])
except Exception as e: print("bench line:", e)
P
done
