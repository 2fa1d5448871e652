#!/bin/bash
# kernel averages of the dense (-m A) pipelined path, one pass at a time (no overlap between passes): tools/dense_stats.sh <tag> [extra bench args]
tag=${1:-dense}; shift
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --kernels-only --motif A --steps 6 --warmup 2 --depth 1 "$@" > $out/bench_dense.json 2> $out/err.log
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_dense.csv; rm -rf $out/stats
python3 - <<P
import csv, json
for row in csv.DictReader(open("$out/kernel_stats_dense.csv")):
    name=row["Name"].replace("(anonymous namespace)::","")[:60]
    print("%-62s %5s %10.1f us" % (name, row["Calls"], float(row["AverageNs"])/1e3))
try:
    d=json.load(open("$out/bench_dense.json")); print(d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
except Exception as e: print("bench line:", e)
P
tail -3 $out/err.log
