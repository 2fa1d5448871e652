#!/bin/bash
# Round profile collection on the GPU box -> gpurun_out/<tag>/ (copy the summaries into profiles/ afterwards, see profiles/README.md).
# usage: tools/collect_profiles.sh <tag> <commit the snapshot was taken at> [quick]
tag=${1:-prof}
head=${2:-unknown}
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --kernels-only > $out/stats_bench.json 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 4 --warmup 0 --kernels-only > /dev/null 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 4 --warmup 0 --kernels-only > /dev/null 2>> $out/bench.err
python3 tools/pmc_summary.py $out/fetch $out/write $out/pmc.json "python3 bench.py --steps 4 --warmup 0 --kernels-only" $head
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/fetch $out/write $out/stats
if [ "$3" != "quick" ]; then
timeout 600 python3 bench.py --no-pipeline --kernels-only > $out/bench_sync.json 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 bench.py --kernels-only --no-pipeline --steps 50 > /dev/null 2>> $out/bench.err
cp $out/stats1/*/*kernel_stats.csv $out/kernel_stats_one_pass_at_a_time.csv; rm -rf $out/stats1
timeout 600 python3 bench.py --events 1e9 --steps 10 --warmup 3 --kernels-only > $out/bench_1e9.json 2>> $out/bench.err
timeout 600 python3 bench.py --motif A --events 1e8 --steps 5 --warmup 2 --kernels-only > $out/bench_dense_1e8.json 2>> $out/bench.err
# the streamed file-to-file path: kernel averages of the device parser and of the per-shard passes
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats2 -- python3 tools/file_to_file.py 1e7 > $out/file_to_file.log 2>&1
cp $out/stats2/*/*kernel_stats.csv $out/kernel_stats_file_to_file.csv; rm -rf $out/stats2
fi
python3 - <<P
import json
for f in ("bench","bench_sync","bench_1e9","bench_dense_1e8"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d["config"]["kernel_ms"].items()}, d["roofline"]["frac"], d["roofline"]["per_table"]["frac"], (d["config"].get("device_e2e") or {}).get("events_per_s"), (d["config"].get("file_to_file") or {}).get("seconds_best"))
    except Exception as e: print(f, e)
P
tail -n 3 $out/bench.err
