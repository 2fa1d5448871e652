#!/bin/bash
# Round profile collection on the GPU box -> gpurun_out/<tag>/ (copy the summaries into profiles/ afterwards, see profiles/README.md).
# usage: tools/collect_profiles.sh <tag> <commit the snapshot was taken at> [quick]
tag=${1:-prof}
head=${2:-unknown}
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
src=$(python3 -c "import bench; print(bench.kernel_source_hash())")
echo "commit $head, kernel sources $src" > $out/HEAD.txt
# the PMC passes first: bench.py quotes roofline.traffic from profiles/r06_pmc.json if that file was collected on the kernel
# sources the library was built from -- on the box's copy of the repository it is, from here on
: > $out/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 4 --warmup 0 --kernels-only > /dev/null 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 4 --warmup 0 --kernels-only > /dev/null 2>> $out/bench.err
python3 tools/pmc_summary.py $out/fetch $out/write $out/pmc.json "python3 bench.py --steps 4 --warmup 0 --kernels-only" $head
cp $out/pmc.json profiles/r06_pmc.json
MCALLER_BENCH_DETAILS=$out/bench_details.json timeout 1200 python3 bench.py > $out/bench.json 2>> $out/bench.err
# the driver's own command line (20 steps: the pipeline's fill and drain inside the timed region; ms_per_step_steady beside it)
MCALLER_BENCH_DETAILS=$out/bench_driver_cmdline_details.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmdline.json 2>> $out/bench.err
# two ranks on the one GPU of this box (plumbing of the N > 1 line: both legs, the reduction's host-sum branch), at 2*10^7 rows
MCALLER_BENCH_DETAILS=$out/bench_2ranks_details.json MCALLER_BENCH_ONE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 --events 2e7 --strong-events 2e7 --f2f-big-events 2e7 --f2f-events 1e6 > $out/bench_2ranks_on_one_gpu_plumbing.json 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --kernels-only > $out/stats_bench.json 2>> $out/bench.err
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/fetch $out/write $out/stats
# ... the same with the side stream as the three kernels it was (k1_rare_dev, k2_mlp, k_pack): what the one kernel replaced, on this box
MCALLER_SIDE_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --kernels-only > $out/stats_bench_three_side_kernels.json 2>> $out/bench.err
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_three_side_kernels.csv; rm -rf $out/stats
# the rows of text written on the GPU (mc_rowtext.hip): its kernels alone (one shard, pass after pass, nothing else on the GPU), and the
# campaign of small files through the CLI as one table (the host formatter) and streamed with the device's row writer
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/statsrt -- python3 tools/rowtext_probe.py 1e6 20 A > $out/rowtext_probe.log 2>&1
cp $out/statsrt/*/*kernel_stats.csv $out/kernel_stats_rowtext_probe.csv; rm -rf $out/statsrt
( echo "commit $head, kernel sources $src"; timeout 900 python3 tests/tools/fuzz_rowtext.py ${FUZZ_ROWTEXT_CASES:-300} ${FUZZ_SEED:-31000000} ) > $out/fuzz_rowtext.log 2>&1
( echo "commit $head, kernel sources $src"; timeout 600 python3 tools/stream_soak.py 3e6 ${SOAK_RUNS_QUICK:-12} A; timeout 600 python3 tools/stream_soak.py 3e6 ${SOAK_RUNS_QUICK:-12} GATC ) > $out/stream_soak_rows_on_the_gpu.log 2>&1
if [ "$3" != "quick" ]; then
MCALLER_BENCH_DETAILS=$out/bench_sync_details.json timeout 600 python3 bench.py --no-pipeline --kernels-only > $out/bench_sync.json 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 bench.py --kernels-only --no-pipeline --steps 50 > /dev/null 2>> $out/bench.err
cp $out/stats1/*/*kernel_stats.csv $out/kernel_stats_one_pass_at_a_time.csv; rm -rf $out/stats1
MCALLER_BENCH_DETAILS=$out/bench_1e9_details.json timeout 600 python3 bench.py --events 1e9 --steps 10 --warmup 3 --kernels-only > $out/bench_1e9.json 2>> $out/bench.err
# dense (-m A): bench line, kernel averages one pass at a time, PMC passes
MCALLER_BENCH_DETAILS=$out/bench_dense_1e8_details.json timeout 600 python3 bench.py --motif A --events 1e8 --steps 5 --warmup 2 --kernels-only > $out/bench_dense_1e8.json 2>> $out/bench.err
# (pipelined passes, one in flight: K0 + k1_fused; the synchronous passes at the end of the same run: the scan + emit pair)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/statsd -- python3 bench.py --motif A --events 1e8 --steps 20 --warmup 3 --kernels-only --depth 1 > /dev/null 2>> $out/bench.err
cp $out/statsd/*/*kernel_stats.csv $out/kernel_stats_dense_one_pass_at_a_time.csv; rm -rf $out/statsd
MCALLER_SIDE_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/statsd -- python3 bench.py --motif A --events 1e8 --steps 20 --warmup 3 --kernels-only --depth 1 > /dev/null 2>> $out/bench.err
cp $out/statsd/*/*kernel_stats.csv $out/kernel_stats_dense_one_pass_at_a_time_three_side_kernels.csv; rm -rf $out/statsd
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/statsd -- python3 bench.py --motif A --events 1e8 --steps 20 --warmup 2 --kernels-only --no-pipeline > /dev/null 2>> $out/bench.err
cp $out/statsd/*/*kernel_stats.csv $out/kernel_stats_dense_synchronous_pair.csv; rm -rf $out/statsd
# dense file to file and config 5, with their kernel averages
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -- python3 tools/file_to_file.py 1e7 --motif A > $out/file_to_file_dense.log 2>&1
cp $out/stats3/*/*kernel_stats.csv $out/kernel_stats_file_to_file_dense.csv; rm -rf $out/stats3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4 -- python3 tools/config5.py 1e7 --runs 3 > $out/config5.log 2>&1
cp $out/stats4/*/*kernel_stats.csv $out/kernel_stats_config5.csv; rm -rf $out/stats4
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetchd -- python3 bench.py --motif A --events 1e8 --steps 3 --warmup 0 --kernels-only > /dev/null 2>> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/writed -- python3 bench.py --motif A --events 1e8 --steps 3 --warmup 0 --kernels-only > /dev/null 2>> $out/bench.err
python3 tools/pmc_summary.py $out/fetchd $out/writed $out/pmc_dense.json "python3 bench.py --motif A --events 1e8 --steps 3 --warmup 0 --kernels-only" $head
rm -rf $out/fetchd $out/writed
# the streamed file-to-file path: kernel averages of the device parser and of the per-shard passes
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats2 -- python3 tools/file_to_file.py 1e7 > $out/file_to_file.log 2>&1
cp $out/stats2/*/*kernel_stats.csv $out/kernel_stats_file_to_file.csv; rm -rf $out/stats2
# evidence on THIS code: fuzz campaign, pipelined soak, streamed-CLI soak (logs carry the commit and the kernel-source hash)
( echo "commit $head, kernel sources $src"; timeout 1500 python3 tests/tools/fuzz_gpu.py ${FUZZ_PER_FLAVOUR:-600} ${FUZZ_SEED:-31000000} ) > $out/fuzz.log 2>&1
( echo "commit $head, kernel sources $src"; timeout 900 python3 tests/tools/fuzz_tables.py ${FUZZ_DENSE_TABLES:-2000} ${FUZZ_SEED:-31000000} dense; timeout 900 python3 tests/tools/fuzz_tables.py ${FUZZ_SPARSE_TABLES:-2000} ${FUZZ_SEED:-31000000} sparse ) > $out/fuzz_tables.log 2>&1
( echo "commit $head, kernel sources $src"; timeout 600 python3 tools/pipeline_soak.py 1e8 ${SOAK_PASSES:-1500} ) > $out/pipeline_soak.log 2>&1
( echo "commit $head, kernel sources $src"; timeout 900 python3 tools/stream_soak.py 3e6 ${SOAK_RUNS:-30} ) > $out/stream_soak.log 2>&1
fi
python3 - <<P
import json
for f in ("bench","bench_driver_cmdline","bench_sync","bench_1e9","bench_dense_1e8"):
    try:
        d=json.load(open("$out/%s.json"%f)); r=d["roofline"]
        print(f, "%.4g"%d["value"], "ms/step %.4f steady %s fp64 %s"%(d["ms_per_step"], d.get("ms_per_step_steady"), d.get("ms_per_step_fp64_mlp")), "frac %.4f"%r["frac"], r.get("kernels_ms"), "traffic", r.get("traffic"),
              "f2f", d.get("file_to_file_calls_per_s"), (d.get("file_to_file_dense") or {}).get("s_per_1e8"), (d.get("strong_scaling") or {}).get("seconds_median"), (d.get("strong_scaling") or {}).get("projected_s"), "len", len(open("$out/%s.json"%f).read()))
    except Exception as e: print(f, e)
P
tail -n 3 $out/bench.err
tail -n 2 $out/fuzz.log $out/fuzz_tables.log $out/pipeline_soak.log $out/stream_soak.log 2>/dev/null
grep -v "^[EW]2026" $out/rowtext_probe.log | tail -n 2; grep "k_rt_" $out/kernel_stats_rowtext_probe.csv | cut -c1-140
tail -n 1 $out/fuzz_rowtext.log; tail -n 2 $out/stream_soak_rows_on_the_gpu.log
