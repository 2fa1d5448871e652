cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06f; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -- python3 tools/file_to_file.py 1e7 --motif A > $out/file_to_file_dense.log 2>&1
cp $out/stats3/*/*kernel_stats.csv $out/kernel_stats_file_to_file_dense.csv; rm -rf $out/stats3
grep "^run\|inputs" $out/file_to_file_dense.log
( echo "commit 03f33d3"; bash tools/check_tools.sh ) > $out/tools_check.log 2>&1
tail -50 $out/tools_check.log
