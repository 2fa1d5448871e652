cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl/p -- python3 bench.py --steps 64 --warmup 5 --kernels-only --time-every 16 > gpurun_out/tl/log 2>&1
python3 tools/trace_timeline.py gpurun_out/tl/p 400 > gpurun_out/tl/timeline.txt
rm -rf gpurun_out/tl/p
