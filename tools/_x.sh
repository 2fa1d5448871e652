timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q 2>&1 | grep -E "passed|failed|rror|Abort" | tail -2
timeout 200 python tests/tools/fuzz_gpu.py 150 19000000 2>&1 | tail -4
for v in default k2second default k2second; do
  [ $v = default ] && unset MCALLER_LIB || export MCALLER_LIB=$GRAFT_REPO_ROOT/mcaller_amd/variants/$v.so
  echo "== $v $(timeout 100 python3 bench.py --kernels-only 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['kernel_ms'])")"
done
