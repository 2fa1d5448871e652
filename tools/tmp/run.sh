timeout 900 python -m pytest tests/test_gpu_devparse.py tests/test_gpu_stream.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
tools/f2f_profile.sh 1e7 2>&1 | grep "kp_\|run "
python bench.py --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline_parser'], d['config'].get('file_to_file_1e8_s'))"
