timeout -k 10 900 python -m pytest tests/test_gpu_window_lm.py tests/test_gpu_sharded_solve.py tests/test_gpu_fit.py -x -q 2>&1 | tail -4
timeout -k 10 200 python tools/fit_bench.py 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v['seconds'],5), round(v['frames_per_s'])) for k,v in d.items()})"
timeout -k 10 300 python bench.py --workload c5 --fit 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['fit'])"
