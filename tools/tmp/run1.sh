timeout -k 10 900 python -m pytest tests/test_gpu_window_lm.py tests/test_gpu_sharded_solve.py -x -q 2>&1 | tail -3
timeout -k 10 300 python bench.py --workload c5 --fit 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['fit']['ms_per_iteration'])"
