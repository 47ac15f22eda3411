timeout -k 10 900 python -m pytest tests/test_gpu_window_lm.py tests/test_gpu_sharded_solve.py -x -q 2>&1 | tail -5
timeout -k 10 200 python tools/fit_bench.py 2>&1 | tail -1
