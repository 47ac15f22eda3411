timeout -k 10 300 python tools/tmp/chk.py || exit 1
timeout -k 10 800 python -m pytest tests/test_gpu_fit.py tests/test_gpu_window_lm.py tests/test_gpu_sharded_solve.py -x -q 2>&1 | tail -15
timeout -k 10 200 python tools/fit_bench.py 2>&1 | tail -3
timeout -k 10 200 python tools/window_bench.py 2>&1 | tail -8
