timeout -k 10 900 python -m pytest tests/test_gpu_sharded_solve.py -x -q 2>&1 | tail -12
