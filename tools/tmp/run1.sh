timeout -k 10 600 python -m pytest tests/test_gpu_cpp_api.py -x -q 2>&1 | tail -25
