timeout -k 10 800 python -m pytest tests/test_gpu_fit.py -x -q 2>&1 | tail -8
timeout -k 10 200 python tools/fit_bench.py 2>&1 | tail -1
make -C 3dbodyanimation_amd/csrc stamps 2>&1 | grep -i "error" -A5
BODYFIT_LIB=3dbodyanimation_amd/libbodyfit_stamps.so timeout -k 10 300 python tools/stamp_lm.py 2>&1 | tail -11
