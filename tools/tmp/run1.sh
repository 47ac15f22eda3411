timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py -x -q 2>&1 | tail -3
timeout -k 10 200 python tools/fit_bench.py 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v['seconds'],5), round(v['frames_per_s'])) for k,v in d.items()})"
BODYFIT_LIB=3dbodyanimation_amd/libbodyfit_stamps.so timeout -k 10 300 python tools/stamp_priors.py 2>&1 | tail -2
