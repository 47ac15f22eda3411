make -C 3dbodyanimation_amd/csrc stamps 2>&1 | grep -i "error" -A5
BODYFIT_LIB=3dbodyanimation_amd/libbodyfit_stamps.so timeout -k 10 300 python tools/stamp_priors.py 2>&1 | tail -25
