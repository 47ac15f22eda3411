timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout -k 10 300 python bench.py --no-fit --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [ (k,round(v['avg_launch_ms']*1e3,2)) for k,v in d['roofline']['kernels'].items()])"; done
