cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_us'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --n-seq 32 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_us'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --two-call | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_us'])"
timeout -k 10 300 python -m pytest tests/test_gpu_decode_step.py -m gpu -x -q 2>&1 | tail -2
