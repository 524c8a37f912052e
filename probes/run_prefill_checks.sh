set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python probes/bench_prefill.py | tee gpurun_out/bench_prefill_lpt.txt
timeout -k 10 600 python -m pytest tests/test_gpu_prefill.py -m gpu -x -q 2>&1 | tail -3
