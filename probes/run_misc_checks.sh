cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_decode_step.py -m gpu -x -q -k "frequency" 2>&1 | tail -3
PKV_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 2>&1 | tail -3
