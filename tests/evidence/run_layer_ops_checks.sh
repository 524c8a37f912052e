cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_layer_ops.py tests/test_gpu_reference_harness.py tests/test_gpu_reference_runtime.py -m gpu -x -q 2>&1 | tail -12
