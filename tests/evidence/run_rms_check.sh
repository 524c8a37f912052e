# layer-op parity (our tests + the reference harness) and the per-kernel breakdown of the 8B Q4_K end-to-end run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_layer_ops.py tests/test_gpu_reference_runtime.py tests/test_gpu_reference_harness.py -m gpu -x -q -k "layer_ops or runtime or RMS_NORM or MUL or SILU" 2>&1 | tail -4 &&
bash tests/evidence/profile_8b.sh &&
timeout -k 10 600 python tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep "tok/s"
