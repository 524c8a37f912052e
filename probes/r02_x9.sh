#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_reference_runtime.py tests/test_gpu_reference_harness.py -m gpu -x -q > $O/tests_x9.log 2>&1; rc=$?
tail -3 $O/tests_x9.log; [ $rc -ne 0 ] && { grep -E "^E  |ggml-mi355" $O/tests_x9.log | head -20; exit $rc; }
echo "plugin e2e:"; timeout -k 10 400 python3 tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep "tok/s"
echo "plugin e2e, NO_FUSE:"; GGML_MI355_NO_FUSE=1 timeout -k 10 400 python3 tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep "tok/s"
echo "tinyllama:"; timeout -k 10 400 python3 tests/evidence/ref_cpu_tinyllama.py 32 --plugin 2>&1 | grep "tok/s\|NMSE"
