#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_gemv.py tests/test_gpu_decoder.py tests/test_gpu_layer_ops.py -m gpu -x -q > $O/tests_x7.log 2>&1; rc=$?
tail -3 $O/tests_x7.log; [ $rc -ne 0 ] && { grep -E "^E  " $O/tests_x7.log | head; exit $rc; }
timeout -k 10 400 python3 probes/gemv_shapes.py > $O/gemv_shapes_x7.txt 2>&1 || { tail -5 $O/gemv_shapes_x7.txt; exit 1; }
grep -v amdgpu.ids $O/gemv_shapes_x7.txt
echo "e2e:"; timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 2>&1 | tail -1
echo "e2e fused merge:"; timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 --fused-merge 2>&1 | tail -1
