#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_x4.log 2>&1; rc=$?
tail -4 $O/tests_x4.log
[ $rc -ne 0 ] && { grep -E "^E  " $O/tests_x4.log | head -12; exit $rc; }
timeout -k 10 300 python3 probes/gemm_bench.py > $O/gemm_bench.txt 2>&1 || { tail -5 $O/gemm_bench.txt; exit 1; }
grep -v amdgpu.ids $O/gemm_bench.txt
timeout -k 10 400 python3 probes/gemv_shapes.py > $O/gemv_shapes_x4.txt 2>&1 || { tail -5 $O/gemv_shapes_x4.txt; exit 1; }
grep -v amdgpu.ids $O/gemv_shapes_x4.txt
echo "e2e:"; timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 2>&1 | tail -1
