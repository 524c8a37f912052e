#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_x8.log 2>&1; rc=$?
tail -3 $O/tests_x8.log; [ $rc -ne 0 ] && { grep -E "^E  " $O/tests_x8.log | head; exit $rc; }
timeout -k 10 300 python3 probes/ab_decode.py --b probes/bin/libv_nopreload.so probes/bin/libmi355_paged_kv_r01.so > $O/ab_x8.txt 2>&1; grep -v amdgpu.ids $O/ab_x8.txt | tail -4
timeout -k 10 300 python3 probes/ab_decode.py --model 70b --ctx 16384 --b probes/bin/libv_nopreload.so probes/bin/libmi355_paged_kv_r01.so > $O/ab_x8_70b.txt 2>&1; grep -v amdgpu.ids $O/ab_x8_70b.txt | tail -4
echo "e2e:"; timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 2>&1 | tail -1
timeout -k 10 300 python3 probes/gemm_bench.py > $O/gemm_bench_x8.txt 2>&1; grep -v amdgpu.ids $O/gemm_bench_x8.txt | grep "q4_K\|f16"
echo "plugin e2e:"; timeout -k 10 400 python3 tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep "tok/s"
