#!/bin/bash
# prefill kernel iteration: parity tests, throughput probe, dispatch-order A/B, phase stamps
set -o pipefail
O=gpurun_out/${1:-p1}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_prefill.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python probes/bench_prefill.py > $O/bench.txt 2>&1 && cat $O/bench.txt || exit 1
for s in 0 1 2; do echo "stagger $s"; PKV_PREFILL_STAGGER=$s PKV_PREFILL_CASE=0 timeout -k 10 120 python probes/bench_prefill.py 2>&1 | grep n_seq; done | tee $O/stagger.txt
timeout -k 10 120 python probes/stamp_prefill.py > $O/stamps.txt 2>&1 && cat $O/stamps.txt
