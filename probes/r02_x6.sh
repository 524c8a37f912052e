#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_reference_runtime.py tests/test_gpu_decode.py tests/test_gpu_decode_step.py -m gpu -x -q > $O/tests_x6.log 2>&1; rc=$?
tail -4 $O/tests_x6.log; grep -E "^E  |ggml-mi355:" $O/tests_x6.log | head -20
[ $rc -ne 0 ] && exit $rc
for l in "" probes/bin/libq_q64.so; do
  PKV_LIB_PATH=${l:+$PWD/$l} python3 bench.py --no-cpu-baseline --no-e2e --n-seq 32 --kv-type q4_0 > $O/b_q4_x6.json && python3 -c "import json; d=json.load(open('$O/b_q4_x6.json')); print('q4_0 n32 [$l]', d['value'], d['roofline']['kernel_us']['mean'], d['roofline']['frac'])"
done
PKV_LIB_PATH=$PWD/probes/bin/libq_q64.so python -m pytest tests/test_gpu_decode.py tests/test_gpu_configs.py -m gpu -x -q -k "many or config3" 2>&1 | tail -2
