#!/bin/bash
# End-of-round evidence on the GPU box: the whole -m gpu suite, smoke(), the profile set (probes/profile_round.sh) and the
# secondary bench configurations quoted in DESIGN.md §5.  Summarise afterwards with probes/summarize_profile.py.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 && tail -2 gpurun_out/smoke.log
bash probes/profile_round.sh
O=gpurun_out/prof_round
python3 bench.py --no-cpu-baseline --n-seq 32 > $O/bench_nseq32.json
python3 bench.py --no-cpu-baseline --n-seq 32 --ctx 640 > $O/bench_nseq32_ctx640.json
python3 bench.py --no-cpu-baseline --kv-type q8_0 > $O/bench_q8_0.json
python3 bench.py --no-cpu-baseline --kv-type q4_0 > $O/bench_q4_0.json
python3 bench.py --no-cpu-baseline --model 70b --ctx 16384 > $O/bench_70b_ctx16384.json
python3 bench.py --no-cpu-baseline --two-call > $O/bench_two_call.json
python3 probes/bench_prefill.py > $O/bench_prefill.txt
python3 probes/batched_bench.py > $O/batched_bench.txt
for f in $O/bench_nseq32.json $O/bench_nseq32_ctx640.json $O/bench_q8_0.json $O/bench_q4_0.json $O/bench_70b_ctx16384.json $O/bench_two_call.json; do
  python3 -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['kernel_us']['mean'], 'us', d['roofline']['frac'])"
done
PKV_PREFILL_GEOM=70b python3 probes/bench_prefill.py > $O/bench_prefill_70b.txt
python3 tests/evidence/ref_cpu_tinyllama.py 32 --plugin 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" > $O/tinyllama_e2e.txt
python3 tests/evidence/ref_runtime_8b_q4k.py --cpu 32 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" > $O/e2e_8b_q4k.txt
python3 tests/evidence/ref_runtime_ctx4096.py 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" > $O/ref_runtime_ctx4096.txt
export GGML_BACKEND_PATH=$PWD/llama.cpp_test_amd/libggml-mi355.so
for op in FLASH_ATTN_EXT ROPE ROPE_BACK CPY RMS_NORM ADD MUL SILU MUL_MAT; do
  oracle/_ref/test-backend-ops test -o $op 2>&1 | sed 's/\x1b\[[0-9;]*m//g' > $O/tbo_$op.txt
  echo "$op OK=$(grep -c ': OK' $O/tbo_$op.txt) FAIL=$(grep -c FAIL $O/tbo_$op.txt)"
done
unset GGML_BACKEND_PATH
cat $O/bench_prefill.txt $O/batched_bench.txt $O/tinyllama_e2e.txt $O/e2e_8b_q4k.txt $O/ref_runtime_ctx4096.txt
