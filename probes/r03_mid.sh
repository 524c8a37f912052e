#!/bin/bash
# mid-round check on the GPU box: the tests that failed last time, the bench line, the prompt pass through the plugin, the reference's harness on the ops this round touched
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_prefill.py tests/test_gpu_decode.py -q -m gpu --tb=line > $O/t4.log 2>&1; tail -4 $O/t4.log
timeout -k 10 900 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || tail -5 $O/bench_n1.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r03/bench_n1.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_us']['mean'], d['roofline']['frac'], d['config'].get('tiles_per_workgroup'))
print('e2e', d['e2e']['value'], d['e2e']['roofline']['frac'])
for k, v in d['configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('tiles_per_workgroup'), v.get('S_PP'), v.get('S_TG'), v.get('error'))
PY
timeout -k 10 600 python3 tests/evidence/ref_runtime_8b_q4k.py --pp 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" > $O/e2e_8b_q4k_plugin.txt; tail -4 $O/e2e_8b_q4k_plugin.txt
export GGML_BACKEND_PATH=$PWD/llama.cpp_test_amd/libggml-mi355.so
for op in MUL_MAT FLASH_ATTN_EXT; do
  timeout -k 10 600 oracle/_ref/test-backend-ops test -o $op 2>&1 | sed 's/\x1b\[[0-9;]*m//g' > $O/tbo_$op.txt
  echo "$op OK=$(grep -c '): OK' $O/tbo_$op.txt) FAIL=$(grep -c FAIL $O/tbo_$op.txt) not-supported=$(grep -c 'not supported' $O/tbo_$op.txt)"
done
