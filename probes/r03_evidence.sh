#!/bin/bash
# End-of-round evidence (r03 form of probes/r02_evidence.sh) on the GPU box (one call): the whole -m gpu suite, smoke(), bench.py (with cpu_baseline and e2e), rocprofv3 kernel statistics of
# the bench command and of the end-to-end step, PMC traffic passes (separate, as MI355X_MICROARCH.md prescribes), in-kernel stamps, A/B against the
# round-1 library, the secondary configurations quoted in DESIGN.md.  Summarise afterwards with probes/r03_summarize.py (dev container).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_final; mkdir -p $O
# PART=a: tests, smoke, bench, rocprof / PMC passes, stamps.  PART=b: secondary configurations, product / prefill probes, the reference runtime and harness.
# (one gpurun call is limited to 20 minutes; the two halves write into the same directory, merged back by gpurun)
PART=${PART:-ab}
step() { echo "== $1" | tee -a $O/steps.log; }
if [[ $PART == *a* ]]; then
step "gpu tests"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
step "smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
step "bench"
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'], d['e2e']['value'], d['e2e']['roofline']['frac'], d['cpu_baseline']['value'], d['e2e'].get('cpu_baseline')); [print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('tiles_per_workgroup'), v.get('S_PP'), v.get('S_PP_f16_io'), v.get('S_TG'), v.get('error')) for k, v in d['configs'].items()]"
step "rocprof kernel stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
python3 probes/timeline.py $O/stats | tee $O/timeline.txt
step "pmc passes"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs --steps 20 --warmup 2 --kernel-reps 2 > $O/pmc_fetch.log 2>&1 || echo "pmc fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs --steps 20 --warmup 2 --kernel-reps 2 > $O/pmc_write.log 2>&1 || echo "pmc write pass failed"
step "rocprof kernel stats of the end-to-end step"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_stats -- python3 probes/e2e_profile.py --steps 10 > $O/e2e_stats.log 2>&1 || { tail -5 $O/e2e_stats.log; exit 1; }
tail -1 $O/e2e_stats.log
step "stamps"
timeout -k 10 300 python3 probes/decode_stamps.py --mode 2 > $O/stamps_8b.txt 2>&1 || tail -3 $O/stamps_8b.txt
timeout -k 10 300 python3 probes/decode_stamps.py --mode 2 --model 70b --ctx 16384 > $O/stamps_70b.txt 2>&1 || tail -3 $O/stamps_70b.txt
fi
if [[ $PART == *b* ]]; then
step "secondary configurations"
for cfg in "nseq32:--n-seq 32" "nseq32_ctx640:--n-seq 32 --ctx 640" "q8_0:--kv-type q8_0" "q4_0:--kv-type q4_0" "q8_0_nseq32:--n-seq 32 --kv-type q8_0" "q4_0_nseq32:--n-seq 32 --kv-type q4_0" "70b_ctx16384:--model 70b --ctx 16384" "two_call:--two-call" "fused_merge:--fused-merge" "one_tile_nseq32:--n-seq 32" "one_tile_70b:--model 70b --ctx 16384"; do
  n=${cfg%%:*}; f=${cfg#*:}
  export PKV_DECODE_NTILE=0; case $n in one_tile_*) export PKV_DECODE_NTILE=1;; esac
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --no-configs $f > $O/bench_$n.json 2> $O/bench_$n.err || { echo "bench $n failed"; tail -3 $O/bench_$n.err; continue; }
  python3 -c "import json; d=json.load(open('$O/bench_$n.json')); print('$n', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['kernel_us']['mean'], 'us', d['roofline']['frac'])"
done
step "gemv shapes / gemm / batched bench / prefill"
unset PKV_DECODE_NTILE
timeout -k 10 400 python3 probes/gemv_shapes.py --packed > $O/gemv_shapes.txt 2>&1; grep -v amdgpu.ids $O/gemv_shapes.txt
timeout -k 10 300 python3 probes/gemv_shapes.py --q5 --packed > $O/gemv_shapes_q5.txt 2>&1; grep -v amdgpu.ids $O/gemv_shapes_q5.txt
{ timeout -k 10 300 python3 probes/e2e_profile.py --wtype q5_K --steps 30; timeout -k 10 300 python3 probes/e2e_profile.py --wtype q5_K --steps 30 --no-pack; } 2>&1 | grep -v amdgpu.ids > $O/e2e_q5km.txt; cat $O/e2e_q5km.txt
timeout -k 10 300 python3 probes/gemm_bench.py > $O/gemm_bench.txt 2>&1; grep -v amdgpu.ids $O/gemm_bench.txt
timeout -k 10 300 python3 probes/batched_bench.py > $O/batched_bench.txt 2>&1; timeout -k 10 300 python3 probes/batched_bench.py --pp-f16 >> $O/batched_bench.txt 2>&1; timeout -k 10 300 python3 probes/batched_bench.py --kv-type q8_0 >> $O/batched_bench.txt 2>&1; grep -v amdgpu.ids $O/batched_bench.txt
timeout -k 10 300 python3 probes/bench_prefill.py > $O/bench_prefill.txt 2>&1; tail -6 $O/bench_prefill.txt
step "prefill: PMC passes, timing ablations, phase stamps"
O_DIR=$O/pmc_prefill bash probes/r02_pmc_prefill.sh r03_final/pmc_prefill > $O/prefill_pmc.txt 2>&1; tail -2 $O/prefill_pmc.txt
step "reference runtime end to end through the plugin"
timeout -k 10 600 python3 tests/evidence/ref_runtime_8b_q4k.py --pp 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" > $O/e2e_8b_q4k_plugin.txt; tail -4 $O/e2e_8b_q4k_plugin.txt
timeout -k 10 600 python3 tests/evidence/ref_cpu_tinyllama.py 32 --plugin 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" > $O/tinyllama_e2e_plugin.txt; tail -3 $O/tinyllama_e2e_plugin.txt
step "the reference's harness"
export GGML_BACKEND_PATH=$PWD/llama.cpp_test_amd/libggml-mi355.so
for op in FLASH_ATTN_EXT ROPE ROPE_BACK CPY CONT RMS_NORM ADD MUL SILU MUL_MAT SOFT_MAX; do
  timeout -k 10 600 oracle/_ref/test-backend-ops test -o $op 2>&1 | sed 's/\x1b\[[0-9;]*m//g' > $O/tbo_$op.txt
  echo "$op OK=$(grep -c '): OK' $O/tbo_$op.txt) FAIL=$(grep -c FAIL $O/tbo_$op.txt) not-supported=$(grep -c 'not supported' $O/tbo_$op.txt)"
done
unset GGML_BACKEND_PATH
fi
find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
grep -l "Memory access fault" $O/*.log $O/*.txt $O/*.err 2>/dev/null && { echo "GPU FAULT in the files above"; exit 1; }
du -sh $O
