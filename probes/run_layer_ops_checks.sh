cd $GRAFT_REPO_ROOT
export GGML_BACKEND_PATH=$PWD/llama.cpp_test_amd/libggml-mi355.so
for op in MUL_MAT; do
  timeout -k 10 300 oracle/_ref/test-backend-ops test -o $op > gpurun_out/tbo_$op.txt 2>&1
  echo "$op rc=$? OK=$(grep -c ': .*OK' gpurun_out/tbo_$op.txt) FAIL=$(grep -c 'FAIL' gpurun_out/tbo_$op.txt) notsup=$(grep -c 'not supported' gpurun_out/tbo_$op.txt)"
  grep FAIL gpurun_out/tbo_$op.txt | head -5
done
unset GGML_BACKEND_PATH
timeout -k 10 600 python -m pytest tests/test_gpu_reference_runtime.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 1000 python probes/ref_cpu_tinyllama.py 32 --plugin 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" | tee gpurun_out/tinyllama_e2e.txt | tail -4
