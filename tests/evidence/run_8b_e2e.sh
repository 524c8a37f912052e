cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_layer_ops.py -m gpu -x -q 2>&1 | tail -3
GGML_BACKEND_PATH=$PWD/llama.cpp_test_amd/libggml-mi355.so timeout -k 10 300 oracle/_ref/test-backend-ops test -o MUL_MAT 2>&1 | sed 's/\x1b\[[0-9;]*m//g' > gpurun_out/tbo_MUL_MAT.txt; echo "MUL_MAT OK=$(grep -c ': OK' gpurun_out/tbo_MUL_MAT.txt) FAIL=$(grep -c FAIL gpurun_out/tbo_MUL_MAT.txt) q4_K_OK=$(grep 'type_a=q4_K' gpurun_out/tbo_MUL_MAT.txt | grep -c ': OK')"
timeout -k 10 1000 python tests/evidence/ref_runtime_8b_q4k.py --cpu 32 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" | tee gpurun_out/e2e_8b_q4k.txt | tail -3
