cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_layer_ops.py tests/test_gpu_reference_runtime.py -m gpu -x -q 2>&1 | tail -4
timeout -k 10 1000 python tests/evidence/ref_cpu_tinyllama.py 32 --plugin 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" | tail -3
timeout -k 10 1000 python tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" | tail -1
