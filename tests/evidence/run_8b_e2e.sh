cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python tests/evidence/ref_runtime_8b_q4k.py --cpu 32 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" | tee gpurun_out/e2e_8b_q4k.txt | tail -6
