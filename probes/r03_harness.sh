#!/bin/bash
# the reference's own harness (oracle/_ref/test-backend-ops, built from /root/reference's sources in the dev container) through the plugin, for the listed ops
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O
export GGML_BACKEND_PATH=$PWD/llama.cpp_test_amd/libggml-mi355.so
for op in "$@"; do
  timeout -k 10 600 oracle/_ref/test-backend-ops test -o $op 2>&1 | sed 's/\x1b\[[0-9;]*m//g' > $O/tbo_$op.txt
  echo "$op OK=$(grep -c '): OK' $O/tbo_$op.txt) FAIL=$(grep -c FAIL $O/tbo_$op.txt) not-supported=$(grep -c 'not supported' $O/tbo_$op.txt)"
done
exit 0
