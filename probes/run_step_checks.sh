set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode.py -m gpu -x -q > gpurun_out/step_tests.log 2>&1 || { tail -40 gpurun_out/step_tests.log; exit 1; }
tail -3 gpurun_out/step_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_step.json
cat gpurun_out/bench_step.json
timeout -k 10 300 python bench.py --no-cpu-baseline --fused-merge > gpurun_out/bench_step_fm.json
cat gpurun_out/bench_step_fm.json
timeout -k 10 300 python bench.py --no-cpu-baseline --fused-merge --kv-type q8_0 > gpurun_out/bench_step_fm_q8.json
cat gpurun_out/bench_step_fm_q8.json
timeout -k 10 300 python bench.py --no-cpu-baseline --n-seq 32 --fused-merge > gpurun_out/bench_step_fm_n32.json
cat gpurun_out/bench_step_fm_n32.json
