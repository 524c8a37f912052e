set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode.py tests/test_gpu_runner.py -m gpu -x -q > gpurun_out/step_tests.log 2>&1 || { tail -40 gpurun_out/step_tests.log; exit 1; }
tail -3 gpurun_out/step_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_step.json
cat gpurun_out/bench_step.json
timeout -k 10 300 python bench.py --no-cpu-baseline --two-call > gpurun_out/bench_twocall.json
cat gpurun_out/bench_twocall.json
timeout -k 10 300 python bench.py --no-cpu-baseline --kv-type q8_0 > gpurun_out/bench_step_q8.json
cat gpurun_out/bench_step_q8.json
timeout -k 10 300 python bench.py --no-cpu-baseline --n-seq 32 > gpurun_out/bench_step_n32.json
cat gpurun_out/bench_step_n32.json
