set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
python bench.py --no-cpu-baseline --fused-merge > gpurun_out/bench_fusedmerge.json 2>> gpurun_out/bench_default.err
cat gpurun_out/bench_fusedmerge.json
rm -rf gpurun_out/trace1 && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace1 -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/trace1.log 2>&1
python probes/timeline.py gpurun_out/trace1 | tee gpurun_out/timeline1.txt
rm -rf gpurun_out/trace1
