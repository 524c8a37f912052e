set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_prefill
rm -rf $O && mkdir -p $O
timeout -k 10 300 python probes/bench_prefill.py | tee $O/bench_prefill.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 probes/bench_prefill.py > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc -- python3 probes/bench_prefill.py > $O/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python3 probes/bench_prefill.py > $O/pmc2.log 2>&1
find $O -name "*_agent_info.csv" -delete
timeout -k 10 600 python probes/batched_bench.py | tee $O/batched_bench.txt
timeout -k 10 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_runner.py -m gpu -x -q 2>&1 | tail -3
