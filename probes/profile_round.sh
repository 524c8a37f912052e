#!/bin/bash
# Round profile on the GPU box: kernel-trace stats, PMC traffic (separate passes, as MI355X_MICROARCH.md prescribes),
# in-graph timeline, and MFMA-busy counters of the prefill kernel.  Outputs under gpurun_out/prof_round/; summarised into
# profiles/ by probes/summarize_profile.py (run afterwards in the dev container).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/prof_round
rm -rf $O && mkdir -p $O
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cat $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/stats.log 2>&1
python3 probes/timeline.py $O/stats | tee $O/timeline.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --kernel-reps 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --kernel-reps 2 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefill_stats -- python3 probes/bench_prefill.py > $O/prefill_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/prefill_pmc -- python3 probes/bench_prefill.py > $O/prefill_pmc.log 2>&1 || echo "prefill pmc pass failed (see log)"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/prefill_pmc2 -- python3 probes/bench_prefill.py > $O/prefill_pmc2.log 2>&1 || echo "prefill pmc2 pass failed (see log)"
# keep only the small CSVs
find $O -name "*_agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
