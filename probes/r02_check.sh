#!/bin/bash
# one GPU call: [tests] -> bench (with e2e) -> per-shape kernel times of the e2e step.  usage: probes/r02_check.sh <tag> [pytest args...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=$1; shift
O=gpurun_out/r02; mkdir -p $O
if [ $# -gt 0 ]; then
  timeout -k 10 900 python -m pytest "$@" -m gpu -x -q > $O/tests_$T.log 2>&1; rc=$?
  tail -4 $O/tests_$T.log
  [ $rc -ne 0 ] && { grep -E "^E|Error|assert" $O/tests_$T.log | head -20; exit $rc; }
fi
timeout -k 10 400 python3 bench.py > $O/bench_$T.json 2> $O/bench_$T.err || { tail -5 $O/bench_$T.err; exit 1; }
python3 - $O/bench_$T.json <<PY
import json, sys
d = json.load(open(sys.argv[1]))
print("attn: %.1f tok/s  %.4f ms/step  kernel %.3f us frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_us"]["mean"], d["roofline"]["frac"]))
e = d.get("e2e")
if e: print("e2e : %.1f tok/s  %.4f ms/token  frac %.4f" % (e["value"], e["ms_per_token"], e["roofline"]["frac"]))
PY
rm -rf $O/e2e_prof_$T && mkdir -p $O/e2e_prof_$T
rocprofv3 --kernel-trace --output-format csv -d $O/e2e_prof_$T -- python3 probes/e2e_profile.py --steps 10 > $O/e2e_prof_$T/run.log 2>&1
tail -1 $O/e2e_prof_$T/run.log
f=$(find $O/e2e_prof_$T -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "pkv" in n:
        i = n.find("pkv::(anonymous namespace)::"); n = n[i + 28:] if i >= 0 else n
        d[(n[:56], r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items()):
    print("%-58s grid %8s wg %5s n %5d  med %7.2f  min %7.2f us" % (k[0], k[1], k[2], len(v), statistics.median(v), min(v)))
PY
find $O/e2e_prof_$T -name "*.csv" -size +3M -delete; find $O/e2e_prof_$T -name "*agent_info*" -delete
