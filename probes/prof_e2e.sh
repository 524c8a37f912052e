#!/bin/bash
# kernel statistics of the end-to-end decode step (bench.py's e2e record alone): rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02/e2e_prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 probes/e2e_profile.py --steps 10 > $O/run.log 2>&1
tail -2 $O/run.log
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(7), ("%.2f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us avg", ("%.1f" % float(r["Percentage"])).rjust(6), "%")
print("total kernel time per step (13 steps + setup): see csv; sum ms", tot / 1e6)
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_agent_info.csv" -delete
