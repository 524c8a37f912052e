#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02/plugin_prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tests/evidence/ref_runtime_8b_q4k.py > $O/run.log 2>&1
grep "tok/s" $O/run.log
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(r['Name'][:100], r['Calls'], '%.1f us avg' % (float(r['AverageNs'])/1e3), '%.1f%%' % (100*float(r['TotalDurationNs'])/tot))
print('total kernel ms', tot/1e6)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
echo "unprofiled:"; python3 tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep "tok/s"
echo "GGML_MI355_KV_MIRROR=1:"; GGML_MI355_KV_MIRROR=1 python3 tests/evidence/ref_runtime_8b_q4k.py 2>&1 | grep "tok/s"
