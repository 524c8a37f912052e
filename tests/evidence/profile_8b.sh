cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/e2e8b_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2e8b_prof -- python3 tests/evidence/ref_runtime_8b_q4k.py > gpurun_out/e2e8b_prof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/e2e8b_prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:10]:
    print(r['Name'][:80], r['Calls'], '%.1f us avg' % (float(r['AverageNs'])/1e3), '%.1f%%' % (100*float(r['TotalDurationNs'])/tot))
print('total kernel ms', tot/1e6)
PY
grep "tok/s" gpurun_out/e2e8b_prof.log
find gpurun_out/e2e8b_prof -name "*kernel_trace.csv" -delete
