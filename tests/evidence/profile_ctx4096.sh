cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python tests/evidence/ref_runtime_ctx4096.py 2>&1 | grep -v "^llama_\|^print_info\|^init:\|^llm_load\|^load\|^\.\.\.\.\.\|^set_abort" | tail -3
rm -rf gpurun_out/ctx4096_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctx4096_prof -- python3 tests/evidence/ref_runtime_ctx4096.py > gpurun_out/ctx4096_prof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ctx4096_prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if 'attn_' in r['Name'] or 'rows_io' in r['Name'] or 'convert_rows' in r['Name']:
        print(r['Name'][:110], r['Calls'], '%.1f us avg' % (float(r['AverageNs'])/1e3))
PY
find gpurun_out/ctx4096_prof -name "*kernel_trace.csv" -delete
