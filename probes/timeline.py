"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py` into the in-graph timeline of one decode step:
average start-to-start interval, kernel duration and idle gap for each kernel of the (prepare, decode, merge) triple.
usage: python probes/timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, statistics, sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "pkv::" not in n:
            continue
        short = "store" if "kv_store" in n else "decode" if "attn_decode" in n else "merge" if "attn_combine" in n else n[:30]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
# keep only back-to-back (graph replay) kernels: gap to the previous kernel < 20 us
stats = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if s1 - e0 > 20000:
        continue
    d = stats.setdefault(n0 + "->" + n1, {"gap": [], "dur": [], "s2s": []})
    d["gap"].append(s1 - e0); d["dur"].append(e0 - s0); d["s2s"].append(s1 - s0)
for k, d in sorted(stats.items()):
    if len(d["gap"]) < 50:
        continue
    print(f"{k:24s} n={len(d['gap']):6d}  dur(first)={statistics.mean(d['dur'])/1e3:6.2f} us  gap={statistics.mean(d['gap'])/1e3:6.2f} us  "
          f"start-to-start={statistics.mean(d['s2s'])/1e3:6.2f} us (median {statistics.median(d['s2s'])/1e3:.2f})")
