#!/usr/bin/env python3
"""Summarise gpurun_out/r03_final (probes/r03_evidence.sh) into profiles/r03_*: the bench line, rocprofv3 kernel statistics of the bench command and of
the end-to-end step, PMC traffic per launch of the decode kernel (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction) -> profiles/traffic.json,
stamps, A/B against the round-1 library, secondary configurations, harness outputs.  usage: python probes/r03_summarize.py"""
import hashlib, csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03_final")
P = os.path.join(ROOT, "profiles")
tag = "r03"


def one(pattern):
    # gpurun MERGES a run's files into the local directory: files of earlier evidence runs (other process ids in their names) stay behind.  The newest one is this run's.
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None


def cp(src, dst):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))
        return True
    print("missing", src)
    return False


def counters(path, key):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {c: sum(v) / len(v) for c, v in acc.items()}, (len(next(iter(acc.values()))) if acc else 0)


cp(os.path.join(SRC, "bench_n1.json"), f"{tag}_bench_n1.json")
cp(one("stats/**/*kernel_stats.csv"), f"{tag}_bench_kernel_stats.csv")
cp(one("e2e_stats/**/*kernel_stats.csv"), f"{tag}_e2e_kernel_stats.csv")
for a, b in (("timeline.txt", "bench_timeline.txt"), ("stamps_8b.txt", "decode_stamps_8b.txt"), ("stamps_70b.txt", "decode_stamps_70b.txt"), ("gemv_shapes.txt", "gemv_shapes.txt"), ("gemv_shapes_q5.txt", "gemv_shapes_q5.txt"), ("e2e_q5km.txt", "e2e_q5km.txt"), ("gemm_bench.txt", "gemm_bench.txt"), ("batched_bench.txt", "batched_bench.txt"),
             ("bench_prefill.txt", "bench_prefill.txt"), ("prefill_pmc.txt", "prefill_pmc.txt"), ("ablate.txt", "prefill_ablation.txt"), ("prefill_stamps.txt", "prefill_stamps.txt"), ("e2e_8b_q4k_plugin.txt", "e2e_8b_q4k_plugin.txt"), ("tinyllama_e2e_plugin.txt", "tinyllama_e2e_plugin.txt"),
             ("gpu_tests.log", "gpu_tests.txt"), ("smoke.log", "smoke.txt")):
    cp(os.path.join(SRC, a), f"{tag}_{b}")
for f in glob.glob(os.path.join(SRC, "tbo_*.txt")):
    cp(f, f"{tag}_test_backend_ops_{os.path.basename(f)[4:]}")

configs = {}
for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))):
    n = os.path.basename(f)[6:-5]
    if n == "n1":
        continue
    try:
        d = json.load(open(f))
    except Exception:
        continue
    configs[n] = {k: d[k] for k in ("value", "unit", "ms_per_step", "dtype")} | {"workload": d["config"]["workload"], "merge": d["config"]["merge"], "step": d["config"]["step"],
                                                                              "roofline": {k: d["roofline"][k] for k in ("achieved", "frac", "algorithmic_bytes_per_launch", "kernel_us")}}
json.dump(configs, open(os.path.join(P, f"{tag}_bench_configs.json"), "w"), indent=1)

stats = []
p = one("stats/**/*kernel_stats.csv")
if p:
    for r in csv.DictReader(open(p)):
        if "pkv::" in r["Name"]:
            stats.append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")})
e2e_stats = []
p = one("e2e_stats/**/*kernel_stats.csv")
if p:
    for r in csv.DictReader(open(p)):
        if "pkv::" in r["Name"]:
            e2e_stats.append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")})
bench = json.load(open(os.path.join(SRC, "bench_n1.json")))
summary = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs   (MI355X, ROCm 7.2, 200 graph-replayed steps + 20 warm-up + kernel-timing launches)",
           "kernel_stats": stats,
           "e2e_command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 probes/e2e_profile.py --steps 10",
           "e2e_kernel_stats": e2e_stats,
           "bench_line": {k: bench[k] for k in ("value", "ms_per_step")} | {"roofline": bench["roofline"], "e2e": bench.get("e2e"), "cpu_baseline": bench.get("cpu_baseline")}}
pf, pw = one("pmc_fetch/**/*counter_collection.csv"), one("pmc_write/**/*counter_collection.csv")
if pf and pw:
    cf, nf = counters(pf, "attn_decode_kernel")
    cw, nw = counters(pw, "attn_decode_kernel")
    if "FETCH_SIZE" in cf and "WRITE_SIZE" in cw:
        hbm = int(2 * cf["FETCH_SIZE"] * 1024 + cw["WRITE_SIZE"] * 1024)
        summary["pmc"] = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE  (separate passes) -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs --steps 20 --warmup 2 --kernel-reps 2",
                          "units": "KiB as reported by rocprofv3; on gfx950 FETCH_SIZE counts a wide coalesced read at 1/2 (MI355X_MICROARCH.md, HBM): HBM read bytes = 2 * FETCH_SIZE * 1024",
                          "attn_decode_kernel": {"FETCH_SIZE_avg_per_launch_KiB": cf["FETCH_SIZE"], "WRITE_SIZE_avg_per_launch_KiB": cw["WRITE_SIZE"], "launches": [nf, nw]},
                          "attn_decode_kernel_hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
        json.dump({"_doc": "HBM bytes per launch of pkv::attn_decode_kernel from rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, KiB->B; gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md), keyed by bench.py workload; source profiles/r03_summary.json (regenerated after the r03 kernel changes)",
                   "kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, "llama.cpp_test_amd", "csrc", "pkv_attn_decode.hip"), "rb").read()).hexdigest(),
                   "f16_ctx4096_nseq1": hbm}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
        print("traffic per launch", hbm, "algorithmic", bench["roofline"]["algorithmic_bytes_per_launch"])
json.dump(summary, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print("wrote profiles/%s_*" % tag)
