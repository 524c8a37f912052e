"""Summarise gpurun_out/prof_round (probes/profile_round.sh) into profiles/<tag>_*: kernel stats CSV, PMC traffic per launch
(FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction), the in-graph timeline, the prefill MFMA-busy counters.
usage: python probes/summarize_profile.py [tag, default r01]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(ROOT, "profiles")


def one(pattern):
    return glob.glob(os.path.join(SRC, pattern), recursive=True)[0]


def short(n):
    for k in ("attn_decode_kernel", "attn_combine_kernel", "kv_store_kernel", "rope_cache_kernel", "attn_prefill_kernel"):
        if k in n:
            return k
    return None


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in acc.items()}


shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(one("prefill_stats/**/*kernel_stats.csv"), os.path.join(P, f"{tag}_prefill_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench_n1.json"), os.path.join(P, f"{tag}_bench_n1.json"))
shutil.copy(os.path.join(SRC, "timeline.txt"), os.path.join(P, f"{tag}_bench_timeline.txt"))
stats = []
with open(one("stats/**/*kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        if "pkv::" in r["Name"]:
            stats.append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")})
fetch, write = counters(one("pmc_fetch/**/*counter_collection.csv")), counters(one("pmc_write/**/*counter_collection.csv"))
per = {}
for k in fetch:
    per[k] = {"FETCH_SIZE_avg_per_launch": fetch[k].get("FETCH_SIZE"), "WRITE_SIZE_avg_per_launch": write.get(k, {}).get("WRITE_SIZE")}
dec = per["attn_decode_kernel"]
hbm = int(2 * dec["FETCH_SIZE_avg_per_launch"] * 1024 + dec["WRITE_SIZE_avg_per_launch"] * 1024)
bench = json.load(open(os.path.join(SRC, "bench_n1.json")))
summary = {
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (MI355X, ROCm 7.2, 200 graph-replayed steps + 20 warm-up + kernel-timing launches)",
    "kernel_stats": stats,
    "pmc": {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE  (separate passes) -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --kernel-reps 2",
            "units": "KiB as reported by rocprofv3; on gfx950 FETCH_SIZE counts a wide coalesced read at 1/2 (MI355X_MICROARCH.md §HBM): HBM read bytes = 2 * FETCH_SIZE * 1024",
            "per_kernel": per},
    "attn_decode_kernel_hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
}
try:   # MFMA-busy fraction of the prefill kernel, per case (kernel template type, grid size)
    def per_case(path):
        out = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for r in csv.DictReader(f):
                if "attn_prefill_kernel" in r["Kernel_Name"]:
                    out[(r["Kernel_Name"].split("<")[1].split(",")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return out
    pa, pb = per_case(one("prefill_pmc/**/*counter_collection.csv")), per_case(one("prefill_pmc2/**/*counter_collection.csv"))
    cases = {}
    for k in pa:
        mf = sum(pa[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(pa[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(pb[k]["GRBM_GUI_ACTIVE"]) / len(pb[k]["GRBM_GUI_ACTIVE"])
        cases[f"type_k={k[0]} grid_threads={k[1]}"] = {"SQ_VALU_MFMA_BUSY_CYCLES": mf, "GRBM_GUI_ACTIVE": gui, "mfma_busy_frac": round(mf / (gui / 8 * 1024), 4)}
    summary["prefill_pmc"] = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES  /  --pmc GRBM_GUI_ACTIVE (separate passes) -- python3 probes/bench_prefill.py",
                              "reading": "SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over the 1024 SIMDs (32 per v_mfma_f32_32x32x16_f16); GRBM_GUI_ACTIVE sums the 8 XCDs: mfma_busy_frac = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)",
                              "per_case": cases}
except Exception as e:   # counters not available on this box
    summary["prefill_pmc"] = {"error": str(e)}
for extra, name in (("bench_prefill.txt", "bench_prefill"), ("bench_prefill_70b.txt", "bench_prefill_70b"), ("batched_bench.txt", "batched_bench"),
                    ("tinyllama_e2e.txt", "tinyllama_e2e_plugin"), ("e2e_8b_q4k.txt", "e2e_8b_q4k_plugin")):
    if os.path.exists(os.path.join(SRC, extra)):
        shutil.copy(os.path.join(SRC, extra), os.path.join(P, f"{tag}_{name}.txt"))
for f in glob.glob(os.path.join(SRC, "tbo_*.txt")):
    shutil.copy(f, os.path.join(P, f"{tag}_test_backend_ops_{os.path.basename(f)[4:]}"))
cfg = {}
for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))):
    d = json.load(open(f))
    cfg[os.path.basename(f)[:-5]] = {"tok_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_us": d["roofline"]["kernel_us"],
                                      "frac_events": d["roofline"]["frac"], "bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"],
                                      "step": d["config"]["step"], "workload": d["config"]["workload"]}
json.dump(cfg, open(os.path.join(P, f"{tag}_bench_configs.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
tr = json.load(open(os.path.join(P, "traffic.json")))
tr["f16_ctx4096_nseq1"] = hbm
json.dump(tr, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps({"decode_hbm_bytes": hbm, "per_kernel": per, "prefill": summary["prefill_pmc"]}, indent=1))
for s_ in stats:
    print(s_["Name"][:70], s_["Calls"], s_["AverageNs"])
