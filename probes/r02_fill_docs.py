#!/usr/bin/env python3
"""Regenerate BASELINE.md §5 (round-2 results table) from probes/r02_baseline_table.tmpl and profiles/r02_* (after probes/r02_summarize.py)."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
b = json.load(open(os.path.join(P, "r02_bench_n1.json"))); c = json.load(open(os.path.join(P, "r02_bench_configs.json"))); s = json.load(open(os.path.join(P, "r02_summary.json")))
rp = [r for r in s["kernel_stats"] if "attn_decode_kernel" in r["Name"]][0]
def gbs(v): return "%s" % format(round(v["roofline"]["achieved"]), ",")
def pct(v): return "%.1f" % (100 * v["roofline"]["frac"])
def toks(v): return format(round(v["value"]), ",")
plug = re.search(r"([\d.]+) tok/s \(([\d.]+) ms/token\)", open(os.path.join(P, "r02_e2e_8b_q4k_plugin.txt")).read())
tiny = re.findall(r"MI355X, TinyLlama.*?tg128: ([\d.]+) tok/s", open(os.path.join(P, "r02_tinyllama_e2e_plugin.txt")).read())
bb = re.search(r"S_PP (\d+) t/s \| T_TG [\d.]+ ms  S_TG (\d+) t/s", open(os.path.join(P, "r02_batched_bench.txt")).read())
gm = re.findall(r"q4_K  M  14336 K   4096 N +(\d+): +[\d.]+ us +([\d.]+) TFLOP/s", open(os.path.join(P, "r02_gemm_bench.txt")).read())
pf = re.findall(r"f16 n_seq +(\d+) n_q +(\d+) past +0: +[\d.]+ us +([\d.]+) TFLOP/s", open(os.path.join(P, "r02_bench_prefill.txt")).read())
pfd = {(a, q): t for a, q, t in pf}
e = b["e2e"]
rep = {
    "R2_HEAD_TOKS": toks(b), "R2_HEAD_GBS": format(round(b["roofline"]["achieved"]), ","), "R2_HEAD_US": "%.2f" % b["roofline"]["kernel_us"]["mean"],
    "R2_ROCPROF_US": "%.2f" % (float(rp["AverageNs"]) / 1e3), "R2_HEAD_PCT": "%.1f" % (100 * b["roofline"]["frac"]),
    "R2_E2E_TOKS": "%.0f" % e["value"], "R2_E2E_MS": "%.2f" % e["ms_per_token"], "R2_E2E_GBS": format(round(e["roofline"]["achieved"]), ","), "R2_E2E_PCT": "%.1f" % (100 * e["roofline"]["frac"]),
    "R2_E2E_CPU": "%.1f" % e["cpu_baseline"]["value"], "R2_PLUGIN_TOKS": "%s" % round(float(plug.group(1))), "R2_TINY_TOKS": "%s" % round(float(tiny[-1])),
    "R2_Q8_TOKS": toks(c["q8_0"]), "R2_Q4_TOKS": toks(c["q4_0"]), "R2_Q8_GBS": gbs(c["q8_0"]), "R2_Q4_GBS": gbs(c["q4_0"]), "R2_Q8_PCT": pct(c["q8_0"]), "R2_Q4_PCT": pct(c["q4_0"]),
    "R2_N32_TOKS": toks(c["nseq32"]), "R2_N32_GBS": gbs(c["nseq32"]), "R2_N32_PCT": pct(c["nseq32"]),
    "R2_N32Q8_TOKS": toks(c["q8_0_nseq32"]), "R2_N32Q4_TOKS": toks(c["q4_0_nseq32"]), "R2_N32Q8_GBS": gbs(c["q8_0_nseq32"]), "R2_N32Q4_GBS": gbs(c["q4_0_nseq32"]),
    "R2_N32Q8_PCT": pct(c["q8_0_nseq32"]), "R2_N32Q4_PCT": pct(c["q4_0_nseq32"]),
    "R2_N32C640_TOKS": toks(c["nseq32_ctx640"]), "R2_N32C640_GBS": gbs(c["nseq32_ctx640"]), "R2_N32C640_PCT": pct(c["nseq32_ctx640"]),
    "R2_BATCHED": "S_PP %s / S_TG %s" % (format(int(bb.group(1)), ","), format(int(bb.group(2)), ",")),
    "R2_70B_TOKS": toks(c["70b_ctx16384"]), "R2_70B_GBS": gbs(c["70b_ctx16384"]), "R2_70B_PCT": pct(c["70b_ctx16384"]),
    "R2_GEMM": " / ".join(t for _, t in gm), "R2_PREFILL": "%s / %s" % (pfd.get(("32", "512"), "?"), pfd.get(("1", "4096"), "?")),
}
tl = open(os.path.join(P, "r02_tinyllama_e2e_plugin.txt")).read()
rep["R2_TINY_PP"] = format(int(re.findall(r"MI355X, TinyLlama.*?pp512: (\d+) tok/s", tl)[-1]), ",")
rep["R2_TINY_CPUPP"] = re.findall(r"CPU backend.*?pp512: (\d+) tok/s", tl)[-1]
rep["R2_TINY_CPU"] = "%.0f" % float(re.findall(r"CPU backend.*?tg128: ([\d.]+) tok/s", tl)[-1])
pl = open(os.path.join(P, "r02_e2e_8b_q4k_plugin.txt")).read()
m8 = re.findall(r"pp512: (\d+) tok/s", pl)
rep["R2_PLUGIN_PP"] = format(int(m8[-1]), ",") if m8 else "n/a"
rep["R2_CPU_ATTN"] = "%.1f" % b["cpu_baseline"]["value"]
pa, pb = float(pfd.get(("32", "512"), 0)), float(pfd.get(("1", "4096"), 0))
rep["R2_PREFILL_PCT"] = "%.0f %% / %.0f %%" % (pa / 25, pb / 25)
# BASELINE.md section 5 is regenerated from probes/r02_baseline_table.tmpl
sec = open(os.path.join(ROOT, "probes", "r02_baseline_table.tmpl")).read()
for k in sorted(rep, key=len, reverse=True):
    sec = sec.replace(k, rep[k])
assert "R2_" not in sec, [w for w in sec.split() if "R2_" in w]
fn = os.path.join(ROOT, "BASELINE.md")
t = open(fn).read()
t = t[:t.index("## 5. Results table (round 2")] + sec
open(fn, "w").write(t)
print(rep)
