#!/usr/bin/env python3
"""probes/decode_stamps.py — where the decode kernel's time goes: runs the STAMPED diagnostic instantiation of
attn_decode_kernel (pkv_attn_decode_stamps) on bench.py's headline workload, layers cycled (cold KV), and prints per-phase tables:
shader-clock deltas between consecutive stamps (median / p90 over waves) and the wall-clock (100 MHz) spread of workgroup start/end.
The stamped build's total time is NOT a measurement; the shares are.  usage: python probes/decode_stamps.py [--model 70b --ctx 16384]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _load_pkg
import bench

NAMES = ["entry", "scalars (args, table, len) in SGPRs", "all loads issued", "Q operand ready (+1st barrier)", "K landed", "QK^T + softmax + P strip",
         "V landed", "PV + lane sums + sO", "barrier passed", "merge + partial stores issued"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="8b"); ap.add_argument("--ctx", type=int, default=4096); ap.add_argument("--n-seq", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--out", default=None)
    ap.add_argument("--mode", type=int, default=1, help="1 = forced vmcnt waits in front of the landed stamps (phase shares), 2 = stamps only (the product kernel's own overlap)")
    a = ap.parse_args()
    pkv = _load_pkg.load(); pkv.lib()
    g = bench.GEOM if a.model == "8b" else bench.GEOM_70B
    w = bench.Workload(pkv, torch, ctx=a.ctx, n_seq=a.n_seq, type_k=pkv.F16, type_v=pkv.F16, seed=1234, g=g)
    L = g["n_layer"]
    kw = dict(n_head_kv=g["n_head_kv"], type_k=w.tk, type_v=w.tv, block_size=g["block_size"], scale=w.scale, max_seq_len=w.ctx, workspace=w.ws, rope_cache=w.rope_cs)
    # the stamped launches run the way bench.py's kernel graph does: n_layer dependent launches captured in ONE hipGraph, replayed back
    # to back (an isolated launch from an idle GPU has a different dispatch spread and clock); every launch stamps into its own slab
    n_wg = a.n_seq * g["n_head_kv"] * ((a.ctx + 127) // 128)
    slabs = torch.zeros((L, n_wg, 8, pkv.DECODE_NSTAMP), dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    def run():
        for l in range(L):
            pkv.attn_decode_stamps(w.q[l], w.k_cur[l], w.v_cur[l], w.pos, w.rope, w.k_pools[l], w.v_pools[l], w.block_table, w.seq_lens, mode=a.mode, stamps_out=slabs[l], **kw)
    all_st = []
    with torch.cuda.stream(stream):
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            run()
        for r in range(a.rounds):
            gr.replay(); gr.replay(); gr.replay()
            torch.cuda.synchronize()
            all_st.append(slabs.clone())
    S = torch.cat(all_st).cpu().numpy().astype(np.int64)        # [launch][wg][wave][12]
    # gap between consecutive launches of a replay: first entry of launch l+1 minus last exit of launch l (wall clock)
    SS = torch.stack(all_st).cpu().numpy().astype(np.int64)      # [round][L][wg][wave][12]
    first_in = SS[..., 0].reshape(SS.shape[0], L, -1).min(axis=2)
    last_out = SS[..., 11].reshape(SS.shape[0], L, -1).max(axis=2)
    gaps = (first_in[:, 1:] - last_out[:, :-1]) * 10.0
    cadence = (first_in[:, 1:] - first_in[:, :-1]) * 10.0
    n_launch, n_wg = S.shape[0], S.shape[1]
    cyc = S[..., 1:11]                                            # s_memtime
    d = np.diff(cyc, axis=-1).astype(np.float64)                  # 9 phases
    tot = (cyc[..., -1] - cyc[..., 0]).astype(np.float64)
    real0, real1 = S[..., 0], S[..., 11]
    wall = (real1 - real0) * 10.0                                 # ns per wave
    # shader clock estimate: cycles / wall
    clk = np.median(tot / np.maximum(wall, 1.0))                  # cycles per ns
    lines = []
    lines.append(f"stamp mode {a.mode}" + (" (no forced waits: 'landed' stamps only mark program order)" if a.mode == 2 else ""))
    lines.append(f"workload: Llama-3-{a.model.upper()} geometry, {a.n_seq} seq, ctx {a.ctx}, f16 KV; {n_launch} stamped launches x {n_wg} workgroups x 8 waves")
    lines.append(f"shader clock ~ {clk:.2f} GHz (median of cycles / 100 MHz wall per wave)")
    lines.append(f"{'phase (ends at)':46s} {'median cyc':>10s} {'p90 cyc':>9s} {'median us':>10s} {'share':>7s}")
    med_tot = np.median(tot)
    for i in range(9):
        m, p = np.median(d[..., i]), np.percentile(d[..., i], 90)
        lines.append(f"{NAMES[i + 1]:46s} {m:10.0f} {p:9.0f} {m / clk / 1e3:10.3f} {m / med_tot:7.1%}")
    lines.append(f"{'wave total (entry -> stores issued)':46s} {med_tot:10.0f} {np.percentile(tot, 90):9.0f} {med_tot / clk / 1e3:10.3f}")
    # per-launch wall-clock picture, relative to the first wave of the launch to start
    t0 = real0.reshape(n_launch, -1).min(axis=1)
    start = (real0.reshape(n_launch, -1) - t0[:, None]) * 10.0
    end = (real1.reshape(n_launch, -1) - t0[:, None]) * 10.0
    lines.append(f"in-graph cadence (first entry of launch l+1 - first entry of launch l): median {np.median(cadence):.0f} ns; gap (first entry of l+1 - last wave end of l): median {np.median(gaps):.0f} ns, min {gaps.min():.0f}, max {gaps.max():.0f}")
    lines.append("wall clock per launch (ns after the first wave's entry; 10 ns resolution), median over launches:")
    lines.append(f"  last wave entry      : median {np.median(start.max(axis=1)):7.0f}   (dispatch spread of the grid)")
    lines.append(f"  first wave done      : median {np.median(end.min(axis=1)):7.0f}")
    lines.append(f"  median wave done     : median {np.median(np.median(end, axis=1)):7.0f}")
    lines.append(f"  last wave done       : median {np.median(end.max(axis=1)):7.0f}   (= in-kernel span of the launch)")
    # which phase ends when, in wall terms: convert each wave's cycle stamps to ns after launch start using its own clock ratio
    ratio = wall / np.maximum(tot, 1.0)
    rel = (cyc - cyc[..., :1]) * ratio[..., None] + (real0 * 10.0 - t0[:, None, None] * 10.0)[..., None]
    for i, nm in enumerate(["entry"] + NAMES[1:]):
        v = rel[..., i].reshape(n_launch, -1)
        lines.append(f"  {nm:44s}: median wave {np.median(np.median(v, axis=1)):7.0f} ns, last wave {np.median(v.max(axis=1)):7.0f} ns")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
