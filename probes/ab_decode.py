#!/usr/bin/env python3
"""probes/ab_decode.py — A/B of two builds of libmi355_paged_kv.so in ONE process, interleaved rounds (cdna_hip_programming.md §5.4
rule 24): the decode kernel alone (graph of n_layer launches, PKV_ATTN_NO_COMBINE) and the whole step (decode + merge), us per layer.
usage: python probes/ab_decode.py [--b probes/bin/libmi355_paged_kv_r01.so] [--model 70b --ctx 16384] [--n-seq N] [--kv-type q8_0]"""
import argparse, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def load_pkg(name, lib_path):
    root = os.path.join(ROOT, "llama.cpp_test_amd")
    if lib_path:
        os.environ["PKV_LIB_PATH"] = os.path.abspath(lib_path)
    else:
        os.environ.pop("PKV_LIB_PATH", None)
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.lib()
    os.environ.pop("PKV_LIB_PATH", None)
    return mod


def graphs(w, pkv, stream):
    L = w.g["n_layer"]
    out = {}
    with torch.cuda.stream(stream):
        for name, fn in (("kernel", lambda: [w.layer_step(l, flags=pkv.ATTN_NO_COMBINE) for l in range(L)]), ("step", w.step)):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                fn()
            out[name] = g
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default=None); ap.add_argument("--b", nargs="+", default=["probes/bin/libmi355_paged_kv_r01.so"])
    ap.add_argument("--model", default="8b"); ap.add_argument("--ctx", type=int, default=4096); ap.add_argument("--n-seq", type=int, default=1)
    ap.add_argument("--kv-type", default="f16"); ap.add_argument("--rounds", type=int, default=12); ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    mods = {"A (this tree)" if not a.a else a.a: load_pkg("pkv_a", a.a)}
    for i, b in enumerate(a.b):
        mods[b] = load_pkg(f"pkv_b{i}", b)
    g = bench.GEOM if a.model == "8b" else bench.GEOM_70B
    L = g["n_layer"]
    stream = torch.cuda.Stream()
    ws, gs = {}, {}
    for k, m in mods.items():
        t = m.TYPE_BY_NAME[a.kv_type]
        ws[k] = bench.Workload(m, torch, ctx=a.ctx, n_seq=a.n_seq, type_k=t, type_v=t, seed=1234, g=g)
        gs[k] = graphs(ws[k], m, stream)
    res = {k: {"kernel": [], "step": []} for k in mods}
    with torch.cuda.stream(stream):
        for r in range(a.rounds):
            for k in mods:
                for what in ("kernel", "step"):
                    gr = gs[k][what]
                    gr.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(a.reps):
                        gr.replay()
                    e1.record(stream)
                    torch.cuda.synchronize()
                    if r >= 2:
                        res[k][what].append(e0.elapsed_time(e1) * 1e3 / (a.reps * L))
    bytes_launch = a.n_seq * bench.algorithmic_bytes(a.ctx, t, t, list(mods.values())[0], g)
    print(f"Llama-3-{a.model.upper()} geometry, {a.n_seq} seq, ctx {a.ctx}, {a.kv_type} KV; us per layer, median [min] over {a.rounds - 2} interleaved rounds")
    for k in mods:
        ke, st = np.array(res[k]["kernel"]), np.array(res[k]["step"])
        print(f"  {k:45s} decode kernel {np.median(ke):7.2f} [{ke.min():6.2f}] us = {bytes_launch / np.median(ke) / 1e3 / 8000:.3f} of 8 TB/s | step (decode + merge) {np.median(st):7.2f} [{st.min():6.2f}] us")


if __name__ == "__main__":
    main()
