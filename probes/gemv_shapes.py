#!/usr/bin/env python3
"""probes/gemv_shapes.py — the decode-step products of Llama-3-8B (Q4_K_M mix) one shape at a time: a hipGraph of 32 dependent launches cycling
through enough distinct weight buffers to defeat the 256 MiB Infinity Cache, us per launch and GB/s of weight bytes (in-graph gap included, like
bench.py's kernel_us).  Several builds of the library can be compared in one process: --libs a.so b.so (interleaved rounds).
Weights are random bytes (timing only)."""
import argparse, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


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


SHAPES = [  # name, [(type, M)], K, norm, residual, silu
    ("wq+wk+wv q4_K (norm)", [(12, 4096), (12, 1024), (12, 1024)], 4096, True, False, False),
    ("wq+wk q4_K + wv q6_K (norm)", [(12, 4096), (12, 1024), (14, 1024)], 4096, True, False, False),
    ("wo q4_K (+residual)", [(12, 4096)], 4096, False, True, False),
    ("gate+up q4_K (norm, SwiGLU)", [(12, 14336), (12, 14336)], 4096, True, False, True),
    ("down q4_K (+residual)", [(12, 4096)], 14336, False, True, False),
    ("down q6_K (+residual)", [(14, 4096)], 14336, False, True, False),
    ("lm_head q6_K (norm)", [(14, 128256)], 4096, True, False, False),
]
SHAPES_Q5 = [  # --q5: the same layer in the tensor mix of a Q5_K_M file
    ("wq+wk+wv q5_K (norm)", [(13, 4096), (13, 1024), (13, 1024)], 4096, True, False, False),
    ("wq+wk q5_K + wv q6_K (norm)", [(13, 4096), (13, 1024), (14, 1024)], 4096, True, False, False),
    ("wo q5_K (+residual)", [(13, 4096)], 4096, False, True, False),
    ("gate+up q5_K (norm, SwiGLU)", [(13, 14336), (13, 14336)], 4096, True, False, True),
    ("down q5_K (+residual)", [(13, 4096)], 14336, False, True, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=[]); ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--hot", type=int, default=0, help="cycle through only this many weight buffers (1-2: the weights stay in the 256 MiB Infinity Cache): what a product would take if its weights had been prefetched")
    ap.add_argument("--q5", action="store_true", help="the q5_K shapes instead")
    ap.add_argument("--packed", action="store_true", help="second arm: the same library on plane-layout rows (type | PKV_TYPE_PACKED; random bytes, timing only)")
    a = ap.parse_args()
    mods = {"this tree": load_pkg("pkv_a", None)}
    flag = {k: 0 for k in mods}
    if a.packed:
        mods = {"gguf order": mods["this tree"], "packed": mods["this tree"]}
        flag = {"gguf order": 0, "packed": 0x100}
    for i, l in enumerate(a.libs):
        mods[os.path.basename(l)] = load_pkg(f"pkv_l{i}", l)
        flag[os.path.basename(l)] = 0x100 if a.packed else 0
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(1)
    st = torch.cuda.Stream()
    rs = {12: 144, 13: 176, 14: 210}
    for name, mats, K, norm, resid, silu in (SHAPES_Q5 if a.q5 else SHAPES):
        nbytes = sum(M * (K // 256) * rs[t] for t, M in mats)
        n_buf = max(2, min(32, int(600e6 // nbytes) + 1))
        if a.hot: n_buf = a.hot
        n_launch = 32 if nbytes < 200e6 else 8
        ws = [[torch.randint(0, 256, (M * (K // 256) * rs[t],), dtype=torch.uint8, device=dev, generator=gen) for t, M in mats] for _ in range(n_buf)]
        x = torch.randn(K, device=dev, generator=gen)
        nw = torch.ones(K, device=dev) if norm else None
        outs = [torch.zeros(M, device=dev) for _, M in mats]
        graphs = {}
        with torch.cuda.stream(st):
            for k, m in mods.items():
                fl = flag[k]

                def run():
                    for i in range(n_launch):
                        w = ws[i % n_buf]
                        if silu:
                            m.gemv_q(w, [t | fl for t, _ in mats], [M for _, M in mats], K, None, [outs[0]], silu_mul=True, x=x, norm_w=nw, eps=1e-5)
                        elif resid:
                            m.gemv_q(w, [t | fl for t, _ in mats], [M for _, M in mats], K, None, [outs[0]], residual=outs[0], x=x)
                        else:
                            m.gemv_q(w, [t | fl for t, _ in mats], [M for _, M in mats], K, None, outs, x=x, norm_w=nw, eps=1e-5)
                run(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    run()
                graphs[k] = g
            res = {k: [] for k in mods}
            for r in range(a.rounds):
                for k in mods:
                    g = graphs[k]; g.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(a.reps): g.replay()
                    e1.record(st); torch.cuda.synchronize()
                    if r >= 1: res[k].append(e0.elapsed_time(e1) * 1e3 / (a.reps * n_launch))
        print(f"{name:32s} {nbytes / 1e6:7.1f} MB/launch: " + " | ".join(f"{k}: {np.median(v):6.2f} us {nbytes / np.median(v) / 1e3:5.0f} GB/s" for k, v in res.items()), flush=True)
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
