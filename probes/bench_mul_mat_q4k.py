"""per-shape timing of the Q4_K mat-vec (the products of a Llama-3-8B layer at decode), weights rotated through > 512 MB so every
product streams from HBM as it does inside a 32-layer model.  python probes/bench_mul_mat_q4k.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _load_pkg
pkv = _load_pkg.load()
Q4_K = 12   # PKV_TYPE_Q4_K

dev = torch.device("cuda:0")
for name, M, K in [("wq/wo", 4096, 4096), ("wk/wv", 1024, 4096), ("gate/up", 14336, 4096), ("down", 4096, 14336)]:
    row = K // 256 * 144
    nbuf = max(2, (600 << 20) // (M * row) + 1)
    ws = [torch.randint(0, 256, (M * row,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    for w in ws:                      # keep d / dmin finite: overwrite the two halves of each block header with small f16 values
        w.view(-1, 144)[:, 0:4] = torch.tensor([0x00, 0x14, 0x00, 0x18], dtype=torch.uint8, device=dev)
    x = torch.randn(1, K, device=dev)
    out = torch.empty(1, M, device=dev)
    for i in range(nbuf):
        pkv.mul_mat(ws[i], Q4_K, M, K, x, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    n = 4 * nbuf
    with torch.cuda.graph(g):
        for i in range(n):
            pkv.mul_mat(ws[i % nbuf], Q4_K, M, K, x, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:8s} M={M:6d} K={K:6d}  {us:6.2f} us  {M * row / us / 1e6:5.2f} TB/s", flush=True)

# the shared-activation groups as ONE launch (pkv_mul_mat_multi)
for name, Ms, K in [("q+k+v", (4096, 1024, 1024), 4096), ("gate+up", (14336, 14336), 4096)]:
    row = K // 256 * 144
    tot = sum(Ms) * row
    nbuf = max(2, (600 << 20) // tot + 1)
    sets = []
    for _ in range(nbuf):
        ws = [torch.randint(0, 256, (M * row,), dtype=torch.uint8, device=dev) for M in Ms]
        for w in ws:
            w.view(-1, 144)[:, 0:4] = torch.tensor([0x00, 0x14, 0x00, 0x18], dtype=torch.uint8, device=dev)
        sets.append(ws)
    x = torch.randn(1, K, device=dev)
    for ws in sets:
        pkv.mul_mat_multi(ws, Q4_K, Ms, K, x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    n = 4 * nbuf
    keep = []
    with torch.cuda.graph(g):
        for i in range(n):
            keep.append(pkv.mul_mat_multi(sets[i % nbuf], Q4_K, Ms, K, x))
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:8s} M={Ms} K={K:6d}  {us:6.2f} us  {tot / us / 1e6:5.2f} TB/s", flush=True)
