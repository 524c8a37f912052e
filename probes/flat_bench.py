"""how fast the coverage kernel (pkv_flash_attn_ext_flat) is on shapes a bf16-cache or head-size-256 model would bring: one query row at ctx 4096 (decode) and a 512-token prompt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import _load_pkg
pkv = _load_pkg.load()
T = {"f16": 1, "bf16": 30, "q8_0": 8, "q5_0": 6}
def run(D, nh, nkv, n_kv, n_q, t):
    rb = pkv.row_size(T[t], D)
    q = torch.rand((nh, n_q, D), device="cuda") * 2 - 1
    reps = 24                                             # distinct K / V per repetition: 24 x (K + V) past the Infinity Cache for the long shapes
    ks = [torch.randint(0, 255, (nkv, n_kv, rb), dtype=torch.uint8, device="cuda") for _ in range(reps)]
    vs = [torch.randint(0, 255, (nkv, n_kv, rb), dtype=torch.uint8, device="cuda") for _ in range(reps)]
    if t in ("f16", "bf16"):                              # random bytes hold inf / nan patterns: clear the exponent's top bit
        for x in ks + vs: x[..., 1::2] &= 0x3F if t == "f16" else 0xBF; 
    out = torch.empty((n_q, nh, D), device="cuda")
    ws = torch.empty(max(16, pkv.lib().pkv_flash_attn_ext_flat_workspace_bytes(n_q, nh, D, n_kv)), dtype=torch.uint8, device="cuda")   # split-KV scratch (decode shapes)
    for i in range(3): pkv.flash_attn_ext_flat(q, ks[i], vs[i], None, type_k=T[t], type_v=T[t], scale=D ** -0.5, out=out, workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): pkv.flash_attn_ext_flat(q, ks[i], vs[i], None, type_k=T[t], type_v=T[t], scale=D ** -0.5, out=out, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byt = 2 * nkv * n_kv * rb
    fl = 4.0 * nh * D * n_q * n_kv
    print(f"{t:5s} D {D:3d} heads {nh:2d}/{nkv} n_kv {n_kv:5d} n_q {n_q:4d}: {us:9.1f} us   {byt / us / 1e3:7.1f} GB/s of KV   {fl / us / 1e6:7.2f} TFLOP/s", flush=True)
for t in ("f16", "bf16", "q8_0", "q5_0"):
    run(128, 32, 8, 4096, 1, t)
run(256, 16, 8, 4096, 1, "f16")
run(256, 16, 8, 4096, 1, "bf16")
run(128, 32, 8, 512, 512, "bf16")
run(256, 16, 8, 512, 512, "f16")
