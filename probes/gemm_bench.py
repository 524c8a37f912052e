#!/usr/bin/env python3
"""probes/gemm_bench.py — the batched (prompt-processing) weight product pkv_mul_mat at llama-3-8B layer shapes: TFLOP/s per weight type"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _load_pkg
pkv = _load_pkg.load(); pkv.lib()
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
rs = {1: lambda K: 2 * K, 8: lambda K: K // 32 * 34, 12: lambda K: K // 256 * 144, 14: lambda K: K // 256 * 210}
for tname, t in (("f16", 1), ("q8_0", 8), ("q4_K", 12), ("q4_Kp", 12 | 0x100), ("q6_K", 14), ("q6_Kp", 14 | 0x100)):   # ..p: plane-layout rows (what the plugin keeps), random bytes
    SHAPES = ((4096, 4096, 512), (6144, 4096, 512), (14336, 4096, 512), (28672, 4096, 512), (4096, 14336, 512), (14336, 4096, 2048), (4096, 4096, 32))
    if os.environ.get("GEMM_TINY"):   # TinyLlama-1.1B layer shapes
        SHAPES = ((2048, 2048, 512), (5632, 2048, 512), (2048, 5632, 512), (32000, 2048, 512))
    for M, K, N in SHAPES:
        if t == 1:
            w = (torch.randn((M, K), device=dev, generator=gen) / K ** 0.5).to(torch.float16).view(torch.uint8).reshape(-1)
        else:
            w = torch.randint(0, 256, (M * rs[t & 0xFF](K),), dtype=torch.uint8, device=dev, generator=gen)
        x = torch.randn((N, K), device=dev, generator=gen)
        out = torch.empty((N, M), device=dev)
        for _ in range(3): pkv.mul_mat(w, t, M, K, x, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 10
        for _ in range(R): pkv.mul_mat(w, t, M, K, x, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
        print(f"{tname:5s} M {M:6d} K {K:6d} N {N:5d}: {dt * 1e6:9.1f} us  {2 * M * K * N / dt / 1e12:7.1f} TFLOP/s", flush=True)
