#!/usr/bin/env python3
"""probes/ab_merge.py — the r03 merge kernel (one wave per head, no LDS) against the r02 one (probes/bin/libpkv_merge_old.so = HEAD's pkv_attn_decode.hip):
outputs bit for bit on several geometries, and us per decode step (attention + merge, hipGraph of 32 layers) for both."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "probes"))
import torch
from gemv_shapes import load_pkg

new, old = load_pkg("pkv_new", None), load_pkg("pkv_old", os.path.join(ROOT, "probes", "bin", "libpkv_merge_old.so"))
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
for (n_seq, H, HK, D, ctx) in [(1, 32, 8, 128, 4096), (3, 32, 8, 128, 1000), (2, 64, 8, 128, 16384), (4, 32, 4, 64, 640), (32, 32, 8, 128, 640)]:
    BS = 128
    bps = -(-ctx // BS)
    nb = n_seq * bps
    kp = (torch.rand((nb, HK, BS * D), generator=gen, device=dev) * 2 - 1).to(torch.float16).view(torch.uint8)
    vp = (torch.rand((nb, HK, BS * D), generator=gen, device=dev) * 2 - 1).to(torch.float16).view(torch.uint8)
    bt = torch.randperm(nb, generator=gen, device=dev).to(torch.int32).reshape(n_seq, bps).contiguous()
    sl = torch.randint(max(1, ctx - 300), ctx + 1, (n_seq,), generator=gen, device=dev).to(torch.int32)
    q = torch.rand((n_seq, H, D), generator=gen, device=dev) * 2 - 1
    outs = []
    for m in (new, old):
        ws = m.attn_workspace(n_seq, H, D, ctx, BS)
        o = m.attn_decode(q, kp, vp, bt, sl, n_head_kv=HK, type_k=m.F16, type_v=m.F16, block_size=BS, scale=D ** -0.5, max_seq_len=ctx, workspace=ws)
        torch.cuda.synchronize()
        outs.append(o.clone())
    same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    print(f"n_seq {n_seq} heads {H}/{HK} D {D} ctx {ctx}: merge outputs bit-identical to r02: {same}", flush=True)
    assert same
