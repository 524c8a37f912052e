#!/usr/bin/env python3
"""probes/check_rope_exact.py — is the device RoPE bit-identical to the CPU oracle's (glibc cosf/sinf)?  rope_f32, rope_cache and the
RoPE'd KV store for every KV type.  Prints mismatch counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _load_pkg
from oracle import cpu_oracle as O
pkv = _load_pkg.load(); pkv.lib()
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(a).cuda()
bad = 0
for D, mode, ff_on in [(128, 0, False), (128, 0, True), (64, 0, False), (128, 2, False)]:
    n_tok, nh = 300, 8
    x = rng.uniform(-1, 1, (n_tok, nh, D)).astype(np.float32)
    pos = np.concatenate([np.arange(0, 150), rng.integers(0, 131072, 150)]).astype(np.int32)
    ff = rng.uniform(1, 8, D // 2).astype(np.float32) if ff_on else None
    rp = pkv.rope_params(n_dims=D, mode=mode, freq_factors=dev(ff) if ff_on else None)
    got = pkv.rope_f32(dev(x), dev(pos), rp).cpu().numpy()
    ref = O.rope(x, pos, ff, n_dims=D, mode=mode)
    mm = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    print(f"rope_f32 D={D} mode={mode} ff={ff_on}: {mm} of {got.size} elements differ, max abs {np.abs(got - ref).max():.3g}")
    bad += mm
    # yarn
    rp2 = pkv.rope_params(n_dims=D, mode=mode, freq_scale=0.25, ext_factor=1.0, attn_factor=1.0, n_ctx_orig=4096)
    got = pkv.rope_f32(dev(x), dev(pos), rp2).cpu().numpy()
    ref = O.rope(x, pos, None, n_dims=D, mode=mode, freq_scale=0.25, ext_factor=1.0, attn_factor=1.0, n_ctx_orig=4096)
    mm = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    print(f"rope_f32 YaRN D={D} mode={mode}: {mm} of {got.size} differ, max abs {np.abs(got - ref).max():.3g}")
    bad += mm
for t in (pkv.F16, pkv.Q8_0, pkv.Q4_0):
    D, nkv, bs, n_tok = 128, 8, 128, 128
    k = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
    pos = rng.integers(0, 32768, n_tok).astype(np.int32)
    sb, so = np.zeros(n_tok, np.int32), np.arange(n_tok, dtype=np.int32)
    kp_d = torch.zeros(pkv.pool_bytes(t, 1, nkv, bs, D), dtype=torch.uint8, device="cuda")
    rp = pkv.rope_params(n_dims=128)
    pkv.kv_store(dev(k), None, kp_d, None, type_k=t, type_v=1, slot_block=dev(sb), slot_off=dev(so), block_size=bs, pos=dev(pos), rope=rp)
    torch.cuda.synchronize()
    kp = O.pool_alloc(t, 1, nkv, bs, D)
    O.pool_store(kp, t, 0, O.from_float(t, O.rope(k, pos, n_dims=D)), sb, so, D=D, block_size=bs)
    mm = int((kp_d.cpu().numpy().reshape(-1) != kp.reshape(-1)).sum())
    print(f"RoPE'd KV store type {t}: {mm} pool bytes differ")
    bad += mm
print("ROPE EXACT" if bad == 0 else f"ROPE MISMATCHES {bad}")
