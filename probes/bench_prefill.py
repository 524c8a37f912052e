"""prefill attention throughput probe (not the driver's bench): pp512 x n_seq sequences, 8B geometry, one layer"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _load_pkg
pkv = _load_pkg.load()
H, HK, D, BS = (64, 8, 128, 128) if os.environ.get("PKV_PREFILL_GEOM") == "70b" else (32, 8, 128, 128)   # Llama-3-70B : 8B
for kv_type in (("f16",) if (os.environ.get("PKV_PREFILL_ONE") or os.environ.get("PKV_PREFILL_CASE")) else ("f16", "q8_0")):
    t = pkv.TYPE_BY_NAME[kv_type]
    CASES = ((32, 512, 0), (1, 512, 0), (4, 2048, 0), (1, 512, 3584), (1, 4096, 0), (1, 512, 15872))
    if os.environ.get("PKV_PREFILL_ONE"):
        CASES = CASES[:1]
    if os.environ.get("PKV_PREFILL_CASE"):
        CASES = [CASES[int(os.environ["PKV_PREFILL_CASE"])]]
    for n_seq, n_q, past in CASES:
        L = n_q + past
        bps = -(-L // BS)
        nb = n_seq * bps
        gen = torch.Generator(device="cuda").manual_seed(1)
        if t == pkv.F16:
            mk = lambda: ((torch.rand((nb, HK, BS * D), generator=gen, device="cuda") * 2 - 1).to(torch.float16).view(torch.uint8))
        else:
            mk = lambda: torch.cat([torch.randint(0, 256, (nb, HK, BS * D), generator=gen, device="cuda", dtype=torch.uint8),
                                    (torch.rand((nb, HK, BS * D // 32), generator=gen, device="cuda") * 0.01 + 0.001).to(torch.float16).view(torch.uint8)], dim=2).contiguous()
        kp, vp = mk(), mk()
        bt = torch.randperm(nb, generator=gen, device="cuda").to(torch.int32).reshape(n_seq, bps)
        sl = torch.full((n_seq,), L, dtype=torch.int32, device="cuda")
        qs = torch.arange(0, (n_seq + 1) * n_q, n_q, dtype=torch.int32, device="cuda")
        q = torch.rand((n_seq * n_q, H, D), generator=gen, device="cuda") * 2 - 1
        out = torch.empty_like(q)
        # allocated once (a per-call workspace costs a fill kernel per launch); sized for the kernel's split-KV mode
        ws = torch.empty(max(16, pkv.attn_prefill_workspace_bytes(n_seq * n_q, H, D)), dtype=torch.uint8, device="cuda")
        run = lambda: pkv.attn_prefill(q, kp, vp, bt, sl, qs, n_head_kv=HK, type_k=t, type_v=t, block_size=BS, scale=D ** -0.5,
                                       max_seq_len=L, max_q_per_seq=n_q, out=out, workspace=ws)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        vis = n_q * past + n_q * (n_q + 1) // 2
        flops = 4 * H * D * vis * n_seq
        print(f"{kv_type} n_seq {n_seq:3d} n_q {n_q:5d} past {past:5d}: {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s  {n_seq*n_q/ms*1e3:10.0f} tok/s/layer")
