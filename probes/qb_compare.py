"""prefill: 64-queries-per-wave kernel (PKV_PREFILL_QB=2) against the 32-query kernel on the same inputs (ragged lengths, pasts, q8_0)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _load_pkg
pkv = _load_pkg.load()
H, HK, D, BS = 32, 8, 128, 128
gen = torch.Generator(device="cuda").manual_seed(3)
for kv_type in os.environ.get("QB_TYPES", "f16,q8_0").split(","):
    t = pkv.TYPE_BY_NAME[kv_type]
    for lens, nqs in (([512] * 32, [512] * 32), ([4096], [4096]), ([1, 31, 33, 64, 65, 100, 700, 1000], [1, 31, 33, 64, 65, 100, 300, 129]), ([777] * 9, [70] * 9)):
        n_seq = len(lens); bps = [-(-l // BS) for l in lens]; nb = sum(bps)
        if t == pkv.F16:
            mk = lambda: ((torch.rand((nb, HK, BS * D), generator=gen, device="cuda") * 2 - 1).to(torch.float16).view(torch.uint8))
        else:
            mk = lambda: torch.cat([torch.randint(0, 256, (nb, HK, BS * D), generator=gen, device="cuda", dtype=torch.uint8),
                                    (torch.rand((nb, HK, BS * D // 32), generator=gen, device="cuda") * 0.01 + 0.001).to(torch.float16).view(torch.uint8)], dim=2).contiguous()
        kp, vp = mk(), mk()
        perm = torch.randperm(nb, generator=gen, device="cuda").to(torch.int32)
        bt = torch.zeros((n_seq, max(bps)), dtype=torch.int32, device="cuda")
        o = 0
        for i, b in enumerate(bps):
            bt[i, :b] = perm[o:o + b]; o += b
        sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
        qs = torch.tensor(np.concatenate([[0], np.cumsum(nqs)]), dtype=torch.int32, device="cuda")
        q = torch.rand((int(sum(nqs)), H, D), generator=gen, device="cuda") * 2 - 1
        outs = {}
        for qb in ("1", "2"):
            os.environ["PKV_PREFILL_QB"] = qb
            out = torch.full_like(q, float("nan"))
            pkv.attn_prefill(q, kp, vp, bt, sl, qs, n_head_kv=HK, type_k=t, type_v=t, block_size=BS, scale=D ** -0.5, max_seq_len=max(lens), max_q_per_seq=max(nqs), out=out)
            torch.cuda.synchronize()
            outs[qb] = out
        d = (outs["1"] - outs["2"]).abs()
        print(kv_type, "lens", lens[:4], "n_q", nqs[:4], "finite", bool(torch.isfinite(outs["2"]).all()), "max |qb1 - qb2|", float(d.max()), "equal", bool(torch.equal(outs["1"], outs["2"])))
