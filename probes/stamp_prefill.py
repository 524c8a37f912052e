"""diagnostic: per-workgroup phase times (shader clock cycles) of the prefill kernel, f16 D128 G4, pp512 x 32"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _load_pkg
pkv = _load_pkg.load()
H, HK, D, BS = 32, 8, 128, 128
n_seq, n_q = 32, 512
bps = n_q // BS; nb = n_seq * bps
gen = torch.Generator(device="cuda").manual_seed(1)
mk = lambda: ((torch.rand((nb, HK, BS * D), generator=gen, device="cuda") * 2 - 1).to(torch.float16).view(torch.uint8))
kp, vp = mk(), mk()
bt = torch.randperm(nb, generator=gen, device="cuda").to(torch.int32).reshape(n_seq, bps)
sl = torch.full((n_seq,), n_q, dtype=torch.int32, device="cuda")
qs = torch.arange(0, (n_seq + 1) * n_q, n_q, dtype=torch.int32, device="cuda")
q = torch.rand((n_seq * n_q, H, D), generator=gen, device="cuda") * 2 - 1
out = torch.empty_like(q)
run = lambda: pkv.attn_prefill(q, kp, vp, bt, sl, qs, n_head_kv=HK, type_k=1, type_v=1, block_size=BS, scale=D ** -0.5, max_seq_len=n_q, max_q_per_seq=n_q, out=out)
for _ in range(3): run()
torch.cuda.synchronize()
stamps = torch.zeros((n_seq * HK * 16, 8), dtype=torch.int64, device="cuda")
pkv.lib().pkv_debug_set_prefill_stamps(C.c_void_p(stamps.data_ptr()))
run(); torch.cuda.synchronize()
pkv.lib().pkv_debug_set_prefill_stamps(None)
s = stamps.cpu().numpy().reshape(n_seq, HK, 16, 8)
names = ["prologue(Q)", "loop", "epilogue", "stage", "qk", "softmax", "pv"]
for tile in (0, 3, 7, 15):
    v = s[:, :, tile, :7].reshape(-1, 7)
    print("tile", tile, "tiles/loop", (tile * 32 + 32 + 63) // 64, {n: int(np.median(v[:, i])) for i, n in enumerate(names)})
t0 = s[..., 7].min(); 
print("span cycles", int((s[..., 7] + s[..., 0] + s[..., 1] + s[..., 2]).max() - t0))
