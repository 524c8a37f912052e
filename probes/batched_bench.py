"""llama-batched-bench semantics (examples/batched-bench/batched-bench.cpp:150-175 of the reference) over the attention path:
for each repetition: clear the cache (alloc/free churn), prefill PP tokens for each of B sequences in ubatches of
n_batch tokens, then TG decode steps of B tokens; S_PP = B*PP/T_PP, S_TG = B*TG/T_TG.  Llama-3-8B geometry, all 32 layers,
random activations (the model side is out of scope).  BASELINE config 3: B=32, PP=512, TG=128."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _load_pkg
pkv = _load_pkg.load()
from llama_cpp_test_amd.runner import AttnPathRunner

ap = argparse.ArgumentParser()
ap.add_argument("--npl", type=int, default=32); ap.add_argument("--npp", type=int, default=512); ap.add_argument("--ntg", type=int, default=128)
ap.add_argument("--n-batch", type=int, default=2048); ap.add_argument("--kv-type", default="f16"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--per-layer", action="store_true", help="one call into the library per layer (r02) instead of pkv_attn_step_layers per ubatch")
ap.add_argument("--pp-f16", action="store_true", help="prompt ubatches hand Q over / take O back as f16 (PKV_ATTN_Q_F16 | PKV_ATTN_OUT_F16)")
ap.add_argument("--graph", action="store_true", help="AttnPathRunner.decode_captured: one cached hipGraph per ubatch shape instead of per-layer launches (measured: no faster, the GPU is the bound)")
a = ap.parse_args()
L, H, HK, D, BS = 32, 32, 8, 128, 128
t = pkv.TYPE_BY_NAME[a.kv_type]
ctx = a.npp + a.ntg
r = AttnPathRunner(n_layer=L, n_head=H, n_head_kv=HK, head_dim=D, type_k=t, type_v=t, n_blocks=a.npl * (-(-ctx // BS)),
                   n_seq_max=a.npl, max_batch_tokens=max(a.n_batch, a.npl), max_seq_len=-(-ctx // BS) * BS)
gen = torch.Generator(device="cuda").manual_seed(0)
nmax = max(a.n_batch, a.npl)
q = [torch.rand((nmax, H, D), generator=gen, device="cuda") * 2 - 1 for _ in range(L)]
k = [torch.rand((nmax, HK, D), generator=gen, device="cuda") * 2 - 1 for _ in range(L)]
v = [torch.rand((nmax, HK, D), generator=gen, device="cuda") * 2 - 1 for _ in range(L)]
out = [torch.empty((nmax, H, D), device="cuda") for _ in range(L)]
qp = [x.to(torch.float16) for x in q] if a.pp_f16 else q
outp = [torch.empty((nmax, H, D), device="cuda", dtype=torch.float16) for _ in range(L)] if a.pp_f16 else out

def run(seq_ids, poss, prompt=False):
    n = len(seq_ids)
    qq, oo = (qp, outp) if prompt else (q, out)
    if a.graph:
        r.decode_captured(seq_ids, poss, [x[:n] for x in qq], [x[:n] for x in k], [x[:n] for x in v], [x[:n] for x in oo])
        return
    plan = r.plan(seq_ids, poss)
    if a.per_layer:
        for il in range(L):
            r.layer(il, plan, qq[il][:n], k[il][:n], v[il][:n], oo[il][:n])
    else:
        r.layers(plan, [x[:n] for x in qq], [x[:n] for x in k], [x[:n] for x in v], [x[:n] for x in oo])

res = []
for rep in range(a.reps + 1):
    r.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = [(s, p) for s in range(a.npl) for p in range(a.npp)]          # sequence-major like batched-bench's prompt batch
    for i in range(0, len(toks), a.n_batch):
        chunk = toks[i:i + a.n_batch]
        run([s for s, _ in chunk], [p for _, p in chunk], prompt=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(a.ntg):
        run(list(range(a.npl)), [a.npp + i] * a.npl)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if rep:
        res.append((t1 - t0, t2 - t1))
tpp, ttg = np.median([x[0] for x in res]), np.median([x[1] for x in res])
print(f"{a.kv_type}{' (prompt Q / O f16)' if a.pp_f16 else ''} B={a.npl} PP={a.npp} TG={a.ntg} n_batch={a.n_batch}: T_PP {tpp*1e3:.1f} ms  S_PP {a.npl*a.npp/tpp:.0f} t/s | "
      f"T_TG {ttg*1e3:.1f} ms  S_TG {a.npl*a.ntg/ttg:.0f} t/s  ({ttg/a.ntg*1e3:.3f} ms/step, {'one cached hipGraph per ubatch shape' if a.graph else ('eager launches, one library call per layer' if a.per_layer else 'eager launches, one library call per ubatch')}, host block manager in the loop)")
