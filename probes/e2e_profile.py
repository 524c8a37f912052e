#!/usr/bin/env python3
"""probes/e2e_profile.py — the end-to-end decode step of bench.py's `e2e` record alone (for rocprofv3 --kernel-trace --stats):
python probes/e2e_profile.py [--steps N] [--eager]"""
import argparse, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _load_pkg
import bench
pkv = _load_pkg.load(); pkv.lib()
from llama_cpp_test_amd import gguf_synth as G
from llama_cpp_test_amd.decoder import LlamaDecoder
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--eager", action="store_true"); ap.add_argument("--ctx", type=int, default=4096); ap.add_argument("--model", default="8b", choices=["8b", "70b"], help="70b = BASELINE config 5 geometry (use --ctx 16384): 80 layers, 8192 wide, gqa 8, ~40 GB of weights"); ap.add_argument("--wtype", default="q4_K", choices=["q4_K", "q5_K"], help="base type of the tensor mix (Q4_K_M / Q5_K_M)"); ap.add_argument("--no-pack", action="store_true"); ap.add_argument("--fused-merge", action="store_true"); ap.add_argument("--image", action="store_true", help="opt-in: the attention merge launch leaves the q8_K image the wo product streams against (pkv_attn_args::out_image)")
a = ap.parse_args()
LLAMA3_70B = dict(n_layer=80, n_embd=8192, n_head=64, n_head_kv=8, n_ff=28672, n_vocab=128256, n_ctx=16384, rope_freq_base=500000.0)
geom = bench.LLAMA3_8B if a.model == "8b" else LLAMA3_70B
tmp = tempfile.mkdtemp(dir="/tmp", prefix="pkv_e2e_")
path = os.path.join(tmp, "m.gguf")
try:
    G.synth_llama(path, wtype=G.GGML_Q4_K if a.wtype == "q4_K" else G.GGML_Q5_K, k_mix=True, seed=8, **geom)
    print(f"synthetic GGUF written ({os.path.getsize(path) / 2**30:.1f} GiB)", flush=True)
    dec = LlamaDecoder(path, n_ctx=a.ctx, pack_weights=not a.no_pack)
    if a.fused_merge: dec.attn_flags = pkv.ATTN_FUSED_COMBINE
    if a.image: dec.use_attn_image()
    dec.fill_kv_synthetic(a.ctx - 1, seed=3)
    dec._place(12345, a.ctx - 1)
    if a.eager:
        run = lambda: dec.launches(max_seq_len=a.ctx)
    else:
        g = dec.capture(); run = g.replay
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    wb = dec.weight_bytes()
    print(f"Llama-3-{a.model.upper()} geometry, ctx {a.ctx}, {a.wtype}_M mix{' (GGUF-order rows)' if a.no_pack else ''}: {1 / dt:.1f} tok/s, {dt * 1e3:.3f} ms/token ({'eager' if a.eager else 'graph'}); weight bytes per token {wb / 1e9:.3f} GB -> {wb / dt / 1e12:.2f} TB/s of weights alone")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
