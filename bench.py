#!/usr/bin/env python3
"""bench.py — decode through the paged-KV attention path at BASELINE.json's headline config.

Workload (config.workload): Llama-3-8B geometry (32 layers, 32 q heads, 8 kv heads, head_dim 128), ONE sequence at
ctx = 4096 in a paged cache (block_size 128, shuffled physical blocks), f16 KV by default.
A "step" = one decoded token through the hot path of all 32 layers: RoPE(Q) -> fused RoPE(K)+convert+scatter KV store
-> paged single-token attention (split-KV kernel + merge).  The non-attention ops of the decoder layer are outside
this path, so `value` is attention-path tok/s, not end-to-end model tok/s.
Inputs are synthetic, seeded, already resident in HBM; the 32 layers hold 512 MiB of distinct KV (> 2x the 256 MiB
Infinity Cache), so the dominant kernel streams from HBM, not from MALL.

    python bench.py                       # 1 GPU: the headline line + `configs` (BASELINE's other single-GPU configurations) + `e2e` (whole model)
    python bench.py --gpus N              # N independent replicas: this process starts the N ranks itself (child processes) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W     # ... or under an outside launcher (one block pool per GPU, no collective
                                                     on the data path; RCCL only carries the barrier / max-time)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "decode tok/s + KV-gather HBM GB/s (% of peak), Llama-3-8B Q4_K ctx=4096"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
GEOM = dict(n_layer=32, n_head=32, n_head_kv=8, head_dim=128, block_size=128)          # Llama-3-8B (the headline config)
GEOM_70B = dict(n_layer=80, n_head=64, n_head_kv=8, head_dim=128, block_size=128)     # Llama-3-70B (BASELINE config 5)


def algorithmic_bytes(n_kv: int, type_k: int, type_v: int, pkv, g=GEOM) -> int:
    """per sequence, per layer, per decode step"""
    n_embd_kv = g["n_head_kv"] * g["head_dim"]
    return (n_kv * (pkv.row_size(type_k, n_embd_kv) + pkv.row_size(type_v, n_embd_kv))
            + 4 * -(-n_kv // g["block_size"]) + 2 * g["n_head"] * g["head_dim"] * 4)


class Workload:
    """the device state of one replica: pools for every layer, block table, per-layer activations"""

    def __init__(self, pkv, torch, *, ctx, n_seq, type_k, type_v, seed, g=GEOM):
        self.pkv, self.torch, self.g = pkv, torch, g
        self.ctx, self.n_seq, self.tk, self.tv = ctx, n_seq, type_k, type_v
        L, H, HK, D, BS = g["n_layer"], g["n_head"], g["n_head_kv"], g["head_dim"], g["block_size"]
        dev = torch.device("cuda")
        gen = torch.Generator(device=dev).manual_seed(seed)
        bps = -(-ctx // BS)
        self.n_blocks = n_seq * bps
        # random K/V through the real store path is exercised by the tests; here the pools are filled directly with
        # random bytes of valid encodings: f16 values in (-1, 1) / random int8 quants with small positive f16 scales
        self.k_pools, self.v_pools = [], []
        for _ in range(L):
            self.k_pools.append(self._random_pool(type_k, gen))
            self.v_pools.append(self._random_pool(type_v, gen))
        perm = torch.randperm(self.n_blocks, generator=gen, device=dev).to(torch.int32)
        self.block_table = perm.reshape(n_seq, bps).contiguous()           # non-contiguous on purpose
        self.seq_lens = torch.full((n_seq,), ctx, dtype=torch.int32, device=dev)
        self.q = [torch.rand((n_seq, H, D), generator=gen, device=dev) * 2 - 1 for _ in range(L)]
        self.k_cur = [torch.rand((n_seq, HK, D), generator=gen, device=dev) * 2 - 1 for _ in range(L)]
        self.v_cur = [torch.rand((n_seq, HK, D), generator=gen, device=dev) * 2 - 1 for _ in range(L)]
        self.out = [torch.empty((n_seq, H, D), device=dev) for _ in range(L)]
        self.pos = torch.full((n_seq,), ctx - 1, dtype=torch.int32, device=dev)
        last = ctx - 1
        self.slot_block = self.block_table[:, last // BS].contiguous()
        self.slot_off = torch.full((n_seq,), last % BS, dtype=torch.int32, device=dev)
        self.rope = pkv.rope_params(n_dims=D)                              # llama-3: NORM pairs, freq_base 5e5
        self.ws = pkv.attn_workspace(n_seq, H, D, ctx, BS)
        self.scale = D ** -0.5
        self.step_flags = 0
        self.rope_cs = pkv.rope_cache(self.pos, self.rope, D)             # refreshed at the start of every step
        self.fused_step = True     # pkv_attn_decode_step: RoPE(Q) + RoPE(K) + KV store inside the decode launch

    def _random_pool(self, t, gen):
        torch, pkv, g = self.torch, self.pkv, self.g
        HK, D, BS = g["n_head_kv"], g["head_dim"], g["block_size"]
        nb = self.n_blocks
        if t == pkv.F16:
            x = (torch.rand((nb, HK, BS * D), generator=gen, device="cuda") * 2 - 1).to(torch.float16)
            return x.view(torch.uint8).reshape(nb, HK, -1)
        qb = BS * D if t == pkv.Q8_0 else BS * D // 2
        quants = torch.randint(0, 256, (nb, HK, qb), generator=gen, device="cuda", dtype=torch.uint8)
        scales = (torch.rand((nb, HK, BS * D // 32), generator=gen, device="cuda") * 0.01 + 0.001).to(torch.float16)
        return torch.cat([quants, scales.view(torch.uint8).reshape(nb, HK, -1)], dim=2).contiguous()

    def layer_step(self, l, flags=None, events=None):
        flags = self.step_flags if flags is None else flags
        pkv, g = self.pkv, self.g
        if self.fused_step:
            pkv.attn_decode_step(self.q[l], self.k_cur[l], self.v_cur[l], self.pos, self.rope, self.k_pools[l], self.v_pools[l],
                                 self.block_table, self.seq_lens, n_head_kv=g["n_head_kv"], type_k=self.tk, type_v=self.tv,
                                 block_size=g["block_size"], scale=self.scale, max_seq_len=self.ctx, out=self.out[l],
                                 workspace=self.ws, events=events, rope_cache=self.rope_cs, flags=flags)
            return
        pkv.qkv_prepare(self.q[l], self.k_cur[l], self.v_cur[l], self.k_pools[l], self.v_pools[l], type_k=self.tk,
                        type_v=self.tv, slot_block=self.slot_block, slot_off=self.slot_off, block_size=g["block_size"],
                        pos=self.pos, rope=self.rope)
        self.attn(l, flags, events)

    def attn(self, l, flags=0, events=None):
        pkv, g = self.pkv, self.g
        pkv.attn_decode(self.q[l], self.k_pools[l], self.v_pools[l], self.block_table, self.seq_lens,
                        n_head_kv=g["n_head_kv"], type_k=self.tk, type_v=self.tv, block_size=g["block_size"],
                        scale=self.scale, max_seq_len=self.ctx, out=self.out[l], workspace=self.ws, flags=flags, events=events)

    def step(self):
        if self.fused_step:   # ggml_rope_cache_init once per step: the (cos, sin) rows are the same for all layers
            self.pkv.rope_cache(self.pos, self.rope, self.g["head_dim"], out=self.rope_cs)
        for l in range(self.g["n_layer"]):
            self.layer_step(l)


def measure_dominant_kernel(w: Workload, torch, reps: int):
    """average duration of ONE launch of the dominant kernel — the paged decode kernel exactly as the timed step runs
    it (with the in-kernel merge when that is on; otherwise the split-KV kernel, whose merge is a separate small
    launch) — from HIP events bound to the launch itself (hipExtLaunchKernel start/stop events = the dispatch's GPU
    timestamps, on the stream the kernel runs on).  Layers are cycled so that every launch reads cold KV from HBM."""
    pkv, g = w.pkv, w.g
    L = g["n_layer"]
    flags = w.step_flags if w.step_flags & pkv.ATTN_FUSED_COMBINE else 0
    timer = pkv.KernelTimer(reps * L)

    def launch(l, ev=None):
        if w.fused_step:
            w.layer_step(l, events=ev)      # the decode-step kernel (+ its merge launch, which the events do not cover)
        else:
            w.attn(l, flags, ev)
    for l in range(L):
        launch(l)
    torch.cuda.synchronize()
    # Python enqueues slower than the GPU drains; park the stream behind a spin kernel while the launches queue up so
    # that they then execute back to back — the regime of the timed (graph-replayed) step — instead of from an idle GPU
    torch.cuda._sleep(int(reps * L * 6e4) + 20_000_000)
    for i in range(reps * L):
        launch(i % L, timer.pair(i))
    torch.cuda.synchronize()
    ms = np.array(timer.elapsed_ms())
    timer.close()
    return float(np.mean(ms)) * 1e-3, float(np.median(ms)) * 1e-3, float(np.min(ms)) * 1e-3


def measure_dominant_kernel_graph(w: Workload, torch, stream, reps: int):
    """the same kernel as it runs in the timed step: a hipGraph of n_layer launches of the dominant kernel ALONE (partials
    left in the workspace, PKV_ATTN_NO_COMBINE: no merge launch), replayed back to back between two HIP events recorded on the
    stream the graph runs on.  (elapsed / launches) includes the ~0.4 us dispatch gap between consecutive kernels of a graph
    and none of the event-bound launch's own overhead; rocprofv3's per-kernel duration sits just below it."""
    pkv, L = w.pkv, w.g["n_layer"]
    with torch.cuda.stream(stream):
        def only_kernel():
            for l in range(L):
                if w.fused_step:
                    w.layer_step(l, flags=pkv.ATTN_NO_COMBINE)
                else:
                    w.attn(l, pkv.ATTN_NO_COMBINE)
        only_kernel()
        torch.cuda.synchronize()
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gk, stream=stream):
            only_kernel()
        for _ in range(3):
            gk.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            gk.replay()
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * L)


def cpu_baseline(w: Workload, pkv, budget_s: float = 20.0):
    """the reference's CPU attention (oracle/_ref when it was shipped prebuilt, else our C restatement) on the host cores,
    on a bounded sample: whole layers of the same workload until ~budget_s of CPU time; tok/s = 1 / (t_layer * n_layer).
    CHECKER ONLY: nothing here feeds the GPU result."""
    from oracle import cpu_oracle as O
    g = w.g
    H, HK, D, BS = g["n_head"], g["n_head_kv"], g["head_dim"], g["block_size"]
    cores = min(os.cpu_count() or 1, H)   # the reference parallelises over Q rows: at batch 1 at most n_head threads help
    kind = "reference" if O.ref_available() else "port"
    bt = w.block_table[0].cpu().numpy()
    t_layers, n_done = [], 0
    t0 = time.time()
    while n_done < g["n_layer"] and (time.time() - t0) < budget_s:
        l = n_done
        kp = w.k_pools[l].cpu().numpy()
        vp = w.v_pools[l].cpu().numpy()
        kc = O.pool_gather(kp, w.tk, 0, bt, w.ctx, D=D, block_size=BS)     # de-paged, reference row format
        vc = O.pool_gather(vp, w.tv, 1, bt, w.ctx, D=D, block_size=BS)
        q = w.q[l][0].cpu().numpy()[:, None, :]
        fn = O.ref_flash_attn_ext if kind == "reference" else O.flash_attn_ext
        ta = time.time()
        fn(q, kc, vc, None, type_k=w.tk, type_v=w.tv, scale=w.scale, n_threads=cores)
        t_layers.append(time.time() - ta)
        n_done += 1
    t_layer = float(np.median(t_layers))
    return {"value": round(1.0 / (t_layer * g["n_layer"]), 3), "unit": "tok/s", "cores": cores, "kind": kind,
            "sample": f"flash_attn_ext of {n_done} of {g['n_layer']} layers (1 seq, ctx {w.ctx}, {cores} threads), "
                      f"median {t_layer * 1e3:.2f} ms/layer, extrapolated to {g['n_layer']} layers; KV store + RoPE excluded"}


LLAMA3_8B = dict(n_layer=32, n_embd=4096, n_head=32, n_head_kv=8, n_ff=14336, n_vocab=128256, n_ctx=8192, rope_freq_base=500000.0)


def e2e_record(pkv, torch, *, ctx: int, steps: int, warmup: int, cpu: bool, reps: int = 5):
    """The metric's other half: a FULL decode step of Llama-3-8B geometry (Q4_K_M tensor-type mix: q4_K with q6_K output / attn_v / ffn_down
    of the use_more_bits layers; synthetic, seeded, well-formed blocks), ONE sequence at `ctx`, run by the repo's own host loop
    (llama.cpp_test_amd/decoder.py: own GGUF loader -> pkv_get_rows / pkv_rms_norm_mul_q / pkv_gemv_q / pkv_attn_decode_step through the C ABI),
    captured as ONE hipGraph per token and replayed: no reference code in the timed region.  roofline = (weight bytes one step reads +
    algorithmic KV bytes) / time / 8 TB/s.  cpu_baseline = the reference's own llama_decode on its CPU backend (oracle/_ref, CHECKER ONLY) on
    the same file at the same depth, context restored from a session blob in the reference's format."""
    import shutil
    import tempfile
    from llama_cpp_test_amd import gguf_synth as G
    from llama_cpp_test_amd.decoder import LlamaDecoder
    tmp = tempfile.mkdtemp(dir="/tmp", prefix="pkv_e2e_")
    path = os.path.join(tmp, "llama3-8b-synth-q4km.gguf")
    rec = {}
    try:
        t0 = time.time()
        G.synth_llama(path, wtype=G.GGML_Q4_K, k_mix=True, seed=8, **LLAMA3_8B)
        t_gen = time.time() - t0
        t0 = time.time()
        dec = LlamaDecoder(path, n_ctx=ctx, type_k=pkv.F16, type_v=pkv.F16)
        dec.fill_kv_synthetic(ctx - 1, seed=3)
        dec._place(12345 % LLAMA3_8B["n_vocab"], ctx - 1)            # the token that makes the sequence `ctx` long: every replay decodes it again
        t_load = time.time() - t0
        side = torch.cuda.Stream()
        g = dec.capture(side)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            rep_tok_s = []
            for _ in range(reps):                                     # llama-bench: repetitions for mean +- stdev
                torch.cuda.synchronize()
                r0 = time.perf_counter()
                for _ in range(steps):
                    g.replay()
                torch.cuda.synchronize()
                rep_tok_s.append(steps / (time.perf_counter() - r0))
        wb = dec.weight_bytes()
        kvb = LLAMA3_8B["n_layer"] * algorithmic_bytes(ctx, pkv.F16, pkv.F16, pkv, GEOM)
        finite = bool(torch.isfinite(dec.logits).all().item())
        n_types = {}
        for name, (t, ty, ne) in dec.w.items():
            n_types[ty & ~pkv.PACKED] = n_types.get(ty & ~pkv.PACKED, 0) + t.numel()
        rec = {"value": round(1.0 / dt, 2), "unit": "tok/s", "ms_per_token": round(dt * 1e3, 4), "steps": steps, "warmup": warmup,
               "repetitions": {"n": reps, "mean_tok_s": round(float(np.mean(rep_tok_s)), 2), "stdev_tok_s": round(float(np.std(rep_tok_s, ddof=1)), 2)} if reps > 1 else None,
               "workload": f"Llama-3-8B geometry, Q4_K_M tensor mix (synthetic GGUF {os.path.getsize(path) / 2**30:.2f} GiB: "
                           + ", ".join(f"{ {0: 'f32', 12: 'q4_K', 14: 'q6_K'}.get(k, k)} {v / 2**20:.0f} MiB" for k, v in sorted(n_types.items()))
                           + f"), f16 KV, 1 sequence at ctx {ctx}, single-token decode incl. lm_head",
               "host_loop": "llama.cpp_test_amd/decoder.py through the C ABI, one hipGraph per token (own GGUF loader; no reference code)",
               "weight_layout": "K-quant rows re-laid inside their own bytes at upload (plane layout, csrc/pkv_wlayout.h; pkv_weights_pack)",
               "logits_finite": finite, "weight_bytes": wb, "kv_bytes": kvb, "setup_s": {"synth_gguf": round(t_gen, 1), "load_fill_capture": round(t_load, 1)},
               "roofline": {"bound": "hbm", "achieved": round((wb + kvb) / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round((wb + kvb) / dt / 1e9 / HBM_PEAK_GBS, 4)}}
        dec.close()
        del dec
        torch.cuda.empty_cache()
        if cpu:
            rec["cpu_baseline"] = e2e_cpu_baseline(path, ctx)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return rec


def e2e_cpu_baseline(path: str, ctx: int, steps: int = 6):
    """CHECKER ONLY (oracle/_ref = the reference compiled in place): the reference's llama_decode on its CPU backend, same GGUF, same depth;
    the ctx - 1 cells of context are restored with llama_state_seq_set_data from a session blob in the reference's own format"""
    import ctypes as C
    import struct
    from oracle import ref_model as R
    if not R.available():
        return None
    NL, HK, D = LLAMA3_8B["n_layer"], LLAMA3_8B["n_head_kv"], 128
    n_cells = ctx - 1
    rng = np.random.default_rng(1)
    row = HK * D * 2
    parts = [struct.pack("<I", n_cells)] + [struct.pack("<iI", p, 0) for p in range(n_cells)] + [struct.pack("<II", 0, NL)]
    for kv in range(2):
        for il in range(NL):
            parts.append(struct.pack("<iQ", 1, row))
            parts.append((rng.standard_normal(n_cells * HK * D, dtype=np.float32) * (0.5 if kv else 1.0)).astype(np.float16).tobytes())
    blob = b"".join(parts)
    L = R._lib()
    L.refmodel_detach_capture.restype = None
    L.refmodel_state_seq_set.restype = C.c_size_t
    cores = min(os.cpu_count() or 1, 32)
    h = C.c_void_p(L.refmodel_load(path.encode(), ctx + 64, 512, 1, 1, 1, 1, cores))
    if not h:
        return None
    L.refmodel_detach_capture(h)
    if L.refmodel_state_seq_set(h, blob, C.c_size_t(len(blob)), 0) != len(blob):
        L.refmodel_free(h)
        return None

    def step(tok, pos):
        a, p, s = np.array([tok], np.int32), np.array([pos], np.int32), np.zeros(1, np.int32)
        return L.refmodel_decode(h, a.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), 1)
    step(1, n_cells)
    t0 = time.time()
    for i in range(steps):
        step(int(i * 13 % 32000), n_cells + 1 + i)
    dt = (time.time() - t0) / steps
    L.refmodel_free(h)
    return {"value": round(1.0 / dt, 2), "unit": "tok/s", "cores": cores, "kind": "reference",
            "sample": f"{steps} llama_decode steps of the reference's CPU backend ({cores} threads) on the same GGUF at ctx {ctx} (context restored from a session blob), {dt * 1e3:.1f} ms/token"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--n-seq", type=int, default=1, help="sequences per GPU (1 = BASELINE config 2)")
    ap.add_argument("--kv-type", default="f16", choices=["f16", "q8_0", "q4_0"])
    ap.add_argument("--model", default="8b", choices=["8b", "70b"], help="geometry: 8b = headline config, 70b = BASELINE config 5 (use --ctx 16384)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-merge", action="store_true",
                    help="merge the split-KV partials inside the decode launch (last-arriving workgroup) instead of a second "
                         "launch; measured slower on MI355X (12.1 us vs 6.9 + 4.7 us), kept for comparison")
    ap.add_argument("--two-call", action="store_true",
                    help="pkv_qkv_prepare + pkv_attn_decode (three launches per layer) instead of the fused pkv_attn_decode_step")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end record (full Llama-3-8B Q4_K_M decode step through the own host loop)")
    ap.add_argument("--e2e-steps", type=int, default=30)
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` sub-record (short runs of BASELINE's other single-GPU configurations)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions behind the timed steps for mean +- stdev (llama-bench: 1 warm-up + 5 repetitions)")
    return ap.parse_args(argv)


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` without an outside launcher: start the N ranks as CHILD processes (`python -m torch.distributed.run`, one per GPU),
    relay rank 0's JSON line, exit non-zero if a rank fails.  Runs before anything in this process has touched torch or the GPU (a process that has
    initialised HIP must never be replaced or forked into ranks)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:])
        sys.stderr.write(f"\nbench.py: the {n}-rank launch failed (exit code {p.returncode}, JSON line {'missing' if line is None else 'present'})\n")
        return p.returncode or 1
    print(line, flush=True)
    return 0


def rehearsal_rank(args) -> int:
    """CPU REHEARSAL of the N-rank protocol (tests only, PKV_BENCH_CPU_REHEARSAL=1; never a measurement and the line says so): each rank drives its own
    bookkeeping-only block manager (find_slot / seq_rm churn of its sequences) where a GPU rank replays its step, then the same barrier, max-over-ranks
    and whole-job aggregation, over gloo."""
    import torch
    import torch.distributed as dist
    import _load_pkg
    pkv = _load_pkg.load()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist.init_process_group("gloo")
    if os.environ.get("PKV_BENCH_REHEARSAL_FAIL_RANK") == str(rank):      # tests: a rank that dies must fail the whole launch
        raise SystemExit(3)
    c = pkv.PagedKVCache(n_blocks=args.n_seq * 4, block_size=128, n_seq_max=args.n_seq, n_layer=0)

    def step(i):
        c.find_slot(list(range(args.n_seq)), [i % 256] * args.n_seq)
        if i % 256 == 255:
            c.clear()
    for i in range(args.warmup):
        step(i)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(world * args.n_seq * args.steps / dt, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "REHEARSAL on CPU (PKV_BENCH_CPU_REHEARSAL): host block-manager traffic only, no kernel ran - not a measurement",
                          "config": {"workload": "rank protocol rehearsal", "parallelism": f"replicas x{world} (no collective on the data path)"}}), flush=True)
    dist.destroy_process_group()
    return 0


def timed_config(pkv, torch, *, model, ctx, n_seq, kv_type, steps, warmup, reps, kernel_reps, dist=None, shared=False, rank=0, world=1, no_graph=False, fused_merge=False,
                 two_call=False, keep=False):
    """one configuration of the attention path: W warm-up steps, K timed steps between barriers (max over ranks), then `reps` repetitions of K steps for
    mean +- stdev (llama-bench's 1 warm-up + 5 repetitions, /root/reference/examples/llama-bench/llama-bench.cpp:1604-1641), the dominant kernel's own duration and its roofline"""
    t = pkv.TYPE_BY_NAME[kv_type]
    g = GEOM if model == "8b" else GEOM_70B
    w = Workload(pkv, torch, ctx=ctx, n_seq=n_seq, type_k=t, type_v=t, seed=1234 + rank, g=g)
    w.step_flags = pkv.ATTN_FUSED_COMBINE if fused_merge else 0
    w.fused_step = not two_call
    side = torch.cuda.Stream()
    graph = None
    with torch.cuda.stream(side):
        w.step()
        torch.cuda.synchronize()
        if not no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                w.step()
        run = graph.replay if graph is not None else w.step

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        rep_tok_s = []
        for _ in range(reps):
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            for _ in range(steps):
                run()
            torch.cuda.synchronize()
            rep_tok_s.append(n_seq * steps / (time.perf_counter() - r0))
        k_mean, k_med, k_min = measure_dominant_kernel(w, torch, kernel_reps)
    k_graph = None
    if not fused_merge and not no_graph:
        k_graph = measure_dominant_kernel_graph(w, torch, side, max(5, kernel_reps))
    bytes_launch = n_seq * algorithmic_bytes(ctx, t, t, pkv, g)
    k_used = k_graph if k_graph is not None else k_mean
    achieved = bytes_launch / k_used / 1e9
    rec = {"tok_s": world * n_seq * steps / dt, "dt": dt, "graph": graph is not None, "w": w if keep else None, "g": g, "t": t,
           "reps": {"n": reps, "mean_tok_s": round(float(np.mean(rep_tok_s)), 2), "stdev_tok_s": round(float(np.std(rep_tok_s, ddof=1)), 2)} if reps > 1 else None,
           "bytes_launch": bytes_launch, "achieved": achieved,
           "kernel_us": {"mean": round(k_used * 1e6, 3), "event_bound_mean": round(k_mean * 1e6, 3), "event_bound_median": round(k_med * 1e6, 3), "event_bound_min": round(k_min * 1e6, 3)},
           "k_graph": k_graph is not None}
    tiles = 1 if fused_merge else pkv.attn_decode_tiles(ctx, g["n_head"], g["n_head_kv"], n_seq)
    rec["tiles_per_workgroup"] = tiles
    rec["merge"] = ("in-kernel (last-arriving workgroup)" if fused_merge else
                    "none: one workgroup per (sequence, kv head) walks the whole sequence and writes the output" if (tiles > 1 and tiles >= -(-ctx // 128)) else "second launch")
    if not keep:
        del w
        torch.cuda.empty_cache()
    return rec


def configs_record(pkv, torch, *, reps, kernel_reps):
    """BASELINE.json's other single-GPU configurations as short runs under the same clock (value of the headline line untouched): 32 sequences x ctx 4096 (f16 and q8_0
    KV), 32 x ctx 640 (the decode step at the end of config 3), Llama-3-70B geometry at ctx 16384 (config 5), and config 3 itself with batched-bench semantics"""
    out = {}
    for name, kw in (("8b_32seq_ctx4096_f16", dict(model="8b", ctx=4096, n_seq=32, kv_type="f16", steps=20, warmup=3)),
                     ("8b_32seq_ctx4096_q8_0", dict(model="8b", ctx=4096, n_seq=32, kv_type="q8_0", steps=20, warmup=3)),
                     ("8b_32seq_ctx640_f16", dict(model="8b", ctx=640, n_seq=32, kv_type="f16", steps=50, warmup=5)),
                     ("70b_1seq_ctx16384_f16", dict(model="70b", ctx=16384, n_seq=1, kv_type="f16", steps=20, warmup=3))):
        try:
            r = timed_config(pkv, torch, reps=reps, kernel_reps=max(3, kernel_reps // 4), **kw)
            out[name] = {"value": round(r["tok_s"], 2), "unit": "tok/s", "ms_per_step": round(r["dt"] / kw["steps"] * 1e3, 5), "steps": kw["steps"], "repetitions": r["reps"],
                         "roofline": {"bound": "hbm", "achieved": round(r["achieved"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(r["achieved"] / HBM_PEAK_GBS, 4),
                                      "algorithmic_bytes_per_launch": r["bytes_launch"], "kernel_us": r["kernel_us"]},
                         "tiles_per_workgroup": r["tiles_per_workgroup"], "merge": r["merge"]}
        except Exception as e:   # one configuration must not take the line down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["8b_batched_bench_32x_pp512_tg128_f16"] = batched_bench_record(pkv, torch, npl=32, npp=512, ntg=128, n_batch=2048, kv_type="f16", reps=reps)
        h = batched_bench_record(pkv, torch, npl=32, npp=512, ntg=128, n_batch=2048, kv_type="f16", reps=reps, pp_f16=True)
        out["8b_batched_bench_32x_pp512_tg128_f16"].update({"S_PP_f16_io": h["S_PP"], "S_PP_f16_io_stdev": h["S_PP_stdev"],
                                                            "S_PP_f16_io_note": "same run with the prompt ubatches' Q handed over / O taken back as f16 (PKV_ATTN_Q_F16 | PKV_ATTN_OUT_F16)"})
        k8 = batched_bench_record(pkv, torch, npl=32, npp=512, ntg=128, n_batch=2048, kv_type="q8_0", reps=reps)
        out["8b_batched_bench_32x_pp512_tg128_f16"].update({"q8_0_kv": {k: k8[k] for k in ("S_PP", "S_PP_stdev", "S_TG", "S_TG_stdev")}})
    except Exception as e:
        out["8b_batched_bench_32x_pp512_tg128_f16"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def batched_bench_record(pkv, torch, *, npl, npp, ntg, n_batch, kv_type, reps, pp_f16=False):
    """llama-batched-bench semantics over the attention path (/root/reference/examples/batched-bench/batched-bench.cpp:150-175): per repetition clear the cache (block alloc /
    free churn), prefill PP tokens for each of B sequences in ubatches of n_batch tokens, then TG decode steps of B tokens, host block manager in the loop, eager launches;
    S_PP = B PP / T_PP, S_TG = B TG / T_TG; 1 warm-up + `reps` repetitions, mean +- stdev"""
    from llama_cpp_test_amd.runner import AttnPathRunner
    L, H, HK, D, BS = GEOM["n_layer"], GEOM["n_head"], GEOM["n_head_kv"], GEOM["head_dim"], GEOM["block_size"]
    t = pkv.TYPE_BY_NAME[kv_type]
    ctx = npp + ntg
    r = AttnPathRunner(n_layer=L, n_head=H, n_head_kv=HK, head_dim=D, type_k=t, type_v=t, n_blocks=npl * (-(-ctx // BS)), n_seq_max=npl, max_batch_tokens=max(n_batch, npl),
                       max_seq_len=-(-ctx // BS) * BS)
    gen = torch.Generator(device="cuda").manual_seed(0)
    nmax = max(n_batch, npl)
    q = [torch.rand((nmax, H, D), generator=gen, device="cuda") * 2 - 1 for _ in range(L)]
    k = [torch.rand((nmax, HK, D), generator=gen, device="cuda") * 2 - 1 for _ in range(L)]
    v = [torch.rand((nmax, HK, D), generator=gen, device="cuda") * 2 - 1 for _ in range(L)]
    out = [torch.empty((nmax, H, D), device="cuda") for _ in range(L)]
    # pp_f16: the prompt ubatches hand Q over and take O back as f16 (PKV_ATTN_Q_F16 | PKV_ATTN_OUT_F16, the own runner's boundary; decode steps stay f32)
    qp = [x.to(torch.float16) for x in q] if pp_f16 else q
    outp = [torch.empty((nmax, H, D), device="cuda", dtype=torch.float16) for _ in range(L)] if pp_f16 else out

    def run(seq_ids, poss, prompt=False):
        n = len(seq_ids)
        plan = r.plan(seq_ids, poss)
        qq, oo = (qp, outp) if prompt else (q, out)
        r.layers(plan, [x[:n] for x in qq], [x[:n] for x in k], [x[:n] for x in v], [x[:n] for x in oo])   # all layers in one trip through the binding (pkv_attn_step_layers)
    spp, stg = [], []
    for rep in range(reps + 1):
        r.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = [(s, p) for s in range(npl) for p in range(npp)]              # sequence-major like batched-bench's prompt batch
        for i in range(0, len(toks), n_batch):
            chunk = toks[i:i + n_batch]
            run([s for s, _ in chunk], [p for _, p in chunk], prompt=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(ntg):
            run(list(range(npl)), [npp + i] * npl)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep:
            spp.append(npl * npp / (t1 - t0))
            stg.append(npl * ntg / (t2 - t1))
    r.close() if hasattr(r, "close") else None
    del q, k, v, out, qp, outp
    torch.cuda.empty_cache()
    sd = lambda a: round(float(np.std(a, ddof=1)), 1) if len(a) > 1 else 0.0   # noqa: E731
    return {"S_PP": round(float(np.mean(spp)), 1), "S_PP_stdev": sd(spp), "S_TG": round(float(np.mean(stg)), 1), "S_TG_stdev": sd(stg), "unit": "tok/s", "repetitions": reps,
            "workload": f"batched-bench semantics, B={npl} PP={npp} TG={ntg} n_batch={n_batch}, {kv_type} KV, attention path of all {L} layers, host block manager in the loop, eager launches (one call into the library per ubatch: pkv_attn_step_layers)"
                        + (", prompt ubatches with f16 Q in / f16 O out" if pp_f16 else "")}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    # N > 1 started plainly: this process only launches the ranks (before torch / the library / any HIP call) and relays rank 0's line
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, argv)
    if os.environ.get("PKV_BENCH_CPU_REHEARSAL"):
        return rehearsal_rank(args)

    import torch
    import _load_pkg
    pkv = _load_pkg.load()
    pkv.lib()  # fail loudly if the HIP extension is missing

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU (the paged-KV path has no CPU fallback)")
    shared = world > n_dev                      # rehearsal only: more ranks than GPUs on this box -> ranks share a device
    torch.cuda.set_device(local % n_dev)
    dist = None
    if world > 1 or os.environ.get("PKV_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        # one process per GPU; RCCL ("nccl") carries only the barrier and the max-time reduction — the data path has no
        # collective.  Ranks that share a device (rehearsal) cannot form an RCCL communicator: gloo then.
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % n_dev))

    # one decode step = 32 layers x (fused RoPE(Q)+RoPE(K)+KV store, split-KV attention, merge); captured once, replayed per step
    r = timed_config(pkv, torch, model=args.model, ctx=args.ctx, n_seq=args.n_seq, kv_type=args.kv_type, steps=args.steps, warmup=args.warmup, reps=args.reps if world == 1 else 0,
                     kernel_reps=args.kernel_reps, dist=dist, shared=shared, rank=rank, world=world, no_graph=args.no_graph, fused_merge=args.fused_merge, two_call=args.two_call, keep=True)
    w, g, t, dt = r["w"], r["g"], r["t"], r["dt"]

    traffic = None
    try:   # PMC traffic cannot be collected from inside the run: taken from the committed rocprofv3 passes, if this workload has one AND the passes
        # were taken of THIS kernel source (profiles/traffic.json carries its hash; a changed kernel reports null until the passes are redone)
        import hashlib
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        src = hashlib.sha256(open(os.path.join(ROOT, "llama.cpp_test_amd", "csrc", "pkv_attn_decode.hip"), "rb").read()).hexdigest()
        if args.model == "8b" and tj.get("kernel_source_sha256") == src:
            traffic = tj.get(f"{args.kv_type}_ctx{args.ctx}_nseq{args.n_seq}")
    except Exception:
        pass
    line = {
        "metric": METRIC, "value": round(r["tok_s"], 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.kv_type if args.kv_type != "f16" else "f16",
        "data": "synthetic",
        "config": {"workload": f"Llama-3-{args.model.upper()} geometry, {args.n_seq} seq/GPU, ctx {args.ctx} paged decode "
                               f"(block_size 128, {args.kv_type} KV), attention path of all {g['n_layer']} layers per token",
                   "n_layer": g["n_layer"], "n_head": g["n_head"], "n_head_kv": g["n_head_kv"], "head_dim": 128, "block_size": 128, "ctx": args.ctx,
                   "kv_type": args.kv_type, "seqs_per_gpu": args.n_seq, "launch": "hipGraph" if r["graph"] else "eager", "merge": r["merge"], "tiles_per_workgroup": r["tiles_per_workgroup"],
                   "step": "pkv_attn_decode_step (RoPE + KV store inside the decode launch) + merge" if w.fused_step else "pkv_qkv_prepare + pkv_attn_decode",
                   "parallelism": f"replicas x{world} (no collective on the data path)"},
        "repetitions": r["reps"],
        "roofline": {"bound": "hbm", "achieved": round(r["achieved"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(r["achieved"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" if traffic else None,
                     "kernel": "pkv::attn_decode_kernel (split-KV paged decode" + (", RoPE(Q,K) + KV store of the new token fused in" if w.fused_step else "") + "; one launch per layer)",
                     "algorithmic_bytes_per_launch": r["bytes_launch"],
                     "kernel_us": r["kernel_us"],
                     "timing": ("mean = HIP events around back-to-back replays of a hipGraph holding ONLY the n_layer launches of this kernel "
                                "(elapsed / launches; layers cycled: cold KV); event_bound_* = hipExtLaunchKernel start/stop events of single "
                                "launches (the event-bound dispatch itself is ~1 us slower)") if r["k_graph"] else
                               "hipExtLaunchKernel start/stop events of each launch (GPU timestamps of the dispatch), layers cycled (cold KV)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(w, pkv)
    del w
    r["w"] = None
    torch.cuda.empty_cache()                                          # the attention-path pools make room for the other records
    headline = args.model == "8b" and args.n_seq == 1 and args.kv_type == "f16"
    if rank == 0 and world == 1 and not args.no_configs and headline:
        line["configs"] = configs_record(pkv, torch, reps=args.reps, kernel_reps=args.kernel_reps)
    if rank == 0 and world == 1 and not args.no_e2e and headline:
        try:
            line["e2e"] = e2e_record(pkv, torch, ctx=args.ctx, steps=args.e2e_steps, warmup=3, cpu=not args.no_cpu_baseline, reps=args.reps)
        except Exception as e:                                        # the headline record must not be lost to a failure of the extra one
            line["e2e"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
