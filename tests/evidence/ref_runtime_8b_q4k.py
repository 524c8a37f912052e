"""BASELINE config 2, end to end, inside the UNMODIFIED reference runtime (GPU box): Llama-3-8B geometry (32 layers, 4096 wide, 32 / 8
heads, head_dim 128, FFN 14336), Q4_K weights (synthetic GGUF, random well-formed blocks), f16 KV, ONE sequence at ctx 4096,
single-token decode.  libggml-mi355.so is registered the way GGML_BACKEND_PATH does it and n_gpu_layers = 99: weights, KV cache,
attention and the first-slice layer ops all run on the MI355X; the runtime, its graph and its scheduler are the reference's.
The 4096-cell context is not computed (the first-slice products are mat-vecs: a 4096-token prompt through them would take minutes)
but RESTORED: llama_state_seq_set_data with a session blob of 4096 random cells in the reference's own format — which also
exercises the plugin's set_tensor -> paged-mirror path at full size.  usage: python tests/evidence/ref_runtime_8b_q4k.py [--cpu N]
(--cpu N: the same on the reference's CPU backend with N threads, 8 decode steps)."""
import os, struct, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _load_pkg
_load_pkg.load()
from llama_cpp_test_amd import gguf_synth as G
from oracle import ref_model as R
import ctypes as C

NL, HK, D, CTX = 32, 8, 128, 4096
path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "llama3-8b-synth-q4k.gguf")
t0 = time.time()
G.synth_llama(path, n_layer=NL, n_embd=4096, n_head=32, n_head_kv=HK, n_ff=14336, n_vocab=32000, n_ctx=8192, rope_freq_base=500000.0,
              wtype=G.GGML_Q4_K, seed=8)
print(f"wrote {os.path.getsize(path) / 2**30:.2f} GiB of synthetic Q4_K weights in {time.time() - t0:.0f} s", flush=True)

# session blob: 4096 cells of sequence 0 (llama_kv_cache_unified::state_write format, single-sequence form: n_seq_id = 0 per cell)
rng = np.random.default_rng(1)
row = HK * D * 2
parts = [struct.pack("<I", CTX)] + [struct.pack("<iI", p, 0) for p in range(CTX)] + [struct.pack("<II", 0, NL)]
for kv in range(2):
    for il in range(NL):
        parts.append(struct.pack("<iQ", 1, row))
        parts.append((rng.standard_normal(CTX * HK * D, dtype=np.float32) * (0.5 if kv else 1.0)).astype(np.float16).tobytes())
blob = b"".join(parts)
print(f"session blob: {len(blob) / 2**20:.0f} MiB ({CTX} cells x {NL} layers, f16 K and V)", flush=True)

L = R._lib()
L.refmodel_detach_capture.restype = None
L.refmodel_state_seq_set.restype = C.c_size_t
cpu_threads = int(sys.argv[sys.argv.index("--cpu") + 1]) if "--cpu" in sys.argv else 0


def run(label, threads, steps):
    h = C.c_void_p(L.refmodel_load(path.encode(), CTX + 256, 512, 1, 1, 1, 1, threads))
    assert h
    L.refmodel_detach_capture(h)
    assert L.refmodel_state_seq_set(h, blob, C.c_size_t(len(blob)), 0) == len(blob)

    def step(tok, pos):
        a, p, s = np.array([tok], np.int32), np.array([pos], np.int32), np.zeros(1, np.int32)
        assert L.refmodel_decode(h, a.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), 1) == 0
    step(1, CTX)
    t0 = time.time()
    for i in range(steps):
        step(int(i * 13 % 32000), CTX + 1 + i)
    dt = time.time() - t0
    print(f"{label}: Llama-3-8B geometry, Q4_K weights, f16 KV, 1 seq at ctx {CTX}: {steps / dt:.2f} tok/s ({dt / steps * 1e3:.2f} ms/token)", flush=True)
    if "--pp" in sys.argv and threads == 16:
        # llama-bench's pp512 on the same model: ONE llama_decode of 512 tokens into an emptied cache (n_batch = n_ubatch = 512), median of 4
        L.refmodel_seq_rm.restype = None
        tok = (np.arange(512, dtype=np.int64) * 7 % 32000).astype(np.int32); pos = np.arange(512, dtype=np.int32); sid = np.zeros(512, np.int32)
        ts = []
        for r in range(5):
            L.refmodel_seq_rm(h, 0, -1, -1)
            t0 = time.time()
            assert L.refmodel_decode(h, tok.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), sid.ctypes.data_as(C.c_void_p), 512) == 0
            lg = np.empty(32000, np.float32)
            L.refmodel_logits(h, lg.ctypes.data_as(C.c_void_p), 32000)      # synchronises
            if r:
                ts.append(time.time() - t0)
        dtp = float(np.median(ts))
        print(f"{label}: Llama-3-8B geometry, Q4_K weights, pp512: {512 / dtp:.0f} tok/s ({dtp * 1e3:.1f} ms)", flush=True)
    L.refmodel_free(h)


if cpu_threads:
    run(f"reference CPU backend, {cpu_threads} threads", cpu_threads, 8)
R.load_backend(os.path.join(ROOT, "llama.cpp_test_amd", "libggml-mi355.so"))
R.set_gpu_layers(99)
run("reference runtime + libggml-mi355.so, all layers on the MI355X", 16, 32)
os.remove(path)
