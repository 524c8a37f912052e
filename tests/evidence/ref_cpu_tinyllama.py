"""Sanity anchor (dev container only): BASELINE config 1 on the REFERENCE itself — TinyLlama-1.1B geometry, Q8_0
weights (synthetic random GGUF written by llama.cpp_test_amd/gguf_synth.py), tg128 from an empty cache on the reference's CPU
backend (oracle/_ref/libref_kv.so, 8 threads of this container).  llama-bench semantics: 1 warm-up token, then 128 timed
single-token llama_decode calls.  usage: python tests/evidence/ref_cpu_tinyllama.py [n_threads] [--plugin]
--plugin (GPU box): the same loop with our ggml-backend plugin registered and n_gpu_layers = 99: every layer (weights, KV cache,
attention, the first-slice layer ops) on the MI355X, the reference runtime unmodified; CPU-only and plugin runs back to back."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _load_pkg
_load_pkg.load()
from llama_cpp_test_amd import gguf_synth as G
from oracle import ref_model as R
import ctypes as C

args = [a for a in sys.argv[1:] if not a.startswith("--")]
nth = int(args[0]) if args else 8
plugin = "--plugin" in sys.argv
path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "tinyllama-synth-q8_0.gguf")
t0 = time.time()
G.synth_llama(path, n_layer=22, n_embd=2048, n_head=32, n_head_kv=4, n_ff=5632, n_vocab=32000, n_ctx=2048, rope_freq_base=10000.0,
              wtype=G.GGML_Q8_0, seed=1)
print(f"wrote {os.path.getsize(path) / 2**20:.0f} MiB in {time.time() - t0:.0f} s", flush=True)
L = R._lib()
L.refmodel_detach_capture.restype = None


def tg128(label):
    h = C.c_void_p(L.refmodel_load(path.encode(), 512, 512, 1, 1, 1, 1, nth))   # flash_attn on, f16 KV
    assert h
    L.refmodel_detach_capture(h)                                                # plain decode, no per-node callbacks

    def step(tok, pos):
        a = np.array([tok], np.int32); p = np.array([pos], np.int32); s = np.zeros(1, np.int32)
        assert L.refmodel_decode(h, a.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), 1) == 0
    step(1, 0)
    t0 = time.time()
    for i in range(128):
        step(int(i * 7 % 32000), i + 1)
    dt = time.time() - t0
    logits = np.empty(32000, np.float32)
    L.refmodel_logits(h, logits.ctypes.data_as(C.c_void_p), 32000)
    print(f"{label}, TinyLlama-1.1B geometry Q8_0, tg128: {128 / dt:.2f} tok/s ({dt / 128 * 1e3:.2f} ms/token)", flush=True)
    L.refmodel_free(h)
    return logits


def pp512(label, reps):
    """llama-bench's pp512: ONE llama_decode of a 512-token prompt (n_batch = n_ubatch = 512) into an empty cache, median of `reps`"""
    h = C.c_void_p(L.refmodel_load(path.encode(), 512, 512, 1, 1, 1, 1, nth))
    assert h
    L.refmodel_detach_capture(h)
    tok = (np.arange(512, dtype=np.int64) * 7 % 32000).astype(np.int32); pos = np.arange(512, dtype=np.int32); sid = np.zeros(512, np.int32)
    ts = []
    for r in range(reps + 1):
        L.refmodel_seq_rm(h, 0, -1, -1)
        t0 = time.time()
        assert L.refmodel_decode(h, tok.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), sid.ctypes.data_as(C.c_void_p), 512) == 0
        logits = np.empty(32000, np.float32)
        L.refmodel_logits(h, logits.ctypes.data_as(C.c_void_p), 32000)      # synchronises
        if r:
            ts.append(time.time() - t0)
    dt = float(np.median(ts))
    print(f"{label}, TinyLlama-1.1B geometry Q8_0, pp512: {512 / dt:.0f} tok/s ({dt * 1e3:.1f} ms)", flush=True)
    L.refmodel_free(h)
    return logits


cpu = tg128(f"reference CPU backend, {nth} threads")
cpu_pp = pp512(f"reference CPU backend, {nth} threads", 2)
if plugin:
    R.load_backend(os.path.join(ROOT, "llama.cpp_test_amd", "libggml-mi355.so"))
    R.set_gpu_layers(99)
    gpu = tg128("reference runtime + libggml-mi355.so, all layers on the MI355X")
    err = float(((gpu - cpu) ** 2).sum() / (cpu ** 2).sum())
    print(f"last-token logits, plugin vs CPU: NMSE {err:.2e}")
    gpu_pp = pp512("reference runtime + libggml-mi355.so, all layers on the MI355X", 5)
    print(f"last-token logits of the prompt, plugin vs CPU: NMSE {float(((gpu_pp - cpu_pp) ** 2).sum() / (cpu_pp ** 2).sum()):.2e}")
os.remove(path)
