"""The headline attention shape inside the UNMODIFIED reference runtime (GPU box): a 2-layer synthetic model with Llama-3-8B's
attention geometry (32 heads, 8 kv heads, head_dim 128), ctx 4096, loaded by the reference's loader with libggml-mi355.so
registered and n_gpu_layers = 99.  4064 prompt tokens go in as eight 508-token ubatches (masked MFMA prefill kernel, the paged
mirrors of the cache tensors grow ubatch by ubatch), then 32 single-token llama_decode calls are timed: each runs the split-KV
decode kernel on ~4096 cells with the KQ_mask row the reference built.  Run under rocprofv3 --kernel-trace --stats to see the
kernels.  usage: python tests/evidence/ref_runtime_ctx4096.py"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _load_pkg
_load_pkg.load()
from llama_cpp_test_amd import gguf_synth as G
from oracle import ref_model as R
import ctypes as C

path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "attn8b-2l.gguf")
G.synth_llama(path, n_layer=2, n_embd=4096, n_head=32, n_head_kv=8, n_ff=1024, n_vocab=128, n_ctx=8192, rope_freq_base=500000.0, seed=3)
L = R._lib()
L.refmodel_detach_capture.restype = None
R.load_backend(os.path.join(ROOT, "llama.cpp_test_amd", "libggml-mi355.so"))
R.set_gpu_layers(99)
h = C.c_void_p(L.refmodel_load(path.encode(), 4352, 512, 1, 1, 1, 1, 16))
assert h
L.refmodel_detach_capture(h)
rng = np.random.default_rng(0)


def decode(tok, pos):
    t, p, s = np.ascontiguousarray(tok, np.int32), np.ascontiguousarray(pos, np.int32), np.zeros(len(tok), np.int32)
    assert L.refmodel_decode(h, t.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), len(t)) == 0


n = 0
t0 = time.time()
for _ in range(8):
    decode(rng.integers(0, 128, 508), np.arange(n, n + 508))
    n += 508
print(f"prompt: {n} tokens in {time.time() - t0:.2f} s", flush=True)
decode([1], [n]); n += 1
t0 = time.time()
for i in range(32):
    decode([int(i % 128)], [n]); n += 1
dt = time.time() - t0
print(f"tg32 at ctx {n}: {dt / 32 * 1e3:.3f} ms/token for 2 layers (reference runtime + plugin, everything on the MI355X)")
os.remove(path)
