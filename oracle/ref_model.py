"""ctypes front end of the model/GGUF half of oracle/_ref/libref_kv.so (driver oracle/ref_model_driver.cpp): the REFERENCE's
GGUF reader/writer and its llama runtime on the CPU backend.  TEST INFRASTRUCTURE ONLY; exists only where the reference was
compiled (this container) — elsewhere tests use the fixtures tests/golden/make_model_golden.py produced with it."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import ref_kv


def available() -> bool:
    return ref_kv.available()


def _lib():
    L = ref_kv.lib()
    L.refgguf_describe.restype = C.c_int64
    L.refmodel_load.restype = C.c_void_p
    L.refmodel_free.restype = None
    L.refmodel_capture_data.restype = None
    L.refmodel_seq_rm.restype = None
    L.refmodel_seq_cp.restype = None
    L.refmodel_seq_add.restype = None
    return L


def gguf_describe(path: str) -> str:
    n = _lib().refgguf_describe(path.encode(), None, C.c_size_t(0))
    if n < 0:
        raise RuntimeError(f"the reference's gguf_init_from_file rejected {path}")
    buf = C.create_string_buffer(int(n) + 1)
    _lib().refgguf_describe(path.encode(), buf, C.c_size_t(int(n) + 1))
    return buf.value.decode()


def gguf_write_demo(path: str) -> None:
    if _lib().refgguf_write_demo(path.encode()) != 0:
        raise RuntimeError("gguf_write_to_file failed")


_loaded_backends = set()


def load_backend(so_path: str) -> None:
    """register a ggml-backend plugin with the reference's registry (ggml_backend_load = what GGML_BACKEND_PATH triggers).  Once per process and path: every
    ggml_backend_load registers the plugin's device AGAIN, and llama_context hands all registered devices to ggml_backend_sched_new, which asserts at
    GGML_SCHED_MAX_BACKENDS = 16 (/root/reference/ggml/src/ggml-backend.cpp) — the sixteenth test of a file that loads the plugin per test aborted there."""
    key = os.path.realpath(so_path)
    if key in _loaded_backends:
        return
    if _lib().refmodel_load_backend(so_path.encode()) != 0:
        raise RuntimeError(f"ggml_backend_load({so_path}) failed")
    _loaded_backends.add(key)


def set_defrag_thold(t: float) -> None:
    """llama_context_params.defrag_thold of the contexts created from here on (-1 = never, the library default; 0.1 = what llama-cli / llama-server run with)"""
    _lib().refmodel_set_defrag_thold(C.c_float(t))


def set_gpu_layers(n: int) -> None:
    """llama_model_params.n_gpu_layers of the models loaded next (0 = everything on the reference's CPU backend)"""
    _lib().refmodel_set_gpu_layers(n)


class RefModel:
    """llama_model + llama_context of the reference on its CPU backend; decode() returns the captured attention tensors"""

    def __init__(self, path: str, *, n_ctx: int, n_batch: int = 512, n_seq_max: int = 1, flash_attn: bool = True, type_k: int = 1,
                 type_v: int = 1, n_threads: int = 8):
        self.h = C.c_void_p(_lib().refmodel_load(path.encode(), n_ctx, n_batch, n_seq_max, int(flash_attn), type_k, type_v, n_threads))
        if not self.h:
            raise RuntimeError(f"the reference could not load {path}")
        self.op_rope = _lib().refmodel_op_rope()

    def __del__(self):
        if getattr(self, "h", None):
            _lib().refmodel_free(self.h)
            self.h = None

    def logits(self, n_vocab: int) -> np.ndarray:
        out = np.empty(n_vocab, np.float32)
        if _lib().refmodel_logits(self.h, out.ctypes.data_as(C.c_void_p), n_vocab) != n_vocab:
            raise RuntimeError("no logits")
        return out

    def state_seq_get(self, seq_id: int) -> bytes:
        L = _lib()
        L.refmodel_state_seq_get.restype = C.c_size_t
        n = L.refmodel_state_seq_get(self.h, None, C.c_size_t(0), seq_id)
        buf = C.create_string_buffer(n)
        got = L.refmodel_state_seq_get(self.h, buf, C.c_size_t(n), seq_id)
        return buf.raw[:got]

    def state_seq_set(self, blob: bytes, seq_id: int) -> int:
        L = _lib()
        L.refmodel_state_seq_set.restype = C.c_size_t
        return int(L.refmodel_state_seq_set(self.h, blob, C.c_size_t(len(blob)), seq_id))

    def weight_get(self, name: str, offset: int = 0, size: int = -1):
        """(bytes of model tensor `name` through ggml_backend_tensor_get, whether its buffer is a device buffer)"""
        L = _lib()
        L.refmodel_weight_get.restype = C.c_size_t
        dev = C.c_int(0)
        n = L.refmodel_weight_get(self.h, name.encode(), None, C.c_size_t(0), C.c_size_t(0), C.byref(dev))
        if n == 0:
            raise KeyError(name)
        size = n - offset if size < 0 else size
        buf = (C.c_uint8 * size)()
        got = L.refmodel_weight_get(self.h, name.encode(), buf, C.c_size_t(size), C.c_size_t(offset), C.byref(dev))
        return bytes(buf[:got]), bool(dev.value)

    def weight_set(self, name: str, data: bytes, offset: int = 0) -> int:
        L = _lib()
        L.refmodel_weight_set.restype = C.c_size_t
        return int(L.refmodel_weight_set(self.h, name.encode(), data, C.c_size_t(len(data)), C.c_size_t(offset)))

    def seq_rm(self, s, p0, p1): _lib().refmodel_seq_rm(self.h, s, p0, p1)
    def seq_cp(self, a, b, p0, p1): _lib().refmodel_seq_cp(self.h, a, b, p0, p1)
    def seq_add(self, s, p0, p1, d): _lib().refmodel_seq_add(self.h, s, p0, p1, d)

    def decode(self, tokens, pos, seq_id):
        """one llama_decode; returns {layer: dict(q, k, v = un-rotated projections [n_tok][heads][D]; q_rope, k_rope; out = kqv_out
        [n_tok][n_head][D])}"""
        t, p, s = (np.ascontiguousarray(x, np.int32) for x in (tokens, pos, seq_id))
        rc = _lib().refmodel_decode(self.h, t.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), len(t))
        if rc != 0:
            raise RuntimeError(f"llama_decode returned {rc}")
        layers = {}
        name = C.create_string_buffer(64)
        for i in range(_lib().refmodel_n_captured(self.h)):
            op, ne = C.c_int(), (C.c_int64 * 4)()
            n = _lib().refmodel_capture_info(self.h, i, name, 64, C.byref(op), ne)
            data = np.empty(n, np.float32)
            _lib().refmodel_capture_data(self.h, i, data.ctypes.data_as(C.c_void_p))
            kind, il = name.value.decode().rsplit("-", 1)
            arr = data.reshape([int(x) for x in reversed(list(ne)) if True])       # ggml ne order reversed = C order
            key = {"Qcur": "q", "Kcur": "k", "Vcur": "v", "kqv_out": "out"}[kind] + ("_rope" if op.value == self.op_rope else "")
            layers.setdefault(int(il), {})[key] = arr
        return layers
