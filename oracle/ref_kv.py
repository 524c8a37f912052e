"""ctypes front end of oracle/_ref/libref_kv.so: the REFERENCE's llama_kv_cache_unified (compiled in place from
/root/reference/src by `make -C oracle ref`) behind our own driver oracle/ref_kv_driver.cpp.  TEST INFRASTRUCTURE ONLY.

Exists only where the reference was compiled (this container); tests that need it skip elsewhere and work from the
fixtures under tests/golden/ that tests/golden/make_state_golden.py produced with it."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_kv.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        # ggml first and globally: a ggml-backend plugin dlopen()ed later resolves ggml_backend_buffer_init & co against it
        C.CDLL(os.path.join(_HERE, "_ref", "libggml_ref.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(_PATH, mode=C.RTLD_GLOBAL)
        L.refkv_create.restype = C.c_void_p
        L.refkv_state_write.restype = C.c_int64
        for f in ("refkv_free", "refkv_advance_head", "refkv_clear", "refkv_seq_cp", "refkv_seq_keep", "refkv_seq_add",
                  "refkv_seq_div", "refkv_cells", "refkv_tensor_set", "refkv_tensor_get", "refkv_view"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefKvCache:
    """llama_kv_cache_unified for a llama-arch model: n_layer layers, n_head_kv * head_dim elements per cell row"""

    def __init__(self, *, n_layer: int, n_head_kv: int, head_dim: int, type_k: int, type_v: int, kv_size: int, flash_attn: bool = True):
        self.h = C.c_void_p(lib().refkv_create(n_layer, n_head_kv, head_dim, type_k, type_v, kv_size, int(flash_attn)))
        if not self.h:
            raise RuntimeError("llama_kv_cache_unified::init failed")
        self.size, self.n_layer = kv_size, n_layer

    def __del__(self):
        if getattr(self, "h", None):
            lib().refkv_free(self.h)
            self.h = None

    def find_slot(self, seq_ids, poss) -> bool:
        s, p = np.ascontiguousarray(seq_ids, np.int32), np.ascontiguousarray(poss, np.int32)
        return lib().refkv_find_slot(self.h, len(s), _p(s), _p(p)) >= 0

    def advance_head(self, n): lib().refkv_advance_head(self.h, n)
    def clear(self): lib().refkv_clear(self.h)
    def seq_rm(self, s, p0, p1) -> bool: return bool(lib().refkv_seq_rm(self.h, s, p0, p1))
    def seq_cp(self, a, b, p0, p1): lib().refkv_seq_cp(self.h, a, b, p0, p1)
    def seq_keep(self, s): lib().refkv_seq_keep(self.h, s)
    def seq_add(self, s, p0, p1, d): lib().refkv_seq_add(self.h, s, p0, p1, d)
    def seq_div(self, s, p0, p1, d): lib().refkv_seq_div(self.h, s, p0, p1, d)
    def seq_pos_max(self, s) -> int: return lib().refkv_seq_pos_max(self.h, s)
    def get_n_tokens(self) -> int: return lib().refkv_n_tokens(self.h)
    def get_used_cells(self) -> int: return lib().refkv_used_cells(self.h)

    @property
    def head(self) -> int: return lib().refkv_head(self.h)
    @property
    def has_shift(self) -> bool: return bool(lib().refkv_has_shift(self.h))
    @property
    def v_trans(self) -> bool: return bool(lib().refkv_v_trans(self.h))

    def cells(self, max_seq: int = 16):
        """[(pos, delta, sorted seq ids)] for every cell of the ring"""
        pos, dl, ns = (np.empty(self.size, np.int32) for _ in range(3))
        seqs = np.zeros((self.size, max_seq), np.int32)
        lib().refkv_cells(self.h, _p(pos), _p(dl), _p(ns), _p(seqs), max_seq)
        return [(int(pos[i]), int(dl[i]), [int(x) for x in seqs[i, :ns[i]]]) for i in range(self.size)]

    def view(self, n_seq_max: int = 4):
        st, pos, seqs = np.empty(5, np.int32), np.empty(self.size, np.int32), np.empty((self.size, n_seq_max), np.int32)
        lib().refkv_view(self.h, n_seq_max, _p(st), _p(pos), _p(seqs))
        return dict(n_cells=int(st[0]), token_count=int(st[1]), used_cells=int(st[2]), max_contiguous=int(st[3]),
                    max_contiguous_idx=int(st[4]), cells=pos, cells_sequences=seqs)

    def tensor_set(self, il: int, is_v: bool, data: np.ndarray, offset: int = 0):
        d = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        lib().refkv_tensor_set(self.h, il, int(is_v), _p(d), C.c_size_t(offset), C.c_size_t(d.size))

    def tensor_get(self, il: int, is_v: bool, nbytes: int, offset: int = 0) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        lib().refkv_tensor_get(self.h, il, int(is_v), _p(out), C.c_size_t(offset), C.c_size_t(nbytes))
        return out

    def state_write(self, seq_id: int = -1) -> bytes:
        n = lib().refkv_state_write(self.h, seq_id, None, C.c_size_t(0))
        buf = C.create_string_buffer(max(int(n), 1))
        lib().refkv_state_write(self.h, seq_id, buf, C.c_size_t(int(n)))
        return buf.raw[:int(n)]

    def state_read(self, data: bytes, seq_id: int = -1) -> bool:
        return lib().refkv_state_read(self.h, seq_id, data, C.c_size_t(len(data))) == 0
