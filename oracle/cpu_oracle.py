"""ctypes front-end for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (llama.cpp_test_amd) never does.

* `port`  = oracle/libpkv_oracle.so  — our C restatement (oracle/pkv_oracle.c), builds anywhere gcc is.
* `ref`   = oracle/_ref/libref_driver.so — the reference's own compiled CPU ggml path (built in place from
            /root/reference by `make -C oracle ref`); present only where it was built (dev container, or
            shipped prebuilt to the GPU box).  `ref_available()` tells.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, IQ4_NL, BF16 = 0, 1, 2, 3, 6, 7, 8, 20, 30
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", IQ4_NL: "iq4_nl", BF16: "bf16"}
KQ_MASK_PAD = 64  # GGML_KQ_MASK_PAD, ggml/include/ggml.h:1792

_port = None
_ref = None


def build_port(force: bool = False) -> str:
    so = os.path.join(HERE, "libpkv_oracle.so")
    src = [os.path.join(HERE, f) for f in ("pkv_oracle.c", "pkv_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", HERE, "port"], stdout=subprocess.DEVNULL)
    return so


def build_ref() -> str | None:
    """Compile the reference in place (only possible where /root/reference exists)."""
    so = os.path.join(HERE, "_ref", "libref_driver.so")
    if os.path.isdir("/root/reference/ggml/src"):
        subprocess.check_call(["make", "-C", HERE, "-j8", "ref"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


def port():
    global _port
    if _port is None:
        _port = C.CDLL(build_port())
        _port.pkv_o_row_size.restype = C.c_size_t
        _port.pkv_o_row_size.argtypes = [C.c_int, C.c_int64]
        _port.pkv_o_tile_bytes.restype = C.c_size_t
        _port.pkv_o_tile_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    return _port


def ref_available() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_driver.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libref_driver.so"))
        _ref.ref_row_size.restype = C.c_size_t
        _ref.ref_row_size.argtypes = [C.c_int, C.c_int64]
    return _ref


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(x):
    return C.c_float(float(x))


def row_size(t: int, ne: int) -> int:
    return port().pkv_o_row_size(t, ne)


def tile_bytes(t: int, block_size: int, D: int) -> int:
    return port().pkv_o_tile_bytes(t, block_size, D)


# ---- codecs ------------------------------------------------------------------------------------------
def from_float(t: int, x: np.ndarray, variant: str = "cpu") -> np.ndarray:
    """f32 [..., ne0] -> reference-format bytes [..., row_size].  variant: 'cpu' (what ggml_cpy stores) | 'ref'."""
    x = np.ascontiguousarray(x, np.float32)
    n = x.size
    out = np.empty(x.shape[:-1] + (row_size(t, x.shape[-1]),), np.uint8)
    if t == Q8_0 and variant == "ref":
        port().pkv_o_quantize_q8_0_ref(_p(x), _p(out), C.c_int64(n))
    else:
        port().pkv_o_from_float(t, _p(x), _p(out), C.c_int64(n))
    return out


def to_float(t: int, b: np.ndarray, ne0: int) -> np.ndarray:
    b = np.ascontiguousarray(b, np.uint8)
    rows = b.size // row_size(t, ne0)
    out = np.empty(b.shape[:-1] + (ne0,), np.float32)
    port().pkv_o_to_float(t, _p(b), _p(out), C.c_int64(rows * ne0))
    return out


def rope(x, pos, ff=None, *, n_dims, mode=0, n_ctx_orig=8192, freq_base=500000.0, freq_scale=1.0,
         ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0, inverse=False) -> np.ndarray:
    """x f32 [n_tok][n_head][ne0]"""
    x = np.ascontiguousarray(x, np.float32)
    pos = np.ascontiguousarray(pos, np.int32)
    ff = None if ff is None else np.ascontiguousarray(ff, np.float32)
    n_tok, n_head, ne0 = x.shape
    out = np.empty_like(x)
    port().pkv_o_rope(_p(x), _p(pos), _p(ff), _p(out), ne0, n_head, n_tok, n_dims, mode, n_ctx_orig,
                      _f(freq_base), _f(freq_scale), _f(ext_factor), _f(attn_factor), _f(beta_fast), _f(beta_slow),
                      int(inverse))
    return out


def flash_attn_ext(q, k, v, mask, *, type_k, type_v, scale, max_bias=0.0, softcap=0.0, n_threads=1,
                   want_logits=False):
    """q f32 [n_head][n_q][D]; k/v uint8 [n_head_kv][n_kv][row]; mask f16 [>=n_q][n_kv] or None -> out [n_q][n_head][D]"""
    q = np.ascontiguousarray(q, np.float32)
    n_head, n_q, D = q.shape
    n_head_kv, n_kv = k.shape[0], k.shape[1]
    k = np.ascontiguousarray(k, np.uint8)
    v = np.ascontiguousarray(v, np.uint8)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.float16)
        assert mask.shape[1] == n_kv and mask.shape[0] >= n_q
    out = np.empty((n_q, n_head, D), np.float32)
    lg = np.empty((n_head, n_q, n_kv), np.float32) if want_logits else None
    rc = port().pkv_o_flash_attn_ext(_p(q), _p(k), _p(v), _p(mask), _p(out), _p(lg), D, n_q, n_head, n_kv, n_head_kv,
                                     n_kv, type_k, type_v, _f(scale), _f(max_bias), _f(softcap), n_threads)
    assert rc == 0, rc
    return (out, lg) if want_logits else out


def flash_attn_f64(q, k, v, mask, *, type_k, type_v, scale, softcap=0.0):
    q = np.ascontiguousarray(q, np.float32)
    n_head, n_q, D = q.shape
    n_head_kv, n_kv = k.shape[0], k.shape[1]
    k = np.ascontiguousarray(k, np.uint8)
    v = np.ascontiguousarray(v, np.uint8)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.float16)
    out = np.empty((n_q, n_head, D), np.float64)
    rc = port().pkv_o_flash_attn_f64(_p(q), _p(k), _p(v), _p(mask), _p(out), D, n_q, n_head, n_kv, n_head_kv, n_kv,
                                     type_k, type_v, _f(scale), _f(softcap))
    assert rc == 0, rc
    return out


# ---- paged pool ----------------------------------------------------------------------------------------
def pool_alloc(t: int, n_blocks: int, n_head_kv: int, block_size: int, D: int) -> np.ndarray:
    return np.zeros((n_blocks, n_head_kv, tile_bytes(t, block_size, D)), np.uint8)


def pool_store(pool, t, is_v, rows, slot_block, slot_off, *, D, block_size):
    """rows uint8 [n_tok][n_head_kv][row_size] scattered to (slot_block[t], slot_off[t])"""
    rows = np.ascontiguousarray(rows, np.uint8)
    sb = np.ascontiguousarray(slot_block, np.int32)
    so = np.ascontiguousarray(slot_off, np.int32)
    assert pool.flags.c_contiguous
    port().pkv_o_pool_store(_p(pool), t, int(is_v), _p(rows), _p(sb), _p(so), rows.shape[0], rows.shape[1], D, block_size)


def pool_gather(pool, t, is_v, block_table, n_kv, *, D, block_size) -> np.ndarray:
    bt = np.ascontiguousarray(block_table, np.int32)
    n_head_kv = pool.shape[1]
    out = np.empty((n_head_kv, n_kv, row_size(t, D)), np.uint8)
    port().pkv_o_pool_gather(_p(pool), t, int(is_v), _p(bt), n_kv, n_head_kv, D, block_size, _p(out))
    return out


def attn_paged(q, k_pool, v_pool, block_table, seq_lens, *, n_q, type_k, type_v, block_size, scale, softcap=0.0,
               n_threads=1) -> np.ndarray:
    """q f32 [n_seq][n_head][n_q][D] -> out [n_seq][n_q][n_head][D]"""
    q = np.ascontiguousarray(q, np.float32)
    n_seq, n_head, nq, D = q.shape
    assert nq == n_q
    bt = np.ascontiguousarray(block_table, np.int32)
    sl = np.ascontiguousarray(seq_lens, np.int32)
    out = np.empty((n_seq, n_q, n_head, D), np.float32)
    rc = port().pkv_o_attn_paged(_p(q), _p(k_pool), _p(v_pool), _p(bt), bt.shape[1], _p(sl), _p(out), n_seq, n_q, D,
                                 n_head, k_pool.shape[1], block_size, type_k, type_v, _f(scale), _f(softcap), n_threads)
    assert rc == 0, rc
    return out


# ---- the compiled reference (oracle/_ref) ----------------------------------------------------------------
def ref_flash_attn_ext(q, k, v, mask, *, type_k, type_v, scale, max_bias=0.0, softcap=0.0, n_threads=1):
    q = np.ascontiguousarray(q, np.float32)
    n_head, n_q, D = q.shape
    n_head_kv, n_kv = k.shape[0], k.shape[1]
    n_q_pad = -(-n_q // KQ_MASK_PAD) * KQ_MASK_PAD
    mp = None
    if mask is not None:
        mp = np.full((n_q_pad, n_kv), -np.inf, np.float16)
        mp[:n_q] = mask[:n_q]
    out = np.empty((n_q, n_head, D), np.float32)
    rc = ref().ref_flash_attn_ext(_p(q), _p(np.ascontiguousarray(k)), _p(np.ascontiguousarray(v)), _p(mp), _p(out),
                                  D, n_q, n_head, n_kv, n_head_kv, n_q_pad, type_k, type_v, _f(scale), _f(max_bias),
                                  _f(softcap), n_threads)
    assert rc == 0, rc
    return out


def ref_rope(x, pos, ff=None, *, n_dims, mode=0, n_ctx_orig=8192, freq_base=500000.0, freq_scale=1.0,
             ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0, n_threads=1):
    x = np.ascontiguousarray(x, np.float32)
    pos = np.ascontiguousarray(pos, np.int32)
    ff = None if ff is None else np.ascontiguousarray(ff, np.float32)
    n_tok, n_head, ne0 = x.shape
    out = np.empty_like(x)
    rc = ref().ref_rope_ext(_p(x), _p(pos), _p(ff), _p(out), ne0, n_head, n_tok, n_dims, mode, n_ctx_orig,
                            _f(freq_base), _f(freq_scale), _f(ext_factor), _f(attn_factor), _f(beta_fast),
                            _f(beta_slow), n_threads)
    assert rc == 0, rc
    return out


def ref_cpy(t: int, x: np.ndarray, n_threads=1) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    ne0 = x.shape[-1]
    rows = x.size // ne0
    out = np.empty(x.shape[:-1] + (ref().ref_row_size(t, ne0),), np.uint8)
    rc = ref().ref_cpy_f32(_p(x), _p(out), ne0, rows, t, n_threads)
    assert rc == 0, rc
    return out


def ref_from_float_ref(t: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape[:-1] + (ref().ref_row_size(t, x.shape[-1]),), np.uint8)
    rc = ref().ref_from_float_ref(_p(x), _p(out), C.c_int64(x.size), t)
    assert rc == 0, rc
    return out


def ref_to_float(t: int, b: np.ndarray, ne0: int) -> np.ndarray:
    b = np.ascontiguousarray(b, np.uint8)
    rows = b.size // ref().ref_row_size(t, ne0)
    out = np.empty(b.shape[:-1] + (ne0,), np.float32)
    rc = ref().ref_to_float(_p(b), _p(out), C.c_int64(rows * ne0), t)
    assert rc == 0, rc
    return out
