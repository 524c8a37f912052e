"""llama.cpp_test_amd — MI355X-native paged-KV-cache attention path for llama.cpp (host-side Python mirror).

This package is a THIN ctypes binding over the C ABI in include/mi355_paged_kv.h
(libmi355_paged_kv.so = hand-written HIP kernels for gfx950 + the C++ host block manager).  It mirrors the
reference's interface for this path — the `llama_kv_cache_*` method set (/root/reference/src/llama-kv-cache.h:17-26,
/root/reference/src/llama-memory.h:7-21) and the three graph ops of `build_attn` (KV store, RoPE, flash-attn;
/root/reference/src/llama-graph.cpp:1340-1432) — so that the parity tests read like the reference's own tests.

There is NO fallback: if the shared library is missing the import of any symbol raises, and every compute entry
point raises `PkvError` when no gfx950 device is present.  PyTorch is only plumbing here (device buffers, streams):
every tensor argument is reduced to a raw device pointer before crossing the C ABI.

The directory name contains a dot, so the package is loaded by path (see `_load_pkg.py` at the repo root) under the
module name `llama_cpp_test_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PKV_LIB_PATH") or os.path.join(_HERE, "libmi355_paged_kv.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mi355_paged_kv.h")

# enum pkv_type (= enum ggml_type values)
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, BF16 = 0, 1, 2, 8, 12, 13, 14, 30
Q4_1, Q5_0, Q5_1, IQ4_NL = 3, 6, 7, 20   # cache types that exist as flat rows only (convert_f32, flash_attn_ext_flat)
TYPE_BY_NAME = {"f32": F32, "f16": F16, "q4_0": Q4_0, "q8_0": Q8_0, "q4_K": Q4_K, "q5_K": Q5_K, "q6_K": Q6_K, "bf16": BF16, "q4_1": Q4_1, "q5_0": Q5_0, "q5_1": Q5_1,
                "iq4_nl": IQ4_NL}
# enum pkv_status
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NO_SPACE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
PACKED = 0x100          # PKV_TYPE_PACKED: K-quant rows in the plane layout (csrc/pkv_wlayout.h), produced by weights_pack()
ATTN_NO_COMBINE = 1
ATTN_FUSED_COMBINE = 2
ATTN_NONCAUSAL = 4
ABI_VERSION = 3      # == PKV_ABI_VERSION of include/mi355_paged_kv.h
ATTN_Q_F16 = 8      # attn_prefill / attn_prefill_step: q is f16 (set from q.dtype)
ATTN_OUT_F16 = 16   # ... out is f16 (out_dtype=torch.float16 or an f16 `out`)


class PkvError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"pkv status {status}: {msg}")
        self.status = status


class RopeParams(C.Structure):
    """pkv_rope_params = op_params of GGML_OP_ROPE (/root/reference/ggml/src/ggml-cpu/ggml-cpu.c:10681-10693)"""
    _fields_ = [("n_dims", C.c_int32), ("mode", C.c_int32), ("n_ctx_orig", C.c_int32),
                ("freq_base", C.c_float), ("freq_scale", C.c_float), ("ext_factor", C.c_float),
                ("attn_factor", C.c_float), ("beta_fast", C.c_float), ("beta_slow", C.c_float),
                ("freq_factors", C.c_void_p)]


class StepArgs(C.Structure):
    _fields_ = [("k_cur", C.c_void_p), ("v_cur", C.c_void_p), ("pos", C.c_void_p), ("rope", C.POINTER(RopeParams)),
                ("rope_cache", C.c_void_p)]


class QkvStoreArgs(C.Structure):
    """pkv_qkv_store_args"""
    _fields_ = [("q", C.c_void_p), ("q_stride_tok", C.c_int64), ("q_stride_head", C.c_int64), ("q_out", C.c_void_p),
                ("k", C.c_void_p), ("k_stride_tok", C.c_int64), ("k_stride_head", C.c_int64), ("k_dst", C.c_void_p), ("k_pool", C.c_void_p), ("k_cell0", C.c_int64),
                ("v", C.c_void_p), ("v_dst", C.c_void_p), ("v_pool", C.c_void_p), ("v_cell0", C.c_int64),
                ("pos", C.c_void_p), ("rope", C.POINTER(RopeParams)),
                ("type_k", C.c_int32), ("type_v", C.c_int32), ("n_tok", C.c_int32), ("n_head", C.c_int32), ("n_head_kv", C.c_int32), ("head_dim", C.c_int32),
                ("block_size", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("q_stride_tok", C.c_int64), ("q_stride_head", C.c_int64), ("out", C.c_void_p),
                ("k_pool", C.c_void_p), ("v_pool", C.c_void_p), ("block_table", C.c_void_p), ("seq_lens", C.c_void_p),
                ("q_start", C.c_void_p), ("slot_pos", C.c_void_p), ("q_pos", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t),
                ("n_seq", C.c_int32), ("n_tok", C.c_int32), ("max_q_per_seq", C.c_int32), ("n_head", C.c_int32),
                ("n_head_kv", C.c_int32), ("head_dim", C.c_int32), ("block_size", C.c_int32), ("max_blocks", C.c_int32),
                ("type_k", C.c_int32), ("type_v", C.c_int32), ("scale", C.c_float), ("logit_softcap", C.c_float),
                ("max_seq_len", C.c_int32), ("flags", C.c_int32),
                ("mask", C.c_void_p), ("mask_stride", C.c_int64), ("max_bias", C.c_float),
                ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p), ("out_image", C.c_void_p)]


class FlatAttnArgs(C.Structure):   # pkv_flat_attn_args
    _fields_ = [("q", C.c_void_p), ("q_stride_tok", C.c_int64), ("q_stride_head", C.c_int64),
                ("k", C.c_void_p), ("k_stride_tok", C.c_int64), ("k_stride_head", C.c_int64),
                ("v", C.c_void_p), ("v_stride_tok", C.c_int64), ("v_stride_head", C.c_int64),
                ("mask", C.c_void_p), ("mask_stride", C.c_int64), ("out", C.c_void_p),
                ("n_q", C.c_int32), ("n_kv", C.c_int32), ("n_head", C.c_int32), ("n_head_kv", C.c_int32), ("head_dim", C.c_int32),
                ("type_k", C.c_int32), ("type_v", C.c_int32), ("scale", C.c_float), ("max_bias", C.c_float), ("logit_softcap", C.c_float),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class Gemv(C.Structure):
    """pkv_gemv"""
    _fields_ = [("n", C.c_int32), ("silu_mul", C.c_int32), ("type_w", C.c_int32 * 3), ("w", C.c_void_p * 3), ("y", C.c_void_p * 3),
                ("M", C.c_int64 * 3), ("K", C.c_int64), ("xq", C.c_void_p), ("residual", C.c_void_p), ("x", C.c_void_p), ("norm_w", C.c_void_p),
                ("eps", C.c_float), ("norm_out", C.c_void_p), ("col_out", C.c_void_p)]


class CacheParams(C.Structure):
    _fields_ = [("n_layer", C.c_int32), ("n_head_kv", C.c_int32), ("head_dim", C.c_int32), ("type_k", C.c_int32),
                ("type_v", C.c_int32), ("block_size", C.c_int32), ("n_blocks", C.c_int32), ("n_seq_max", C.c_int32),
                ("device", C.c_int32), ("evict_when_full", C.c_int32)]


_lib = None


def lib() -> C.CDLL:
    """The C-ABI library.  Fails loudly when it has not been built (no Python/CPU substitute exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); this path has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.pkv_last_error.restype = C.c_char_p
        L.pkv_device_arch.restype = C.c_char_p
        for f in ("pkv_row_size", "pkv_tile_bytes", "pkv_pool_bytes", "pkv_attn_workspace_bytes"):
            getattr(L, f).restype = C.c_size_t
        L.pkv_row_size.argtypes = [C.c_int, C.c_int64]
        L.pkv_cache_create.restype = C.c_void_p
        L.pkv_attn_prefill_workspace_bytes.restype = C.c_size_t
        L.pkv_flash_attn_ext_flat_workspace_bytes.restype = C.c_size_t
        if hasattr(L, "pkv_act_image_bytes"):   # (absent only from the round-1 build that probes/ab_decode.py loads for A/B runs)
            L.pkv_act_image_bytes.restype = C.c_size_t
            L.pkv_act_image_bytes.argtypes = [C.c_int64]
        L.pkv_cache_state_size.restype = C.c_int64
        L.pkv_cache_state_write.restype = C.c_int64
        L.pkv_cache_create.argtypes = [C.POINTER(CacheParams)]
        for f in ("pkv_cache_k_pool", "pkv_cache_v_pool", "pkv_toy_bm_create", "pkv_toy_kvm_create"):
            getattr(L, f).restype = C.c_void_p
        for f in ("pkv_cache_free", "pkv_cache_clear", "pkv_cache_seq_cp", "pkv_cache_seq_keep", "pkv_cache_seq_add",
                  "pkv_cache_seq_div", "pkv_cache_defrag", "pkv_toy_bm_free", "pkv_toy_bm_release", "pkv_toy_kvm_free"):
            getattr(L, f).restype = None
        if L.pkv_abi_version() != ABI_VERSION:   # (a stale build would read an f16 q as f32: refuse it here, not in a kernel)
            raise RuntimeError(f"libmi355_paged_kv.so is ABI version {L.pkv_abi_version()}, this binding needs {ABI_VERSION}: rebuild (make -C llama.cpp_test_amd/csrc)")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise PkvError(rc, lib().pkv_last_error().decode())


def device_count() -> int:
    return lib().pkv_device_count()


def _ptr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(f"expected a device tensor or an address, got {type(x)}")


def _stream(stream=None) -> int:
    if stream is not None:
        return stream if isinstance(stream, int) else stream.cuda_stream
    import torch
    return torch.cuda.current_stream().cuda_stream


def row_size(t: int, ne: int) -> int:
    return lib().pkv_row_size(t, ne)


def tile_bytes(t: int, block_size: int, head_dim: int) -> int:
    return lib().pkv_tile_bytes(t, block_size, head_dim)


def pool_bytes(t: int, n_blocks: int, n_head_kv: int, block_size: int, head_dim: int) -> int:
    return lib().pkv_pool_bytes(t, n_blocks, n_head_kv, block_size, head_dim)


def rope_params(*, n_dims: int, mode: int = 0, n_ctx_orig: int = 8192, freq_base: float = 500000.0,
                freq_scale: float = 1.0, ext_factor: float = 0.0, attn_factor: float = 1.0, beta_fast: float = 32.0,
                beta_slow: float = 1.0, freq_factors=None) -> RopeParams:
    """Defaults are Llama-3 (rope.freq_base 5e5, NORM pairs; /root/reference/src/llama-model.cpp:11798)."""
    rp = RopeParams(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow,
                    _ptr(freq_factors))
    rp._keepalive = freq_factors   # the struct only carries the device address
    return rp


# ---- ops (device tensors in, device tensors out) -----------------------------------------------------------------
def rope_f32(x, pos, rope: RopeParams, *, inverse: bool = False, out=None, stream=None):
    """ggml_rope_ext on x f32 [n_tok][n_head][head_dim] (/root/reference/ggml/include/ggml.h:1438)."""
    n_tok, n_head, D = x.shape
    out = x.new_empty(x.shape) if out is None else out
    check(lib().pkv_rope_f32(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(pos)), C.c_void_p(_ptr(out)), n_tok, n_head, D,
                             C.byref(rope), int(inverse), C.c_void_p(_stream(stream))))
    return out


def kv_store(k_cur, v_cur, k_pool, v_pool, *, type_k: int, type_v: int, slot_block, slot_off, block_size: int,
             pos=None, rope: Optional[RopeParams] = None, stream=None) -> None:
    """ggml_cpy(k_cur/v_cur -> cache) of build_attn (/root/reference/src/llama-graph.cpp:1367-1398), optionally fused
    with RoPE(K).  k_cur/v_cur f32 [n_tok][n_head_kv][head_dim]."""
    ref = k_cur if k_cur is not None else v_cur
    n_tok, n_head_kv, D = ref.shape
    check(lib().pkv_kv_store(C.c_void_p(_ptr(k_cur)), C.c_void_p(_ptr(v_cur)), C.c_void_p(_ptr(k_pool)),
                             C.c_void_p(_ptr(v_pool)), type_k, type_v, C.c_void_p(_ptr(slot_block)),
                             C.c_void_p(_ptr(slot_off)), C.c_void_p(_ptr(pos)), C.byref(rope) if rope is not None else None,
                             n_tok, n_head_kv, D, block_size, C.c_void_p(_stream(stream))))


def qkv_prepare(q, k_cur, v_cur, k_pool, v_pool, *, type_k: int, type_v: int, slot_block, slot_off, block_size: int,
                pos, rope: RopeParams, stream=None) -> None:
    """RoPE(Q) in place + RoPE(K) + KV store in one launch (llm_build_llama :4091-4102 + build_attn :1367-1398)."""
    n_tok, n_head, D = q.shape
    check(lib().pkv_qkv_prepare(C.c_void_p(_ptr(q)), n_head, C.c_void_p(_ptr(k_cur)), C.c_void_p(_ptr(v_cur)),
                                C.c_void_p(_ptr(k_pool)), C.c_void_p(_ptr(v_pool)), type_k, type_v,
                                C.c_void_p(_ptr(slot_block)), C.c_void_p(_ptr(slot_off)), C.c_void_p(_ptr(pos)),
                                C.byref(rope), n_tok, k_cur.shape[1], D, block_size, C.c_void_p(_stream(stream))))


def attn_workspace_bytes(n_tok: int, n_head: int, head_dim: int, max_seq_len: int, block_size: int) -> int:
    return lib().pkv_attn_workspace_bytes(n_tok, n_head, head_dim, max_seq_len, block_size)


def attn_decode_tiles(max_seq_len: int, n_head: int, n_head_kv: int, n_seq: int) -> int:
    """128-token tiles one workgroup of the plain decode launch walks for this launch shape (pkv_attn_decode_tiles)"""
    return int(lib().pkv_attn_decode_tiles(int(max_seq_len), int(n_head), int(n_head_kv), int(n_seq)))


def attn_prefill_workspace_bytes(n_tok: int, n_head: int, head_dim: int) -> int:
    """scratch attn_prefill can use for its split-KV mode (small launches over a long past); 0 above 2048 tokens"""
    return lib().pkv_attn_prefill_workspace_bytes(n_tok, n_head, head_dim)


def _attn(fn, q, k_pool, v_pool, block_table, seq_lens, *, n_head_kv, type_k, type_v, block_size, scale, max_seq_len,
          q_start=None, max_q_per_seq=1, slot_pos=None, q_pos=None, logit_softcap=0.0, out=None, workspace=None,
          stream=None, flags=0, events=None, mask=None, max_bias=0.0, step=None, stamps=None, slots=None, out_dtype=None, out_image=None):
    import torch
    n_tok, n_head, D = q.shape
    n_seq, max_blocks = block_table.shape
    out = torch.empty((n_tok, n_head, D), dtype=out_dtype or torch.float32, device=q.device) if out is None else out
    if q.dtype == torch.float16:
        flags |= ATTN_Q_F16
    if out.dtype == torch.float16:
        flags |= ATTN_OUT_F16
    # decode: per-block partials of the split-KV kernel; prefill: scratch of its split-KV mode (small launches only)
    need = (attn_workspace_bytes(n_tok, n_head, D, max_seq_len, block_size) if q_start is None
            else attn_prefill_workspace_bytes(n_tok, n_head, D))
    if workspace is None:
        workspace = torch.zeros(max(need, 16), dtype=torch.uint8, device=q.device)   # zeroed: valid for ATTN_FUSED_COMBINE
    a = AttnArgs(_ptr(q), q.stride(0), q.stride(1), _ptr(out), _ptr(k_pool), _ptr(v_pool), _ptr(block_table),
                 _ptr(seq_lens), _ptr(q_start), _ptr(slot_pos), _ptr(q_pos), _ptr(workspace), workspace.numel(),
                 n_seq, n_tok, max_q_per_seq, n_head, n_head_kv, D, block_size, max_blocks, type_k, type_v,
                 scale, logit_softcap, max_seq_len, flags, _ptr(mask), mask.stride(0) if mask is not None else 0, max_bias,
                 events[0] if events else None, events[1] if events else None, _ptr(out_image))
    if step is not None:
        k_cur, v_cur, pos, rope, rope_cache = step
        st = StepArgs(_ptr(k_cur), _ptr(v_cur), _ptr(pos), C.pointer(rope) if rope is not None else None, _ptr(rope_cache))
        if stamps is not None:
            check(fn(C.byref(a), C.byref(st), C.c_void_p(_ptr(stamps[0])), int(stamps[1]), C.c_void_p(_stream(stream))))
        elif slots is not None:
            check(fn(C.byref(a), C.byref(st), C.c_void_p(_ptr(slots[0])), C.c_void_p(_ptr(slots[1])), C.c_void_p(_stream(stream))))
        else:
            check(fn(C.byref(a), C.byref(st), C.c_void_p(_stream(stream))))
    else:
        check(fn(C.byref(a), C.c_void_p(_stream(stream))))
    return out


def attn_workspace(n_tok: int, n_head: int, head_dim: int, max_seq_len: int, block_size: int, device="cuda", min_bytes: int = 0):
    """a workspace that is valid for every flag combination (zeroed through pkv_attn_workspace_init)"""
    import torch
    ws = torch.empty(max(16, min_bytes, attn_workspace_bytes(n_tok, n_head, head_dim, max_seq_len, block_size)), dtype=torch.uint8,
                     device=device)
    check(lib().pkv_attn_workspace_init(C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), C.c_void_p(_stream())))
    return ws


def attn_decode(q, k_pool, v_pool, block_table, seq_lens, **kw):
    """Paged single-token decode: ggml_flash_attn_ext for one query per sequence
    (/root/reference/src/llama-graph.cpp:1187-1197).  q f32 [n_seq][n_head][D] -> out f32 [n_seq][n_head][D]."""
    return _attn(lib().pkv_attn_decode, q, k_pool, v_pool, block_table, seq_lens, **kw)


def rope_cache(pos, rope: RopeParams, head_dim: int, *, out=None, stream=None):
    """ggml_rope_cache_init for a micro-batch (pkv_rope_cache): f32 [n_tok][head_dim/2][2] (cos, sin) rows, computed once
    per step and shared by every layer's attn_decode_step."""
    import torch
    n_tok = pos.shape[0]
    out = torch.empty((n_tok, head_dim // 2, 2), dtype=torch.float32, device=pos.device) if out is None else out
    check(lib().pkv_rope_cache(C.c_void_p(_ptr(pos)), n_tok, head_dim, C.byref(rope), C.c_void_p(_ptr(out)),
                               C.c_void_p(_stream(stream))))
    return out


def attn_decode_step(q, k_cur, v_cur, pos, rope, k_pool, v_pool, block_table, seq_lens, *, rope_cache=None, **kw):
    """One decode step of a layer in one launch: RoPE(Q) + RoPE(K) + KV store + paged decode (pkv_attn_decode_step).
    q f32 [n_seq][n_head][D] un-rotated (left untouched), k_cur/v_cur f32 [n_seq][n_head_kv][D], pos i32 [n_seq];
    the new token of sequence s is cell seq_lens[s]-1.  rope=None: no RoPE.  rope_cache: rows from rope_cache(pos, rope, D),
    shared by all layers of the step (None: built per call)."""
    return _attn(lib().pkv_attn_decode_step, q, k_pool, v_pool, block_table, seq_lens,
                 step=(k_cur, v_cur, pos, rope, rope_cache), **kw)


DECODE_NSTAMP = 12


def attn_decode_stamps(q, k_cur, v_cur, pos, rope, k_pool, v_pool, block_table, seq_lens, *, rope_cache=None, mode=1, stamps_out=None, **kw):
    """DIAGNOSTIC (pkv_attn_decode_stamps): the decode step through the stamped instantiation of the kernel; returns
    (out-less) the u64 stamp tensor [n_seq * n_head_kv * n_splits][8][DECODE_NSTAMP].  Not a product path."""
    import torch
    n_seq = block_table.shape[0]
    n_splits = (kw["max_seq_len"] + 127) // 128
    st = stamps_out if stamps_out is not None else torch.zeros((n_seq * kw["n_head_kv"] * n_splits, 8, DECODE_NSTAMP), dtype=torch.int64, device=q.device)
    _attn(lib().pkv_attn_decode_stamps, q, k_pool, v_pool, block_table, seq_lens,
          step=(k_cur, v_cur, pos, rope, rope_cache), stamps=(st, mode), flags=kw.pop("flags", 0) | 1, **kw)
    return st


class KernelTimer:
    """pairs of raw hipEvent_t that `attn_decode(..., events=timer.pair(i))` binds to the decode launch itself"""

    def __init__(self, n: int):
        L = lib()
        L.pkv_event_create.restype = C.c_void_p
        L.pkv_event_destroy.restype = None
        self.ev = [(L.pkv_event_create(), L.pkv_event_create()) for _ in range(n)]
        if any(a is None or b is None for a, b in self.ev):
            raise PkvError(ERR_NO_DEVICE, "cannot create HIP events")

    def pair(self, i):
        return self.ev[i]

    def elapsed_ms(self):
        out = []
        for a, b in self.ev:
            check(lib().pkv_event_synchronize(C.c_void_p(b)))
            ms = C.c_float()
            check(lib().pkv_event_elapsed_ms(C.c_void_p(a), C.c_void_p(b), C.byref(ms)))
            out.append(ms.value)
        return out

    def close(self):
        for a, b in self.ev:
            lib().pkv_event_destroy(C.c_void_p(a))
            lib().pkv_event_destroy(C.c_void_p(b))
        self.ev = []


def attn_step_layers(q, k_cur, v_cur, out, k_pools, v_pools, pos, rope, block_table, seq_lens, *, rope_cache, n_head_kv, type_k, type_v, block_size, scale,
                     max_seq_len, workspace, q_start=None, max_q_per_seq=1, slot_block=None, slot_off=None, stream=None, flags=0):
    """pkv_attn_step_layers: one ubatch through all layers in ONE call — attn_decode_step per layer (slot_block None) or attn_prefill_step per layer.
    q / k_cur / v_cur / out: lists of per-layer device tensors (same shapes and strides in every layer); k_pools / v_pools: lists of per-layer pools."""
    import torch
    L = len(q)
    n_tok, n_head, D = q[0].shape
    n_seq, max_blocks = block_table.shape
    if q[0].dtype == torch.float16:
        flags |= ATTN_Q_F16
    if out[0].dtype == torch.float16:
        flags |= ATTN_OUT_F16
    a = AttnArgs(_ptr(q[0]), q[0].stride(0), q[0].stride(1), _ptr(out[0]), _ptr(k_pools[0]), _ptr(v_pools[0]), _ptr(block_table),
                 _ptr(seq_lens), _ptr(q_start), None, None, _ptr(workspace), workspace.numel(),
                 n_seq, n_tok, max_q_per_seq, n_head, n_head_kv, D, block_size, max_blocks, type_k, type_v,
                 scale, 0.0, max_seq_len, flags, None, 0, 0.0, None, None, None)
    st = StepArgs(_ptr(k_cur[0]), _ptr(v_cur[0]), _ptr(pos), C.pointer(rope) if rope is not None else None, _ptr(rope_cache))
    arr = lambda ts: (C.c_void_p * L)(*[_ptr(t) for t in ts])   # noqa: E731
    check(lib().pkv_attn_step_layers(C.byref(a), C.byref(st), L, arr(q), arr(k_cur), arr(v_cur), arr(out), arr(k_pools), arr(v_pools),
                                     C.c_void_p(_ptr(slot_block)), C.c_void_p(_ptr(slot_off)), C.c_void_p(_stream(stream))))
    return out


def attn_prefill(q, k_pool, v_pool, block_table, seq_lens, q_start, **kw):
    """Paged causal prefill (many query tokens per sequence); q f32 [n_tok][n_head][D]."""
    return _attn(lib().pkv_attn_prefill, q, k_pool, v_pool, block_table, seq_lens, q_start=q_start, **kw)


def flash_attn_ext_flat(q, k, v, mask=None, *, type_k: int, type_v: int, scale: float, max_bias: float = 0.0, softcap: float = 0.0, out=None, stream=None, split: bool = True,
                        workspace=None):
    """ggml_flash_attn_ext on the reference's own tensor layout (pkv_flash_attn_ext_flat): what the paged kernels do not take — head sizes up to 256, bf16
    rows, any gqa ratio, strided views.  q f32 [n_head][n_q][D] (any strides with a unit last one); k / v uint8 [n_head_kv][n_kv][row bytes] in the
    reference's bytes (views allowed: strides are taken from the tensors); mask f16 [>= n_q][n_kv] or None.  Returns out f32 [n_q][n_head][D]."""
    import torch
    n_head, n_q, D = q.shape
    n_head_kv, n_kv = k.shape[0], k.shape[1]
    assert q.dtype == torch.float32 and q.stride(2) == 1 and k.dtype == torch.uint8 and v.dtype == torch.uint8 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty((n_q, n_head, D), dtype=torch.float32, device=q.device)
    if mask is not None:
        assert mask.dtype == torch.float16 and mask.stride(1) == 1 and mask.shape[0] >= n_q and mask.shape[1] == n_kv
    a = FlatAttnArgs(q=_ptr(q), q_stride_tok=q.stride(1), q_stride_head=q.stride(0), k=_ptr(k), k_stride_tok=k.stride(1), k_stride_head=k.stride(0),
                     v=_ptr(v), v_stride_tok=v.stride(1), v_stride_head=v.stride(0), mask=_ptr(mask), mask_stride=mask.stride(0) if mask is not None else 0,
                     out=_ptr(out), n_q=n_q, n_kv=n_kv, n_head=n_head, n_head_kv=n_head_kv, head_dim=D, type_k=type_k, type_v=type_v, scale=scale,
                     max_bias=max_bias, logit_softcap=softcap)
    if split:                                        # split-KV for launches of few query rows over a long K / V (the library decides; needs the scratch)
        need = lib().pkv_flash_attn_ext_flat_workspace_bytes(n_q, n_head, D, n_kv)
        if need:
            ws = workspace if workspace is not None and workspace.numel() * workspace.element_size() >= need else torch.empty(need, dtype=torch.uint8, device=q.device)
            a.workspace, a.workspace_bytes = _ptr(ws), need
    check(lib().pkv_flash_attn_ext_flat(C.byref(a), C.c_void_p(_stream(stream))))
    return out


def attn_prefill_step(q, k_cur, v_cur, pos, rope, k_pool, v_pool, block_table, seq_lens, q_start, slot_block, slot_off, *, rope_cache=None, **kw):
    """A prefill ubatch of a layer: RoPE(K) + KV store into (slot_block, slot_off), then the prefill attention with RoPE(Q) fused into its Q
    fetch (pkv_attn_prefill_step): q stays un-rotated and untouched; same pools and output bits as qkv_prepare + attn_prefill.  rope_cache:
    rows from rope_cache(pos, rope, D) for the ubatch's tokens (required with rope; shared by all layers).  NEOX rope: PkvError (use the two calls)."""
    return _attn(lib().pkv_attn_prefill_step, q, k_pool, v_pool, block_table, seq_lens, q_start=q_start,
                 step=(k_cur, v_cur, pos, rope, rope_cache), slots=(slot_block, slot_off), **kw)


def k_shift(k_pool, *, type_k, cell_block, cell_off, cell_delta, n_head_kv, head_dim, block_size, rope: RopeParams,
            stream=None) -> None:
    """build_kv_self_shift (/root/reference/src/llama-context.cpp:524-571) for an explicit cell list."""
    check(lib().pkv_k_shift(C.c_void_p(_ptr(k_pool)), type_k, C.c_void_p(_ptr(cell_block)), C.c_void_p(_ptr(cell_off)),
                            C.c_void_p(_ptr(cell_delta)), cell_block.numel(), n_head_kv, head_dim, block_size,
                            C.byref(rope), C.c_void_p(_stream(stream))))


def copy_blocks(pool, *, type, src, dst, n_head_kv, head_dim, block_size, stream=None) -> None:
    check(lib().pkv_copy_blocks(C.c_void_p(_ptr(pool)), type, C.c_void_p(_ptr(src)), C.c_void_p(_ptr(dst)), src.numel(),
                                n_head_kv, head_dim, block_size, C.c_void_p(_stream(stream))))


def pool_scatter_rows(pool, rows, *, type, is_v, slot_block, slot_off, n_head_kv, head_dim, block_size, stream=None):
    """ggml_backend_tensor_set of reference-format KV rows (/root/reference/src/llama-kv-cache.cpp:1105-1221)."""
    check(lib().pkv_pool_scatter_rows(C.c_void_p(_ptr(pool)), type, int(is_v), C.c_void_p(_ptr(rows)),
                                      C.c_void_p(_ptr(slot_block)), C.c_void_p(_ptr(slot_off)), slot_block.numel(),
                                      n_head_kv, head_dim, block_size, C.c_void_p(_stream(stream))))


def pool_gather_rows(pool, rows, *, type, is_v, slot_block, slot_off, n_head_kv, head_dim, block_size, stream=None):
    """ggml_backend_tensor_get of reference-format KV rows (/root/reference/src/llama-kv-cache.cpp:904-1001)."""
    check(lib().pkv_pool_gather_rows(C.c_void_p(_ptr(pool)), type, int(is_v), C.c_void_p(_ptr(rows)),
                                     C.c_void_p(_ptr(slot_block)), C.c_void_p(_ptr(slot_off)), slot_block.numel(),
                                     n_head_kv, head_dim, block_size, C.c_void_p(_stream(stream))))
    return rows


def convert_f32(x, dst, type: int, stream=None):
    """GGML_OP_CPY f32 -> f16 / q8_0 / q4_0 rows in the reference's byte format (x [n_rows][ne0] f32, dst device bytes)"""
    check(lib().pkv_convert_f32(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(dst)), type, C.c_int64(x.numel() // x.shape[-1]), C.c_int64(x.shape[-1]),
                                C.c_void_p(_stream(stream))))
    return dst


def convert_f32_paged(x, dst, type: int, pool, *, is_v, cell0: int, n_head_kv: int, head_dim: int, block_size: int, stream=None):
    """pkv_convert_f32_paged: x [n_cells][n_head_kv][head_dim] f32 -> dst (reference bytes of those cells) AND the pool rows of cells
    cell0.. (identity placement), one launch"""
    n_cells = x.numel() // (n_head_kv * head_dim)
    check(lib().pkv_convert_f32_paged(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(dst)), type, C.c_void_p(_ptr(pool)), int(is_v), C.c_int64(cell0),
                                      C.c_int64(n_cells), n_head_kv, head_dim, block_size, C.c_void_p(_stream(stream))))
    return dst


def rope_convert_f32_paged(x, pos, rope: RopeParams, dst, type: int, pool, *, cell0: int, block_size: int, roped=None, stream=None):
    """pkv_rope_convert_f32_paged: RoPE of k_cur x [n_tok][n_head_kv][head_dim] (packed), then what convert_f32_paged stores (K layout)"""
    n_tok, n_head_kv, D = x.shape
    check(lib().pkv_rope_convert_f32_paged(C.c_void_p(_ptr(x)), C.c_int64(0), C.c_int64(0), C.c_void_p(_ptr(pos)), C.byref(rope), C.c_void_p(_ptr(roped)),
                                           C.c_void_p(_ptr(dst)), type, C.c_void_p(_ptr(pool)), C.c_int64(cell0), C.c_int64(n_tok), n_head_kv, D, block_size,
                                           C.c_void_p(_stream(stream))))
    return dst


def qkv_store_f32_paged(q, k, v, pos, rope: RopeParams, *, q_out, k_dst, k_pool, v_dst, v_pool, type_k: int, type_v: int, cell0: int, block_size: int,
                        stream=None):
    """pkv_qkv_store_f32_paged: q [n_tok][n_head][D], k / v [n_tok][n_head_kv][D] packed f32; RoPE(q) -> q_out, RoPE(k) + store, store(v)"""
    n_tok, n_head, D = q.shape
    a = QkvStoreArgs(_ptr(q), 0, 0, _ptr(q_out), _ptr(k), 0, 0, _ptr(k_dst), _ptr(k_pool), cell0, _ptr(v), _ptr(v_dst), _ptr(v_pool), cell0,
                     _ptr(pos), C.pointer(rope), type_k, type_v, n_tok, n_head, k.shape[1], D, block_size)
    check(lib().pkv_qkv_store_f32_paged(C.byref(a), C.c_void_p(_stream(stream))))
    return q_out


# ---- first slice of the rest of the decoder layer ----------------------------------------------------------
def rms_norm(x, eps: float, out=None, stream=None):
    """ggml_rms_norm over the last dim of a contiguous f32 tensor"""
    out = x.new_empty(x.shape) if out is None else out
    check(lib().pkv_rms_norm_f32(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), C.c_int64(x.numel() // x.shape[-1]), C.c_int64(x.shape[-1]),
                                 C.c_float(eps), C.c_void_p(_stream(stream))))
    return out


def rms_norm_mul(x, w, eps: float, stream=None):
    """rms_norm(x) * w in one launch (w: one row of x.shape[-1] weights)"""
    out = x.new_empty(x.shape)
    check(lib().pkv_rms_norm_mul_f32(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(w)), None, C.c_void_p(_ptr(out)), C.c_int64(x.numel() // x.shape[-1]),
                                     C.c_int64(x.shape[-1]), C.c_float(eps), C.c_void_p(_stream(stream))))
    return out


def add_rms_norm_mul(a, b, w, eps: float, in_place: bool = False, stream=None):
    """(a + b, rms_norm(a + b) * w) in one launch; in_place writes the sum over a (what ggml-alloc's in-place ADD does)"""
    s_ = a if in_place else a.new_empty(a.shape)
    out = a.new_empty(a.shape)
    check(lib().pkv_add_rms_norm_mul_f32(C.c_void_p(_ptr(a)), C.c_void_p(_ptr(b)), C.c_void_p(_ptr(s_)), C.c_void_p(_ptr(w)), None, C.c_void_p(_ptr(out)),
                                         C.c_int64(a.numel() // a.shape[-1]), C.c_int64(a.shape[-1]), C.c_float(eps), C.c_void_p(_stream(stream))))
    return s_, out


def silu_mul(g, u, stream=None):
    """silu(g) * u in one launch"""
    out = g.new_empty(g.shape)
    check(lib().pkv_silu_mul_f32(C.c_void_p(_ptr(g)), C.c_void_p(_ptr(u)), None, C.c_void_p(_ptr(out)), C.c_int64(g.numel()), C.c_void_p(_stream(stream))))
    return out


def _binary(fn, a, b, out, stream):
    out = a.new_empty(a.shape) if out is None else out
    ne0 = a.shape[-1]
    check(fn(C.c_void_p(_ptr(a)), C.c_void_p(_ptr(b)), C.c_void_p(_ptr(out)), C.c_int64(a.numel() // ne0), C.c_int64(ne0),
             C.c_int64(b.numel() // ne0), C.c_void_p(_stream(stream))))
    return out


def add(a, b, out=None, stream=None):
    """ggml_add: a + b, b's rows repeating over a's rows (same shape, residual, bias)"""
    return _binary(lib().pkv_add_f32, a, b, out, stream)


def mul(a, b, out=None, stream=None):
    """ggml_mul: a * b with the same row broadcast (norm weights)"""
    return _binary(lib().pkv_mul_f32, a, b, out, stream)


def silu(x, out=None, stream=None):
    out = x.new_empty(x.shape) if out is None else out
    check(lib().pkv_silu_f32(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), C.c_int64(x.numel()), C.c_void_p(_stream(stream))))
    return out


def _ggml_dims(t):
    """(ne[4], nb[4]) of a torch tensor in ggml's order: dimension 0 = the last torch dimension, strides in BYTES"""
    sh, st = list(t.shape)[::-1], [x * t.element_size() for x in t.stride()][::-1]
    while len(sh) < 4:
        sh.append(1)
        st.append(st[-1] * sh[-2] if st else t.element_size())
    return (C.c_int64 * 4)(*sh), (C.c_int64 * 4)(*st)


def _plain_type(t):
    import torch
    return {torch.float32: F32, torch.float16: F16}[t.dtype]


def cpy_strided(src, dst, stream=None):
    """ggml_cpy / ggml_cont between strided f32 / f16 tensors (any torch views) of equal element count"""
    ns, bs = _ggml_dims(src)
    nd, bd = _ggml_dims(dst)
    check(lib().pkv_cpy_strided(C.c_void_p(_ptr(src)), _plain_type(src), ns, bs, C.c_void_p(_ptr(dst)), _plain_type(dst), nd, bd, C.c_void_p(_stream(stream))))
    return dst


def mul_mat_strided(w, x, out=None, stream=None):
    """ggml_mul_mat on strided / batched / broadcast operands: w f16 / f32 [b3][b2][M][K], x f32 [B3][B2][N][K] (any torch views with unit stride in K)
    -> f32 [B3][B2][N][M] contiguous"""
    import torch
    w4 = w.reshape((1,) * (4 - w.dim()) + tuple(w.shape)) if w.dim() < 4 else w
    x4 = x.reshape((1,) * (4 - x.dim()) + tuple(x.shape)) if x.dim() < 4 else x
    out = torch.empty((x4.shape[0], x4.shape[1], x4.shape[2], w4.shape[2]), dtype=torch.float32, device=x.device) if out is None else out
    nw, bw = _ggml_dims(w4)
    nx, bx = _ggml_dims(x4)
    check(lib().pkv_mul_mat_strided(C.c_void_p(_ptr(w4)), _plain_type(w4), nw, bw, C.c_void_p(_ptr(x4)), nx, bx, C.c_void_p(_ptr(out)), C.c_void_p(_stream(stream))))
    return out


def soft_max_ext(x, mask=None, scale: float = 1.0, max_bias: float = 0.0, out=None, stream=None):
    """ggml_soft_max_ext over the last dimension of x f32 [ne3][ne2][ne1][ne0] (fewer dimensions: leading ones); mask f32 / f16 [>= ne1][ne0]"""
    import torch
    out = x.new_empty(x.shape) if out is None else out
    sh = (1,) * (4 - x.dim()) + tuple(x.shape)
    mf16 = int(mask is not None and mask.dtype == torch.float16)
    check(lib().pkv_soft_max_ext_f32(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(mask) if mask is not None else None), C.c_int(mf16), C.c_void_p(_ptr(out)),
                                     C.c_int64(sh[3]), C.c_int64(sh[2]), C.c_int64(sh[1]), C.c_int64(sh[0]), C.c_float(scale), C.c_float(max_bias),
                                     C.c_void_p(_stream(stream))))
    return out


def mul_mat(w, type_w: int, M: int, K: int, x, out=None, stream=None):
    """ggml_mul_mat: w = M rows of K elements in f32 / f16 / q8_0 reference row format (device bytes), x f32 [n_cols][K] ->
    f32 [n_cols][M]"""
    import torch
    n_cols = x.shape[0]
    out = torch.empty((n_cols, M), dtype=torch.float32, device=x.device) if out is None else out
    check(lib().pkv_mul_mat(C.c_void_p(_ptr(w)), type_w, C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), C.c_int64(M), C.c_int64(K),
                            C.c_int64(n_cols), C.c_void_p(_stream(stream))))
    return out


def mul_mat_multi(ws, type_w: int, Ms, K: int, x, stream=None):
    """pkv_mul_mat_multi: len(ws) products on the same activations x (wq / wk / wv, gate / up); q4_K weights run as one launch.
    Returns the list of f32 [n_cols][M_i] results."""
    import torch
    n, n_cols = len(ws), x.shape[0]
    outs = [torch.empty((n_cols, int(M)), dtype=torch.float32, device=x.device) for M in Ms]
    wp = (C.c_void_p * n)(*[_ptr(w) for w in ws])
    yp = (C.c_void_p * n)(*[_ptr(o) for o in outs])
    mp = (C.c_int64 * n)(*[int(M) for M in Ms])
    check(lib().pkv_mul_mat_multi(n, wp, type_w, C.c_void_p(_ptr(x)), yp, mp, C.c_int64(K), C.c_int64(n_cols), C.c_void_p(_stream(stream))))
    return outs


# ---- decode-step products for K-quant weights (pkv_gemv.hip) ---------------------------------------------------------------
def act_image(K: int, device="cuda"):
    """buffer for the int8-per-32 image of one activation column of K elements (pkv_act_image_bytes)"""
    import torch
    return torch.empty(lib().pkv_act_image_bytes(K), dtype=torch.uint8, device=device)


def quantize_act(x, image=None, stream=None):
    """pkv_quantize_act: x f32 [K] -> activation image"""
    K = x.numel()
    image = act_image(K, x.device) if image is None else image
    check(lib().pkv_quantize_act(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(image)), C.c_int64(K), C.c_void_p(_stream(stream))))
    return image


def rms_norm_mul_q(x, w, eps: float, image, *, res=None, x_out=None, y=None, stream=None):
    """pkv_rms_norm_mul_q: x_out = x + res (optional), image = quantize(rms_norm(x_out) * w), y (optional) = the f32 row"""
    check(lib().pkv_rms_norm_mul_q(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(res)), C.c_void_p(_ptr(x_out)), C.c_void_p(_ptr(w)), C.c_void_p(_ptr(y)),
                                   C.c_void_p(_ptr(image)), C.c_int64(x.numel()), C.c_float(eps), C.c_void_p(_stream(stream))))
    return image


def gemv_q(ws, types, Ms, K: int, image, outs, *, residual=None, silu_mul: bool = False, x=None, norm_w=None, eps: float = 0.0, stream=None, norm_out=None, col_out=None):
    """pkv_gemv_q: up to three K-quant products on one activation column in one launch; outs = list of f32 [M_i] tensors
    (silu_mul: outs[0] = silu(W0 x) * (W1 x), outs[1] unused; residual: outs[0] = residual + W0 x).  image = the column's activation image,
    or None: the kernel quantizes the f32 column `x` itself, after rms_norm(x) * norm_w when norm_w is given."""
    n = len(ws)
    g = Gemv()
    g.n, g.silu_mul, g.K, g.xq, g.residual = n, int(silu_mul), K, _ptr(image), _ptr(residual)
    g.x, g.norm_w, g.eps = _ptr(x), _ptr(norm_w), eps
    g.norm_out, g.col_out = _ptr(norm_out), _ptr(col_out)
    for i in range(n):
        g.type_w[i], g.w[i], g.y[i], g.M[i] = int(types[i]), _ptr(ws[i]), _ptr(outs[i] if i < len(outs) else outs[0]), int(Ms[i])
    check(lib().pkv_gemv_q(C.byref(g), C.c_void_p(_stream(stream))))
    return outs


def weights_pack(w, type_w: int, n_rows: int, K: int, *, inverse: bool = False, out=None, stream=None):
    """pkv_weights_pack: GGUF-order q4_K / q5_K / q6_K rows (device bytes) -> the plane layout the decode-step products stream (`type | PACKED`), or back
    (inverse=True, byte-exact).  Returns a new device byte tensor."""
    import torch
    out = torch.empty_like(w) if out is None else out
    check(lib().pkv_weights_pack(C.c_void_p(_ptr(w)), C.c_void_p(_ptr(out)), int(type_w) & ~PACKED, C.c_int64(n_rows), C.c_int64(K), C.c_int(int(inverse)),
                                 C.c_void_p(_stream(stream))))
    return out


def pack_rows_host(rows: np.ndarray, type_w: int, K: int, inverse: bool = False) -> np.ndarray:
    """host restatement of the plane layout (csrc/pkv_wlayout.h) for tests: rows = uint8 [n_rows][row_bytes] in GGUF order -> packed (or back)"""
    gpr = K // 64
    nb = K // 256
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n = rows.shape[0]
    if type_w == Q4_K:
        if not inverse:
            blk = rows.reshape(n, nb, 144)
            hdr = blk[:, :, :16].reshape(n, nb * 16)
            qs = blk[:, :, 16:].reshape(n, nb, 4, 2, 16)                    # [block][p][half][16]
            qa = qs[:, :, :, 0, :].reshape(n, gpr * 16)
            qb = qs[:, :, :, 1, :].reshape(n, gpr * 16)
            return np.concatenate([qa, qb, hdr], axis=1)
        qa = rows[:, :16 * gpr].reshape(n, nb, 4, 1, 16)
        qb = rows[:, 16 * gpr:32 * gpr].reshape(n, nb, 4, 1, 16)
        hdr = rows[:, 32 * gpr:].reshape(n, nb, 16)
        qs = np.concatenate([qa, qb], axis=3).reshape(n, nb, 128)
        return np.concatenate([hdr, qs], axis=2).reshape(n, nb * 144)
    if type_w == Q6_K:
        if not inverse:
            blk = rows.reshape(n, nb, 210)
            ql = blk[:, :, :128].reshape(n, nb, 2, 2, 2, 16)                 # [block][n][hi][is][16]
            la = ql[:, :, :, 0, :, :].reshape(n, gpr * 16)
            lb = ql[:, :, :, 1, :, :].reshape(n, gpr * 16)
            hh = blk[:, :, 128:192].reshape(n, gpr * 16)                     # [block][n][is][16] is already item order
            sc = blk[:, :, 192:208].reshape(n, nb * 16)
            d = blk[:, :, 208:210].reshape(n, nb * 2)
            return np.concatenate([la, lb, hh, sc, d], axis=1)
        la = rows[:, :16 * gpr].reshape(n, nb, 2, 1, 2, 16)
        lb = rows[:, 16 * gpr:32 * gpr].reshape(n, nb, 2, 1, 2, 16)
        hh = rows[:, 32 * gpr:48 * gpr].reshape(n, nb, 64)
        sc = rows[:, 48 * gpr:48 * gpr + 16 * nb].reshape(n, nb, 16)
        d = rows[:, 52 * gpr:].reshape(n, nb, 2)
        ql = np.concatenate([la, lb], axis=3).reshape(n, nb, 128)
        return np.concatenate([ql, hh, sc, d], axis=2).reshape(n, nb * 210)
    if type_w == Q5_K:
        if not inverse:
            blk = rows.reshape(n, nb, 176)
            hdr = blk[:, :, :16].reshape(n, nb * 16)
            ha = blk[:, :, 16:32].reshape(n, nb * 16)
            hb = blk[:, :, 32:48].reshape(n, nb * 16)
            qs = blk[:, :, 48:].reshape(n, nb, 4, 2, 16)                    # [block][p][half][16]
            qa = qs[:, :, :, 0, :].reshape(n, gpr * 16)
            qb = qs[:, :, :, 1, :].reshape(n, gpr * 16)
            return np.concatenate([qa, qb, ha, hb, hdr], axis=1)
        qa = rows[:, :16 * gpr].reshape(n, nb, 4, 1, 16)
        qb = rows[:, 16 * gpr:32 * gpr].reshape(n, nb, 4, 1, 16)
        ha = rows[:, 32 * gpr:36 * gpr].reshape(n, nb, 16)
        hb = rows[:, 36 * gpr:40 * gpr].reshape(n, nb, 16)
        hdr = rows[:, 40 * gpr:].reshape(n, nb, 16)
        qs = np.concatenate([qa, qb], axis=3).reshape(n, nb, 128)
        return np.concatenate([hdr, ha, hb, qs], axis=2).reshape(n, nb * 176)
    raise ValueError("pack_rows_host: q4_K / q5_K / q6_K only")


def get_rows(w, type_w: int, ids, K: int, n_rows_w: int, out=None, stream=None):
    """ggml_get_rows: out[i] = dequantize(row ids[i] of w) (pkv_get_rows); ids device i32 [n]"""
    import torch
    n = ids.numel()
    out = torch.empty((n, K), dtype=torch.float32, device=ids.device) if out is None else out
    check(lib().pkv_get_rows(C.c_void_p(_ptr(w)), type_w, C.c_void_p(_ptr(ids)), C.c_void_p(_ptr(out)), C.c_int64(n), C.c_int64(K), C.c_int64(n_rows_w),
                             C.c_void_p(_stream(stream))))
    return out


# ---- llama_kv_cache mirror ---------------------------------------------------------------------------------------------
class PagedKVCache:
    """`llama_kv_cache` method set (/root/reference/src/llama-kv-cache.h:17-26) over the C++ block manager.

    n_layer == 0 gives a bookkeeping-only cache that needs no GPU (used by the CPU test-suite)."""

    def __init__(self, *, n_blocks: int, block_size: int = 128, n_seq_max: int = 1, n_layer: int = 0, n_head_kv: int = 8,
                 head_dim: int = 128, type_k: int = F16, type_v: int = F16, device: int = 0, evict_when_full: bool = False):
        self.params = CacheParams(n_layer, n_head_kv, head_dim, type_k, type_v, block_size, n_blocks, n_seq_max, device,
                                  int(evict_when_full))
        self._h = lib().pkv_cache_create(C.byref(self.params))
        if not self._h:
            raise PkvError(ERR_INVALID, lib().pkv_last_error().decode())
        self.block_size, self.n_blocks, self.n_seq_max, self.n_layer = block_size, n_blocks, n_seq_max, n_layer

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().pkv_cache_free(C.c_void_p(self._h))
            except TypeError:          # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    __del__ = close

    @property
    def h(self):
        return C.c_void_p(self._h)

    # llama_memory_i
    def clear(self): lib().pkv_cache_clear(self.h)
    def defrag(self): lib().pkv_cache_defrag(self.h)
    def seq_rm(self, seq_id, p0, p1) -> bool: return bool(lib().pkv_cache_seq_rm(self.h, seq_id, p0, p1))
    def seq_cp(self, src, dst, p0, p1): lib().pkv_cache_seq_cp(self.h, src, dst, p0, p1)
    def seq_keep(self, seq_id): lib().pkv_cache_seq_keep(self.h, seq_id)
    def seq_add(self, seq_id, p0, p1, delta): lib().pkv_cache_seq_add(self.h, seq_id, p0, p1, delta)
    def seq_div(self, seq_id, p0, p1, d): lib().pkv_cache_seq_div(self.h, seq_id, p0, p1, d)
    def seq_pos_max(self, seq_id) -> int: return lib().pkv_cache_seq_pos_max(self.h, seq_id)
    def get_can_edit(self) -> bool: return self.get_can_shift()
    # llama_kv_cache
    def get_n_tokens(self) -> int: return lib().pkv_cache_n_tokens(self.h)
    def get_used_cells(self) -> int: return lib().pkv_cache_used_cells(self.h)
    def get_can_shift(self) -> bool: return bool(lib().pkv_cache_can_shift(self.h))

    def find_slot(self, seq_id: Sequence[int], pos: Sequence[int]):
        """llama_kv_cache_unified::find_slot for a ubatch; returns (block[n_tok], off[n_tok]) or raises PkvError(NO_SPACE)."""
        sid = np.ascontiguousarray(seq_id, np.int32)
        p = np.ascontiguousarray(pos, np.int32)
        blk, off = np.empty(len(sid), np.int32), np.empty(len(sid), np.int32)
        check(lib().pkv_cache_find_slot(self.h, len(sid), sid.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
                                        blk.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p)))
        return blk, off

    # block-table introspection
    def seq_len(self, s) -> int: return lib().pkv_cache_seq_len(self.h, s)
    def seq_is_regular(self, s) -> bool: return bool(lib().pkv_cache_seq_is_regular(self.h, s))
    def n_free_blocks(self) -> int: return lib().pkv_cache_n_free_blocks(self.h)
    def block_refcount(self, b) -> int: return lib().pkv_cache_block_refcount(self.h, int(b))
    def has_shift(self) -> bool: return bool(lib().pkv_cache_has_shift(self.h))

    def seq_blocks(self, s) -> np.ndarray:
        n = lib().pkv_cache_seq_n_blocks(self.h, s)
        out = np.empty(n, np.int32)
        lib().pkv_cache_seq_blocks(self.h, s, out.ctypes.data_as(C.c_void_p), n)
        return out

    def seq_slot_pos(self, s) -> np.ndarray:
        n = self.seq_len(s)
        out = np.empty(n, np.int32)
        lib().pkv_cache_seq_slot_pos(self.h, s, out.ctypes.data_as(C.c_void_p), n)
        return out

    def pending(self) -> np.ndarray:
        """queued device work: rows {is_cell_copy, src_block, src_off, dst_block, dst_off}"""
        n = lib().pkv_cache_n_pending(self.h)
        out = np.empty((n, 5), np.int32)
        lib().pkv_cache_pending(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def block_table(self, seq_ids: Sequence[int], max_blocks: Optional[int] = None):
        """(block_table[n_seq][max_blocks] padded with 0, seq_lens[n_seq]) for the given sequences (host arrays)."""
        tabs = [self.seq_blocks(s) for s in seq_ids]
        mb = max_blocks or max(1, max(len(t) for t in tabs))
        bt = np.zeros((len(tabs), mb), np.int32)
        for i, t in enumerate(tabs):
            bt[i, :len(t)] = t
        return bt, np.array([self.seq_len(s) for s in seq_ids], np.int32)

    def block_tables(self, seq_ids: Sequence[int], width: int):
        """(block_table[n_seq][width] padded with 0, seq_lens[n_seq], longest block list, all sequences regular) in ONE library call
        (pkv_cache_block_tables): a decode step of 32 sequences spent ~100 ctypes calls on block_table() + seq_is_regular()"""
        sid = np.ascontiguousarray(seq_ids, np.int32)
        bt = np.empty((len(sid), width), np.int32)
        sl = np.empty(len(sid), np.int32)
        reg = C.c_int32(1)
        mb = lib().pkv_cache_block_tables(self.h, len(sid), sid.ctypes.data_as(C.c_void_p), width, bt.ctypes.data_as(C.c_void_p),
                                           sl.ctypes.data_as(C.c_void_p), C.byref(reg))
        check(int(mb) if mb < 0 else 0)
        return bt, sl, int(mb), bool(reg.value)

    # device side
    def k_pool(self, layer: int) -> int: return lib().pkv_cache_k_pool(self.h, layer)
    def v_pool(self, layer: int) -> int: return lib().pkv_cache_v_pool(self.h, layer)

    # session state in the reference's byte format (llama_kv_cache_unified::state_write / state_read,
    # /root/reference/src/llama-kv-cache.cpp:848-1221); seq_id = -1: the whole cache
    def state_write(self, seq_id: int = -1, stream=None) -> bytes:
        n = lib().pkv_cache_state_size(self.h, seq_id)
        if n < 0:
            check(int(n))
        buf = C.create_string_buffer(max(int(n), 1))
        st = C.c_void_p(_stream(stream)) if self.n_layer > 0 else None
        w = lib().pkv_cache_state_write(self.h, seq_id, buf, C.c_size_t(int(n)), st)
        if w < 0:
            check(int(w))
        return buf.raw[:int(w)]

    def state_read(self, data: bytes, seq_id: int = -1, stream=None) -> None:
        st = C.c_void_p(_stream(stream)) if self.n_layer > 0 else None
        check(lib().pkv_cache_state_read(self.h, seq_id, data, C.c_size_t(len(data)), st))

    def view(self, n_seq_max: int = 4):
        """llama_kv_cache_view_update (/root/reference/src/llama-kv-cache.cpp:1364-1433) over the physical cells:
        dict(n_cells, token_count, used_cells, max_contiguous, max_contiguous_idx, cells=pos[n_cells], cells_sequences[n_cells][n_seq_max])"""
        n = lib().pkv_cache_view_update(self.h, n_seq_max, None, None, None)
        pos, seqs, st = np.empty(n, np.int32), np.empty((n, n_seq_max), np.int32), np.empty(5, np.int32)
        lib().pkv_cache_view_update(self.h, n_seq_max, pos.ctypes.data_as(C.c_void_p), seqs.ctypes.data_as(C.c_void_p),
                                    st.ctypes.data_as(C.c_void_p))
        return dict(n_cells=int(st[0]), token_count=int(st[1]), used_cells=int(st[2]), max_contiguous=int(st[3]),
                    max_contiguous_idx=int(st[4]), cells=pos, cells_sequences=seqs)

    def update(self, rope: Optional[RopeParams] = None, stream=None):
        """kv_self_update (/root/reference/src/llama-context.cpp:719-803): queued copies + pending K-shift."""
        st = C.c_void_p(_stream(stream)) if self.n_layer > 0 else None
        check(lib().pkv_cache_update(self.h, C.byref(rope) if rope is not None else None, st))


class ToyBlockManager:
    """The fork prototype's BlockManager (/root/reference/tests/test_block_manager.cpp)."""

    def __init__(self, n): self._h = lib().pkv_toy_bm_create(n); self.n = n
    def __del__(self): lib().pkv_toy_bm_free(C.c_void_p(self._h))
    def allocateBlock(self) -> int: return lib().pkv_toy_bm_allocate(C.c_void_p(self._h))
    def freeBlock(self, i): lib().pkv_toy_bm_release(C.c_void_p(self._h), i)

    def getBlockStatus(self):
        out = np.zeros(self.n, np.uint8)
        lib().pkv_toy_bm_status(C.c_void_p(self._h), out.ctypes.data_as(C.c_void_p), self.n)
        return [bool(x) for x in out]


class ToyKVCacheManager:
    """The fork prototype's KVCacheManager / PhysicalBlock (/root/reference/tests/test_kv_cache.cpp)."""

    def __init__(self, slot_capacity): self._h = lib().pkv_toy_kvm_create(slot_capacity)
    def __del__(self): lib().pkv_toy_kvm_free(C.c_void_p(self._h))

    def insert(self):
        b, s = C.c_int(), C.c_int()
        assert lib().pkv_toy_kvm_insert(C.c_void_p(self._h), C.byref(b), C.byref(s)) == 0
        return b.value, s.value

    def free(self, entry) -> bool:
        return lib().pkv_toy_kvm_release(C.c_void_p(self._h), entry[0], entry[1]) == 0
