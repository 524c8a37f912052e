"""Synthetic random-weight GGUF writer + minimal loader.

File format = GGUF v3 as the reference reads and writes it (/root/reference/ggml/include/gguf.h:1-30, 54-67; reader
/root/reference/ggml/src/gguf.cpp): magic "GGUF", u32 version, i64 n_tensors, i64 n_kv; KV pairs {string key, u32 type,
value}; tensor infos {string name, u32 n_dims, i64 ne[n_dims], u32 ggml_type, u64 offset}; padding to
`general.alignment` (default 32); tensor data, each tensor padded to the alignment.  Strings are u64 length + bytes.

`synth_llama()` writes a llama-architecture model with random weights (keys as /root/reference/src/llama-model.cpp:406-521
reads them, tensors as :1585-1625 creates them, tokenizer "no_vocab", /root/reference/src/llama-vocab.cpp:1358) so that the
reference's own loader and this package read the same file.  There is no network and no real checkpoint in this build:
every weight is seeded noise, as BASELINE.md prescribes.  Host-side only (numpy); nothing here touches the GPU.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Tuple

import numpy as np

MAGIC, VERSION, DEFAULT_ALIGNMENT = b"GGUF", 3, 32
# gguf_type (gguf.h:54-67)
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f", T_BOOL: "<?", T_U64: "<Q",
           T_I64: "<q", T_F64: "<d"}
# ggml_type ids used here (/root/reference/ggml/include/ggml.h:351-392)
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q8_0, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K = 0, 1, 2, 8, 12, 13, 14
_BLOCK = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_Q4_0: (32, 18), GGML_Q8_0: (32, 34), GGML_Q4_K: (256, 144), GGML_Q5_K: (256, 176), GGML_Q6_K: (256, 210)}   # (elements, bytes) per block


def row_bytes(t: int, ne0: int) -> int:
    n, b = _BLOCK[t]
    assert ne0 % n == 0
    return ne0 // n * b


def _s(x: str) -> bytes:
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def _kv(key: str, typ: int, val) -> bytes:
    out = _s(key) + struct.pack("<I", typ)
    if typ == T_STR:
        return out + _s(val)
    if typ == T_ARR:
        et, items = val
        out += struct.pack("<IQ", et, len(items))
        return out + (b"".join(_s(v) for v in items) if et == T_STR else b"".join(struct.pack(_SCALAR[et], v) for v in items))
    return out + struct.pack(_SCALAR[typ], val)


def quantize_q8_0(x: np.ndarray) -> bytes:
    """quantize_row_q8_0_ref (/root/reference/ggml/src/ggml-quants.c:194-215): d = amax/127 as f16, q = roundf(x / d)"""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 32)
    amax = np.abs(x).max(axis=1)
    d = (amax / 127.0).astype(np.float32)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    y = x * inv[:, None]
    q = np.trunc(y + np.copysign(np.float32(0.5), y)).astype(np.int8)          # roundf: half away from zero
    out = np.empty((x.shape[0], 34), np.uint8)
    out[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.tobytes()


def random_q4_K(rng, n_rows: int, ne0: int) -> bytes:
    """random but well-formed block_q4_K rows (/root/reference/ggml/src/ggml-common.h:279-290): synthetic weights need no quantizer,
    only blocks whose dequantized values (d * sc * q - dmin * m, dequantize_row_q4_K) look like N(0, 1/ne0) weights"""
    nb = n_rows * (ne0 // 256)
    out = np.empty((nb, 144), np.uint8)
    d = (rng.uniform(0.7, 1.3, nb) / (np.sqrt(ne0) * 145.0)).astype(np.float16)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = (d.astype(np.float32) * 7.5).astype(np.float16).view(np.uint8).reshape(nb, 2)   # dmin * E[m] cancels d * E[sc] * E[q]
    out[:, 4:16] = rng.integers(0, 256, (nb, 12), dtype=np.uint8)
    out[:, 16:] = rng.integers(0, 256, (nb, 128), dtype=np.uint8)
    return out.tobytes()


def random_q5_K(rng, n_rows: int, ne0: int) -> bytes:
    """random but well-formed block_q5_K rows (/root/reference/ggml/src/ggml-common.h:291-308: d, dmin, scales[12], qh[32], qs[128]): dequantized values
    d * sc * q - dmin * m with q in 0..31 (dequantize_row_q5_K) look like N(0, 1/ne0) weights"""
    nb = n_rows * (ne0 // 256)
    out = np.empty((nb, 176), np.uint8)
    d = (rng.uniform(0.7, 1.3, nb) / (np.sqrt(ne0) * 290.0)).astype(np.float16)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = (d.astype(np.float32) * 15.5).astype(np.float16).view(np.uint8).reshape(nb, 2)   # dmin * E[m] cancels d * E[sc] * E[q]
    out[:, 4:] = rng.integers(0, 256, (nb, 172), dtype=np.uint8)
    return out.tobytes()


def dequantize_q5_K(raw, ne0: int) -> np.ndarray:
    """dequantize_row_q5_K (/root/reference/ggml/src/ggml-quants.c:1482-1507) in numpy: rows of ne0 f32"""
    b = np.frombuffer(raw, np.uint8).reshape(-1, 176)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0]
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32)[:, 0]
    sc, qh, qs = b[:, 4:16].astype(np.int32), b[:, 16:48].astype(np.int32), b[:, 48:]
    out = np.empty((b.shape[0], 256), np.float32)
    for j in range(8):                                            # get_scale_min_k4
        if j < 4:
            s_, m_ = sc[:, j] & 63, sc[:, j + 4] & 63
        else:
            s_, m_ = (sc[:, j + 4] & 0xF) | ((sc[:, j - 4] >> 6) << 4), (sc[:, j + 4] >> 4) | ((sc[:, j] >> 6) << 4)
        q = qs[:, 32 * (j // 2):32 * (j // 2) + 32].astype(np.int32)
        q = ((q & 0xF) if j % 2 == 0 else (q >> 4)) + (((qh >> j) & 1) << 4)
        d1, m1 = d * s_.astype(np.float32), dmin * m_.astype(np.float32)                       # f32 products, as the reference
        out[:, 32 * j:32 * j + 32] = d1[:, None] * q.astype(np.float32) - m1[:, None]
    return out.reshape(-1, ne0)


def random_q6_K(rng, n_rows: int, ne0: int) -> bytes:
    """random but well-formed block_q6_K rows (/root/reference/ggml/src/ggml-common.h:309-319: ql[128] low nibbles, qh[64] high 2 bits, 16 int8
    scales, f16 d): dequantized values d * sc * (q - 32) look like N(0, 1/ne0) weights (q uniform 0..63: std 18.5, |sc| ~ 64)"""
    nb = n_rows * (ne0 // 256)
    out = np.empty((nb, 210), np.uint8)
    out[:, 0:192] = rng.integers(0, 256, (nb, 192), dtype=np.uint8)
    out[:, 192:208] = (rng.integers(40, 100, (nb, 16)) * rng.choice([-1, 1], (nb, 16))).astype(np.int8).view(np.uint8)
    d = (rng.uniform(0.7, 1.3, nb) / (np.sqrt(ne0) * 18.5 * 70.0)).astype(np.float16)
    out[:, 208:210] = d.view(np.uint8).reshape(nb, 2)
    return out.tobytes()


def dequantize_q6_K(raw, ne0: int) -> np.ndarray:
    """dequantize_row_q6_K (/root/reference/ggml/src/ggml-quants.c:1690-1719) in numpy: rows of ne0 f32"""
    b = np.frombuffer(raw, np.uint8).reshape(-1, 210)
    d = b[:, 208:210].copy().view(np.float16).astype(np.float32)[:, 0]
    ql, qh, sc = b[:, 0:128].astype(np.int32), b[:, 128:192].astype(np.int32), b[:, 192:208].copy().view(np.int8).astype(np.float32)
    out = np.empty((b.shape[0], 256), np.float32)
    for n in range(2):
        for j in range(4):
            lq = ql[:, 64 * n + 32 * (j & 1):64 * n + 32 * (j & 1) + 32]
            lq = (lq >> 4) if (j >> 1) else (lq & 0xF)
            q = (lq | (((qh[:, 32 * n:32 * n + 32] >> (2 * j)) & 3) << 4)) - 32
            s = sc[:, 8 * n + 2 * j:8 * n + 2 * j + 2].repeat(16, axis=1)          # is = l / 16
            out[:, 128 * n + 32 * j:128 * n + 32 * j + 32] = (d[:, None] * s) * q.astype(np.float32)   # d * sc * q: left to right, f32
    return out.reshape(-1, ne0)


def dequantize_q4_K(raw, ne0: int) -> np.ndarray:
    """dequantize_row_q4_K (/root/reference/ggml/src/ggml-quants.c:1280-1302) in numpy: rows of ne0 f32"""
    b = np.frombuffer(raw, np.uint8).reshape(-1, 144)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0]
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32)[:, 0]
    sc, qs = b[:, 4:16].astype(np.int32), b[:, 16:]
    out = np.empty((b.shape[0], 256), np.float32)
    for j in range(8):                                            # get_scale_min_k4
        if j < 4:
            s_, m_ = sc[:, j] & 63, sc[:, j + 4] & 63
        else:
            s_, m_ = (sc[:, j + 4] & 0xF) | ((sc[:, j - 4] >> 6) << 4), (sc[:, j + 4] >> 4) | ((sc[:, j] >> 6) << 4)
        q = qs[:, 32 * (j // 2):32 * (j // 2) + 32]
        q = (q & 0xF) if j % 2 == 0 else (q >> 4)
        d1, m1 = d * s_.astype(np.float32), dmin * m_.astype(np.float32)                       # f32 products, as the reference
        out[:, 32 * j:32 * j + 32] = d1[:, None] * q.astype(np.float32) - m1[:, None]
    return out.reshape(-1, ne0)


def encode(x: np.ndarray, t: int) -> bytes:
    if t == GGML_F32:
        return np.ascontiguousarray(x, np.float32).tobytes()
    if t == GGML_F16:
        return np.ascontiguousarray(x, np.float32).astype(np.float16).tobytes()
    if t == GGML_Q8_0:
        return quantize_q8_0(x)
    raise ValueError(f"unsupported tensor type {t}")


def write_gguf(path: str, kv: List[Tuple[str, int, object]], tensors: List[Tuple[str, Tuple[int, ...], int, bytes]],
               alignment: int = DEFAULT_ALIGNMENT) -> None:
    """kv: (key, gguf_type, value); tensors: (name, ne (ggml order, ne[0] fastest), ggml_type, raw bytes)"""
    if alignment != DEFAULT_ALIGNMENT:
        kv = list(kv) + [("general.alignment", T_U32, alignment)]
    head = MAGIC + struct.pack("<Iqq", VERSION, len(tensors), len(kv)) + b"".join(_kv(*e) for e in kv)
    infos, off = [], 0
    for name, ne, t, data in tensors:
        assert len(data) == int(np.prod(ne[1:], dtype=np.int64)) * row_bytes(t, ne[0]), name
        infos.append(_s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<q", n) for n in ne) + struct.pack("<IQ", t, off))
        off += -(-len(data) // alignment) * alignment
    meta = head + b"".join(infos)
    with open(path, "wb") as f:
        f.write(meta)
        f.write(b"\0" * (-len(meta) % alignment))
        for _, _, _, data in tensors:
            f.write(data)
            f.write(b"\0" * (-len(data) % alignment))


def read_gguf(path: str):
    """minimal loader: (kv dict key -> (gguf_type, value), tensors dict name -> (ne, ggml_type, offset, np.uint8 view), alignment)"""
    buf = np.memmap(path, np.uint8, "r")
    raw = memoryview(buf)
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, raw, pos)
        pos += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    def rs():
        nonlocal pos
        n = rd("<Q")
        s = bytes(raw[pos:pos + n]).decode()
        pos += n
        return s
    if bytes(raw[:4]) != MAGIC:
        raise ValueError("not a GGUF file")
    pos = 4
    version, n_tensors, n_kv = rd("<Iqq")
    if version != VERSION:
        raise ValueError(f"GGUF version {version} not supported")
    kv: Dict[str, Tuple[int, object]] = {}
    for _ in range(n_kv):
        key, typ = rs(), rd("<I")
        if typ == T_STR:
            val = rs()
        elif typ == T_ARR:
            et, n = rd("<IQ")
            val = (et, [rs() if et == T_STR else rd(_SCALAR[et]) for _ in range(n)])
        else:
            val = rd(_SCALAR[typ])
        kv[key] = (typ, val)
    infos = []
    for _ in range(n_tensors):
        name, nd = rs(), rd("<I")
        ne = tuple(rd("<q") for _ in range(nd))
        t, off = rd("<IQ")
        infos.append((name, ne, t, off))
    alignment = kv.get("general.alignment", (T_U32, DEFAULT_ALIGNMENT))[1]
    data0 = -(-pos // alignment) * alignment
    tensors = {}
    for name, ne, t, off in infos:
        n = int(np.prod(ne[1:], dtype=np.int64)) * row_bytes(t, ne[0])
        tensors[name] = (ne, t, off, buf[data0 + off:data0 + off + n])
    return kv, tensors, alignment


def synth_llama(path: str, *, n_layer: int, n_embd: int, n_head: int, n_head_kv: int, n_ff: int, n_vocab: int, n_ctx: int = 4096,
                rope_freq_base: float = 10000.0, rms_eps: float = 1e-5, wtype: int = GGML_F16, seed: int = 1234, name: str = "synthetic-llama",
                rope_freqs: bool = False, k_mix: bool = False, quantize=None):
    """random-weight llama-arch GGUF (norm weights f32 near 1, matrices N(0, 1/sqrt(fan_in)) in `wtype`); rope_freqs adds the
    Llama-3.1-style `rope_freqs.weight` table of per-pair frequency factors (/root/reference/src/llama-model.cpp:1625, :11431-11443).
    k_mix (with wtype q4_K): the tensor-type mix of a Q4_K_M file — output.weight, and attn_v / ffn_down of the first and last eighth of
    the layers and of every third layer in between, in q6_K (llama_tensor_get_type + use_more_bits, /root/reference/src/llama-quant.cpp:130-340).
    quantize (optional): callable (ggml_type, f32 rows) -> bytes that quantizes REAL Gaussian weights (tests pass the compiled
    reference's from_float_ref); without it K-quant tensors are random well-formed blocks of the same statistics."""
    rng = np.random.default_rng(seed)
    hd = n_embd // n_head
    kv = [("general.architecture", T_STR, "llama"), ("general.name", T_STR, name), ("llama.context_length", T_U32, n_ctx),
          ("llama.embedding_length", T_U32, n_embd), ("llama.block_count", T_U32, n_layer), ("llama.feed_forward_length", T_U32, n_ff),
          ("llama.attention.head_count", T_U32, n_head), ("llama.attention.head_count_kv", T_U32, n_head_kv),
          ("llama.rope.dimension_count", T_U32, hd), ("llama.rope.freq_base", T_F32, rope_freq_base),
          ("llama.attention.layer_norm_rms_epsilon", T_F32, rms_eps), ("llama.vocab_size", T_U32, n_vocab),
          ("tokenizer.ggml.model", T_STR, "no_vocab")]
    tensors = []

    def more_bits(il):                          # use_more_bits(i_layer, n_layer)
        return il < n_layer // 8 or il >= 7 * n_layer // 8 or (il - n_layer // 8) % 3 == 2

    def mat(nm, ne0, ne1, q6=False):            # ggml [ne0 = in, ne1 = out]; numpy array [ne1][ne0]
        if wtype in (GGML_Q4_K, GGML_Q5_K) and ne0 % 256 == 0:     # (wtype q5_K + k_mix: the tensor mix of a Q5_K_M file, same rule with q5_K as the base type)
            t = GGML_Q6_K if (k_mix and q6) else wtype
            if quantize is not None:
                w = (rng.standard_normal((ne1, ne0)) / np.sqrt(ne0)).astype(np.float32)
                tensors.append((nm, (ne0, ne1), t, bytes(quantize(t, w))))
            else:
                tensors.append((nm, (ne0, ne1), t, {GGML_Q6_K: random_q6_K, GGML_Q5_K: random_q5_K, GGML_Q4_K: random_q4_K}[t](rng, ne1, ne0)))
            return
        t = GGML_F16 if wtype in (GGML_Q4_K, GGML_Q5_K) else wtype          # rows that are not a multiple of 256 stay f16
        w = (rng.standard_normal((ne1, ne0)) / np.sqrt(ne0)).astype(np.float32)
        tensors.append((nm, (ne0, ne1), t, encode(w, t)))

    def vec(nm, n):
        tensors.append((nm, (n,), GGML_F32, encode(1.0 + 0.1 * rng.standard_normal(n), GGML_F32)))
    mat("token_embd.weight", n_embd, n_vocab)
    vec("output_norm.weight", n_embd)
    mat("output.weight", n_embd, n_vocab, q6=True)
    if rope_freqs:   # factors 1 (high frequencies) ... 8 (low frequencies), the shape of the llama-3 long-context table
        ff = np.clip(8.0 * (np.arange(hd // 2) / (hd // 2 - 1)) ** 3, 1.0, 8.0).astype(np.float32)
        tensors.append(("rope_freqs.weight", (hd // 2,), GGML_F32, encode(ff, GGML_F32)))
    for il in range(n_layer):
        p = f"blk.{il}."
        vec(p + "attn_norm.weight", n_embd)
        mat(p + "attn_q.weight", n_embd, n_embd)
        mat(p + "attn_k.weight", n_embd, hd * n_head_kv)
        mat(p + "attn_v.weight", n_embd, hd * n_head_kv, q6=more_bits(il))
        mat(p + "attn_output.weight", n_embd, n_embd)
        vec(p + "ffn_norm.weight", n_embd)
        mat(p + "ffn_gate.weight", n_embd, n_ff)
        mat(p + "ffn_down.weight", n_ff, n_embd, q6=more_bits(il))
        mat(p + "ffn_up.weight", n_embd, n_ff)
    write_gguf(path, kv, tensors)
    return kv, tensors


def describe(path: str) -> str:
    """canonical text dump (one line per KV / tensor, data as FNV-1a 64) — the parity tests compare it with the same dump
    produced through the reference's gguf API"""
    kv, tensors, _ = read_gguf(path)
    lines = []
    for k, (t, v) in kv.items():
        if t == T_ARR:
            v = f"[{v[0]}:{len(v[1])}] " + ",".join(_fmt(v[0], x) for x in v[1])
        else:
            v = _fmt(t, v)
        lines.append(f"kv {k} {t} {v}")
    for name, (ne, t, off, data) in tensors.items():
        lines.append(f"tensor {name} {t} {'x'.join(str(n) for n in ne)} {off} {fnv1a64(data):016x}")
    return "\n".join(lines) + "\n"


def _fmt(t: int, v) -> str:
    if t == T_F32:
        return "%.9g" % float(np.float32(v))
    if t == T_F64:
        return "%.17g" % float(v)
    if t == T_BOOL:
        return "1" if v else "0"
    return str(v)


def fnv1a64(data) -> int:
    h = 0xcbf29ce484222325
    a = np.asarray(data, np.uint8)
    # vectorised over 8-byte lanes would change the hash; files here are small, so plain chunks through Python ints
    for b in a.tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h
