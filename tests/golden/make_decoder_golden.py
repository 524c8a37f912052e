#!/usr/bin/env python3
"""Generate tests/golden/decoder_logits.npz: final logits of the REFERENCE's llama_decode (CPU backend, oracle/_ref/libref_kv.so) on a
synthetic llama GGUF whose K-quant tensors are REAL Gaussian weights quantized by the compiled reference's own quantizers
(from_float_ref of q4_K / q6_K) in the tensor-type mix of a Q4_K_M file — so that the own host loop (llama.cpp_test_amd/decoder.py,
tests/test_gpu_decoder.py) is held to the reference harness' bar (NMSE 5e-4) on end-to-end logits, not to a bar loosened for
random blocks.  Scenario: a 150-token prompt (crosses a 128-token block) as ONE llama_decode batch -> logits of its last
token, then 6 single-token decode steps -> logits after each.  f16 and q8_0 KV, flash attention on.

The GGUF is regenerated from the seed wherever oracle/_ref exists (this container, and the GPU box: oracle/_ref travels); the fixture
stores its SHA-256 so that a drift of the generator or of the quantizer is noticed.   Run:  python tests/golden/make_decoder_golden.py"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _load_pkg  # noqa: E402

_load_pkg.load()
from llama_cpp_test_amd import gguf_synth as G  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MODEL = dict(n_layer=2, n_embd=512, n_head=4, n_head_kv=2, n_ff=768, n_vocab=256, n_ctx=1024, rope_freq_base=500000.0, seed=2024, wtype=G.GGML_Q4_K, k_mix=True)
N_PROMPT, N_DECODE = 150, 6


def build(path):
    from oracle import cpu_oracle as O
    O.ref_rope(np.zeros((1, 1, 64), np.float32), np.zeros(1, np.int32), n_dims=64)      # any op: ggml_init fills the f16 table
    G.synth_llama(path, quantize=lambda t, w: O.ref_from_float_ref(t, w).tobytes(), **MODEL)
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def tokens():
    return np.random.default_rng(7).integers(0, MODEL["n_vocab"], N_PROMPT + N_DECODE).astype(np.int32)


def exact_logits(path, tok):
    """the same model in exact arithmetic: f64 everywhere, the STORED weights (dequantized as the reference dequantizes them), no activation
    quantization, no f16 anywhere — the yardstick that tells the reference's own rounding noise from an implementation error"""
    kv, tensors, _ = G.read_gguf(path)
    E, H, HK, L, eps = MODEL["n_embd"], MODEL["n_head"], MODEL["n_head_kv"], MODEL["n_layer"], 1e-5
    D = E // H

    def W(name):
        ne, t, _, data = tensors[name]
        raw = bytes(data)
        if t == G.GGML_Q4_K:
            return G.dequantize_q4_K(raw, ne[0]).astype(np.float64)
        if t == G.GGML_Q6_K:
            return G.dequantize_q6_K(raw, ne[0]).astype(np.float64)
        assert t == G.GGML_F32
        return np.frombuffer(raw, np.float32).astype(np.float64)
    w = {n: W(n) for n in tensors}
    n = len(tok)
    x = w["token_embd.weight"][tok]                                  # [n][E]
    pos = np.arange(n)
    inv = MODEL["rope_freq_base"] ** (-2.0 * np.arange(D // 2) / D)
    ang = pos[:, None] * inv[None, :]
    cos, sin = np.cos(ang), np.sin(ang)

    def rope(t):                                                     # [n][h][D], NORM pairs (2i, 2i+1)
        a, b = t[..., 0::2], t[..., 1::2]
        out = np.empty_like(t)
        out[..., 0::2] = a * cos[:, None, :] - b * sin[:, None, :]
        out[..., 1::2] = a * sin[:, None, :] + b * cos[:, None, :]
        return out

    def norm(t, wn):
        return t / np.sqrt((t ** 2).mean(-1, keepdims=True) + eps) * wn
    for l in range(L):
        p = f"blk.{l}."
        h = norm(x, w[p + "attn_norm.weight"])
        q = rope((h @ w[p + "attn_q.weight"].T).reshape(n, H, D))
        k = rope((h @ w[p + "attn_k.weight"].T).reshape(n, HK, D))
        v = (h @ w[p + "attn_v.weight"].T).reshape(n, HK, D)
        att = np.empty((n, H, D))
        for hh in range(H):
            s = (q[:, hh] @ k[:, hh // (H // HK)].T) / np.sqrt(D)
            s = np.where(pos[None, :] <= pos[:, None], s, -np.inf)
            pr = np.exp(s - s.max(-1, keepdims=True))
            att[:, hh] = (pr / pr.sum(-1, keepdims=True)) @ v[:, hh // (H // HK)]
        x = x + att.reshape(n, E) @ w[p + "attn_output.weight"].T
        h = norm(x, w[p + "ffn_norm.weight"])
        g = h @ w[p + "ffn_gate.weight"].T
        x = x + ((g / (1 + np.exp(-g))) * (h @ w[p + "ffn_up.weight"].T)) @ w[p + "ffn_down.weight"].T
    return (norm(x, w["output_norm.weight"]) @ w["output.weight"].T)[N_PROMPT - 1:].astype(np.float32)


def main():
    from oracle import ref_model as R
    if not R.available():
        raise SystemExit("oracle/_ref/libref_kv.so missing: run `make -C oracle ref`")
    path = os.path.join(tempfile.mkdtemp(), "synthetic-llama-q4km.gguf")
    sha = build(path)
    tok = tokens()
    flat = {"sha256": np.frombuffer(bytes.fromhex(sha), np.uint8), "tokens": tok, "exact__logits": exact_logits(path, tok)}
    for name, (tk, tv) in {"f16": (1, 1), "q8_0": (8, 8)}.items():
        m = R.RefModel(path, n_ctx=512, n_batch=256, n_seq_max=1, flash_attn=True, type_k=tk, type_v=tv, n_threads=8)
        m.decode(tok[:1], [0], [0])                                  # one token, one cell: no attention rounding to amplify
        flat[f"{name}__first_logits"] = m.logits(MODEL["n_vocab"])
        m.seq_rm(0, -1, -1)
        m.decode(tok[:N_PROMPT], np.arange(N_PROMPT), np.zeros(N_PROMPT, np.int32))
        lg = [m.logits(MODEL["n_vocab"])]
        for i in range(N_DECODE):
            m.decode(tok[N_PROMPT + i:N_PROMPT + i + 1], [N_PROMPT + i], [0])
            lg.append(m.logits(MODEL["n_vocab"]))
        flat[f"{name}__logits"] = np.stack(lg)
        ex = flat["exact__logits"].astype(np.float64)
        print(name, "reference vs exact arithmetic, NMSE per step:", " ".join("%.2e" % (((a - b) ** 2).sum() / (b ** 2).sum()) for a, b in zip(np.stack(lg).astype(np.float64), ex)))
        del m
    np.savez_compressed(os.path.join(OUT, "decoder_logits.npz"), **flat)
    print("wrote decoder_logits.npz; gguf", os.path.getsize(path) // 1024, "KiB, sha256", sha[:16])


if __name__ == "__main__":
    main()
