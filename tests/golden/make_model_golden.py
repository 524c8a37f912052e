#!/usr/bin/env python3
"""Generate tests/golden/model_attn.npz: the attention path's inputs and outputs inside a REAL llama_decode of the reference.

A synthetic random-weight llama GGUF (llama.cpp_test_amd/gguf_synth.py: 2 layers, head_dim 64, gqa 2) is loaded by the
reference's own loader and run on its CPU backend (oracle/_ref/libref_kv.so, driver oracle/ref_model_driver.cpp).  Through
cparams.cb_eval the driver captures, per layer and per llama_decode call, the un-rotated Q/K/V projections
(/root/reference/src/llama-model.cpp:4059-4090) and kqv_out (/root/reference/src/llama-graph.cpp:1432).  Scenario: a
130-token prompt (crosses a 128-token block) + a 40-token prompt in ONE batch, three interleaved decode steps, a 20-token
second chunk for sequence 1, one more decode step — with f16 KV and with q8_0 KV (flash attention on).

Run in the dev container only:   python tests/golden/make_model_golden.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _load_pkg  # noqa: E402

_load_pkg.load()
from llama_cpp_test_amd import gguf_synth as G  # noqa: E402
from oracle import ref_model as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MODEL = dict(n_layer=2, n_embd=128, n_head=2, n_head_kv=1, n_ff=256, n_vocab=64, n_ctx=1024, rope_freq_base=10000.0, seed=4321)


def scenario():
    """list of (seq_ids, positions) per llama_decode call"""
    calls = [([0] * 130 + [1] * 40, list(range(130)) + list(range(40)))]
    n0, n1 = 130, 40
    for _ in range(3):
        calls.append(([0, 1], [n0, n1]))
        n0 += 1; n1 += 1
    calls.append(([1] * 20, list(range(n1, n1 + 20))))
    n1 += 20
    calls.append(([1, 0], [n1, n0]))
    return calls


def run(path, type_k, type_v, flash_attn, rng):
    from oracle import cpu_oracle as O
    m = R.RefModel(path, n_ctx=512, n_batch=256, n_seq_max=2, flash_attn=flash_attn, type_k=type_k, type_v=type_v, n_threads=8)
    out, store = {}, {}
    for ci, (sid, pos) in enumerate(scenario()):
        tok = rng.integers(0, MODEL["n_vocab"], len(sid))
        layers = m.decode(tok, pos, sid)
        out[f"c{ci}__seq"], out[f"c{ci}__pos"] = np.array(sid, np.int32), np.array(pos, np.int32)
        H, HK, D = MODEL["n_head"], MODEL["n_head_kv"], MODEL["n_embd"] // MODEL["n_head"]
        for il, d in layers.items():
            n = len(sid)
            out[f"c{ci}__l{il}__q"] = d["q"].reshape(n, H, D)
            out[f"c{ci}__l{il}__k"] = d["k"].reshape(n, HK, D)
            out[f"c{ci}__l{il}__v"] = d["v"].reshape(n, HK, D)
            out[f"c{ci}__l{il}__out"] = d["out"].reshape(n, H, D)
            # yardstick: exact (f64) attention over the cells the cache really holds (rope -> stored type -> dequantized), and a
            # check that the CPU oracle reproduces the model's output bit for bit
            pos_a = np.array(pos, np.int32)
            kw = dict(n_dims=D, freq_base=MODEL["rope_freq_base"], n_ctx_orig=MODEL["n_ctx"])
            qr = O.rope(out[f"c{ci}__l{il}__q"], pos_a, **kw)
            kr = O.from_float(type_k, O.rope(out[f"c{ci}__l{il}__k"], pos_a, **kw))
            vr = O.from_float(type_v, out[f"c{ci}__l{il}__v"])
            exact = np.empty((n, H, D), np.float32)
            for i, s_ in enumerate(sid):
                ks, vs = store.setdefault((il, int(s_)), ([], []))
                ks.append(kr[i]); vs.append(vr[i])
                kc, vc = np.stack(ks).transpose(1, 0, 2), np.stack(vs).transpose(1, 0, 2)
                exact[i] = O.flash_attn_f64(qr[i][:, None, :], kc, vc, None, type_k=type_k, type_v=type_v, scale=D ** -0.5)[0]
                same = O.flash_attn_ext(qr[i][:, None, :], kc, vc, None, type_k=type_k, type_v=type_v, scale=D ** -0.5)[0]
                assert np.array_equal(same, out[f"c{ci}__l{il}__out"][i]), "the CPU oracle no longer reproduces llama_decode bit for bit"
            out[f"c{ci}__l{il}__exact"] = exact
    return out


def main():
    if not R.available():
        raise SystemExit("oracle/_ref/libref_kv.so missing: run `make -C oracle ref`")
    path = os.path.join(tempfile.mkdtemp(), "synthetic-llama.gguf")
    G.synth_llama(path, **MODEL)
    flat = {"n_calls": np.array(len(scenario()))}
    for name, (tk, tv) in {"f16": (1, 1), "q8_0": (8, 8)}.items():
        for k, v in run(path, tk, tv, True, np.random.default_rng(99)).items():
            flat[f"{name}__{k}"] = v
    np.savez_compressed(os.path.join(OUT, "model_attn.npz"), **flat)
    print("wrote model_attn.npz,", sum(v.nbytes for v in flat.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
