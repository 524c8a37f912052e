#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own compiled CPU path (oracle/_ref).

Run in the dev container only (needs /root/reference to have been compiled by `make -C oracle ref`):
    python tests/golden/make_golden.py
Each fixture stores the seeded inputs AND the outputs the reference produced, so consumers never depend on
numpy's RNG or on the reference being present.  ISA of the producing build: x86-64 AVX2+FMA+F16C (oracle/Makefile).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def causal_mask(n_q, n_kv):
    m = np.zeros((n_q, n_kv), np.float16)
    for j in range(n_q):
        m[j, n_kv - n_q + j + 1:] = -np.inf
    return m


def gen_flash_attn():
    rng = np.random.default_rng(1234)
    cases = {}
    specs = [  # name, D, n_head, n_head_kv, n_kv, n_q, type_k, type_v, softcap
        ("f16_d128_g4_q1", 128, 8, 2, 260, 1, O.F16, O.F16, 0.0),
        ("f16_d128_g4_q5", 128, 8, 2, 200, 5, O.F16, O.F16, 0.0),
        ("q8_0_d128_g4_q1", 128, 8, 2, 260, 1, O.Q8_0, O.Q8_0, 0.0),
        ("q8_0_d128_g2_q3", 128, 4, 2, 130, 3, O.Q8_0, O.Q8_0, 0.0),
        ("q4_0_d128_g4_q1", 128, 8, 2, 260, 1, O.Q4_0, O.Q4_0, 0.0),
        ("q8k_f16v_d128_g8_q1", 128, 8, 1, 140, 1, O.Q8_0, O.F16, 0.0),
        ("f16_d64_g8_q1", 64, 8, 1, 200, 1, O.F16, O.F16, 0.0),     # TinyLlama-like head size
        ("q8_0_d64_g1_q2", 64, 2, 2, 150, 2, O.Q8_0, O.Q8_0, 0.0),
        ("f16_d128_g1_softcap", 128, 2, 2, 100, 1, O.F16, O.F16, 10.0),
    ]
    for name, D, nh, nkv, n_kv, n_q, tk, tv, cap in specs:
        q = rng.uniform(-1, 1, (nh, n_q, D)).astype(np.float32)
        k = O.ref_cpy(tk, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
        v = O.ref_cpy(tv, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
        mask = causal_mask(n_q, n_kv)
        out = O.ref_flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, softcap=cap)
        for key, val in (("q", q), ("k", k), ("v", v), ("mask", mask), ("out", out),
                         ("meta", np.array([D, nh, nkv, n_kv, n_q, tk, tv], np.int32)), ("softcap", np.float32(cap))):
            cases[f"{name}__{key}"] = val
    np.savez_compressed(os.path.join(OUT, "flash_attn_ext.npz"), **cases)


def gen_rope():
    rng = np.random.default_rng(4321)
    cases = {}
    ff = rng.uniform(1.0, 8.0, 64).astype(np.float32)
    specs = [  # name, ne0, n_head, pos, kwargs
        ("llama3_norm", 128, 8, [0, 1, 2, 127, 128, 4095, 4096, 16383], dict(n_dims=128)),
        ("llama3_ff", 128, 4, [0, 3, 511, 8191], dict(n_dims=128, ff=ff)),
        ("yarn", 128, 4, [0, 7, 4095, 20000], dict(n_dims=128, freq_base=10000.0, freq_scale=0.25, ext_factor=1.0,
                                                   attn_factor=1.2, n_ctx_orig=4096)),
        ("neox", 128, 4, [0, 5, 1000], dict(n_dims=128, mode=2, freq_base=10000.0)),
        ("tiny_d64", 64, 4, [0, 9, 2047], dict(n_dims=64, freq_base=10000.0)),
        ("partial_dims", 128, 2, [0, 11, 333], dict(n_dims=64, freq_base=10000.0)),
    ]
    for name, ne0, nh, pos, kw in specs:
        x = rng.uniform(-1, 1, (len(pos), nh, ne0)).astype(np.float32)
        kw = dict(kw)
        f = kw.pop("ff", None)
        out = O.ref_rope(x, np.array(pos, np.int32), f, **kw)
        cases[f"{name}__x"] = x
        cases[f"{name}__pos"] = np.array(pos, np.int32)
        cases[f"{name}__out"] = out
        if f is not None:
            cases[f"{name}__ff"] = f
        full = dict(n_dims=kw["n_dims"], mode=kw.get("mode", 0), n_ctx_orig=kw.get("n_ctx_orig", 8192),
                    freq_base=kw.get("freq_base", 500000.0), freq_scale=kw.get("freq_scale", 1.0),
                    ext_factor=kw.get("ext_factor", 0.0), attn_factor=kw.get("attn_factor", 1.0),
                    beta_fast=kw.get("beta_fast", 32.0), beta_slow=kw.get("beta_slow", 1.0))
        cases[f"{name}__params"] = np.array([full[k] for k in ("n_dims", "mode", "n_ctx_orig", "freq_base", "freq_scale",
                                                               "ext_factor", "attn_factor", "beta_fast", "beta_slow")],
                                            np.float64)
    np.savez_compressed(os.path.join(OUT, "rope.npz"), **cases)


def gen_cpy():
    rng = np.random.default_rng(99)
    x = rng.uniform(-1, 1, (48, 128)).astype(np.float32)
    x[0] = 0.0                        # all-zero block (d == 0 branch)
    x[1, :32] = np.linspace(-4, 4, 32)  # exact .5 ties after scaling
    x[2] *= 1e-4
    x[3] *= 300.0
    x[4, 5] = -x[4, 6] if abs(x[4, 6]) > abs(x[4, 5]) else x[4, 5]
    x[5, 0], x[5, 1] = 0.75, -0.75    # |max| tie, first wins (q4_0)
    x[5, 2:32] *= 0.5
    cases = {"x": x}
    for t in (O.F16, O.Q8_0, O.Q4_0):
        b = O.ref_cpy(t, x)
        cases[f"cpy_{O.TYPE_NAMES[t]}"] = b
        cases[f"deq_{O.TYPE_NAMES[t]}"] = O.ref_to_float(t, b, 128)
    cases["q8_0_ref"] = O.ref_from_float_ref(O.Q8_0, x)
    np.savez_compressed(os.path.join(OUT, "cpy.npz"), **cases)


if __name__ == "__main__":
    assert O.ref_available(), "build the reference first: make -C oracle ref"
    gen_flash_attn()
    gen_rope()
    gen_cpy()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
