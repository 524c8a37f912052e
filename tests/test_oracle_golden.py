"""The CPU restatement (oracle/pkv_oracle.c) against the golden vectors the reference's own compiled CPU path
produced (tests/golden/make_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import numpy as np

from conftest import cases_of


def test_flash_attn_ext_matches_reference(O, golden):
    g = golden["flash_attn_ext"]
    for name in cases_of(g):
        D, nh, nkv, n_kv, n_q, tk, tv = (int(x) for x in g[f"{name}__meta"])
        out = O.flash_attn_ext(g[f"{name}__q"], g[f"{name}__k"], g[f"{name}__v"], g[f"{name}__mask"], type_k=tk, type_v=tv,
                               scale=D ** -0.5, softcap=float(g[f"{name}__softcap"]))
        ref = g[f"{name}__out"]
        # same evaluation order as the reference's AVX2 build -> bit-exact on x86; 2e-6 allows another libm's expf
        assert np.abs(out - ref).max() <= 2e-6, name
        # and the reference itself sits within f16-accumulation noise of exact arithmetic
        e = O.flash_attn_f64(g[f"{name}__q"], g[f"{name}__k"], g[f"{name}__v"], g[f"{name}__mask"], type_k=tk, type_v=tv,
                             scale=D ** -0.5, softcap=float(g[f"{name}__softcap"]))
        assert np.abs(ref - e).max() < 1e-3, name


def test_rope_matches_reference(O, golden):
    g = golden["rope"]
    for name in cases_of(g):
        p = g[f"{name}__params"]
        ff = g[f"{name}__ff"] if f"{name}__ff" in g.files else None
        out = O.rope(g[f"{name}__x"], g[f"{name}__pos"], ff, n_dims=int(p[0]), mode=int(p[1]), n_ctx_orig=int(p[2]),
                     freq_base=p[3], freq_scale=p[4], ext_factor=p[5], attn_factor=p[6], beta_fast=p[7], beta_slow=p[8])
        assert np.abs(out - g[f"{name}__out"]).max() <= 1e-6, name


def test_rope_composition(O):
    """tests/test-rope.cpp:140-180: rope(rope(x, p0), p1) == rope(x, p0 + p1)"""
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (4, 2, 128)).astype(np.float32)
    p0, p1 = np.array([3, 100, 7, 2000], np.int32), np.array([5, 28, 1, 47], np.int32)
    for mode in (0, 2):
        a = O.rope(O.rope(x, p0, n_dims=128, mode=mode), p1, n_dims=128, mode=mode)
        b = O.rope(x, p0 + p1, n_dims=128, mode=mode)
        assert np.abs(a - b).max() < 2e-4
        inv = O.rope(O.rope(x, p0, n_dims=128, mode=mode), p0, n_dims=128, mode=mode, inverse=True)
        assert np.abs(inv - x).max() < 1e-5


def test_kv_codecs_bit_exact(O, golden):
    g = golden["cpy"]
    x = g["x"]
    for t in (O.F16, O.Q8_0, O.Q4_0):
        n = O.TYPE_NAMES[t]
        b = O.from_float(t, x)
        assert np.array_equal(b, g[f"cpy_{n}"]), n          # byte-for-byte what ggml_cpy stores on the CPU backend
        assert np.array_equal(O.to_float(t, b, 128), g[f"deq_{n}"]), n
    assert np.array_equal(O.from_float(O.Q8_0, x, "ref"), g["q8_0_ref"])


def test_quantize_error_bounds(O):
    """tests/test-quantize-fns.cpp:17-54: total quantization error sqrt(sum d^2)/n < 0.002 on 0.1 + 2cos(i), n = 32*128"""
    n = 32 * 128
    x = (0.1 + 2.0 * np.cos(np.arange(n, dtype=np.float32))).astype(np.float32)[None]
    for t in (O.Q8_0, O.Q4_0, O.F16):
        y = O.to_float(t, O.from_float(t, x), n)
        assert np.sqrt(np.sum((x.astype(np.float64) - y) ** 2)) / n < 0.002


def test_pool_layout_roundtrip(O):
    rng = np.random.default_rng(0)
    for t in (O.F16, O.Q8_0, O.Q4_0):
        for D in (64, 128):
            for is_v in (0, 1):
                pool = O.pool_alloc(t, 5, 2, 128, D)
                rows = O.from_float(t, rng.uniform(-1, 1, (300, 2, D)).astype(np.float32))
                bt = np.array([4, 0, 2], np.int32)
                idx = np.arange(300)
                O.pool_store(pool, t, is_v, rows, bt[idx // 128], idx % 128, D=D, block_size=128)
                got = O.pool_gather(pool, t, is_v, bt, 300, D=D, block_size=128)
                assert np.array_equal(got, rows.transpose(1, 0, 2))
                assert not pool[1].any() and not pool[3].any()   # untouched blocks stay zero


def test_paged_equals_contiguous(O):
    """paged(K_pool, table) == contiguous flash_attn_ext on the de-paged K/V"""
    rng = np.random.default_rng(5)
    D, nh, nkv, bs = 128, 8, 2, 128
    lens = [1, 127, 128, 129, 300]
    nb = sum(-(-n // bs) for n in lens)
    perm = rng.permutation(nb + 3).astype(np.int32)
    for tk, tv in ((O.F16, O.F16), (O.Q8_0, O.Q8_0), (O.Q4_0, O.Q8_0)):
        kp, vp = O.pool_alloc(tk, nb + 3, nkv, bs, D), O.pool_alloc(tv, nb + 3, nkv, bs, D)
        bt = np.zeros((len(lens), 3), np.int32)
        q = rng.uniform(-1, 1, (len(lens), nh, 1, D)).astype(np.float32)
        expect, cur = [], 0
        for s, n in enumerate(lens):
            m = -(-n // bs)
            bt[s, :m] = perm[cur:cur + m]
            cur += m
            kr = O.from_float(tk, rng.uniform(-1, 1, (n, nkv, D)).astype(np.float32))
            vr = O.from_float(tv, rng.uniform(-1, 1, (n, nkv, D)).astype(np.float32))
            idx = np.arange(n)
            O.pool_store(kp, tk, 0, kr, bt[s][idx // bs], idx % bs, D=D, block_size=bs)
            O.pool_store(vp, tv, 1, vr, bt[s][idx // bs], idx % bs, D=D, block_size=bs)
            expect.append(O.flash_attn_ext(q[s], kr.transpose(1, 0, 2), vr.transpose(1, 0, 2), None, type_k=tk, type_v=tv,
                                           scale=D ** -0.5))
        got = O.attn_paged(q, kp, vp, bt, np.array(lens, np.int32), n_q=1, type_k=tk, type_v=tv, block_size=bs,
                           scale=D ** -0.5)
        for s in range(len(lens)):
            assert np.array_equal(got[s], expect[s])
