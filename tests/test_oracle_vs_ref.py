"""Pins the CPU restatement against the reference itself, compiled in place (oracle/_ref).  Skipped where the
compiled reference is not present (it is built in the dev container and travels prebuilt to the GPU box)."""
import numpy as np
import pytest

from oracle import cpu_oracle as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("tk,tv", [(O.F16, O.F16), (O.Q8_0, O.Q8_0), (O.Q4_0, O.Q4_0), (O.Q8_0, O.F16), (O.F16, O.Q4_0), (O.BF16, O.BF16), (O.BF16, O.Q8_0)])
@pytest.mark.parametrize("D,nh,nkv,n_kv,n_q", [(128, 8, 2, 512, 1), (128, 4, 4, 257, 7), (64, 8, 1, 300, 2), (256, 4, 2, 200, 3), (96, 4, 4, 130, 2)])
def test_flash_attn_bit_exact(tk, tv, D, nh, nkv, n_kv, n_q):
    rng = np.random.default_rng(n_kv * 31 + tk * 7 + tv)
    q = rng.uniform(-1, 1, (nh, n_q, D)).astype(np.float32)
    k = O.from_float(tk, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    v = O.from_float(tv, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    mask = np.zeros((n_q, n_kv), np.float16)
    for j in range(n_q):
        mask[j, n_kv - n_q + j + 1:] = -np.inf
    mask[:, rng.integers(0, n_kv // 2, 5)] = -np.inf   # holes, as after seq_rm
    for cap in (0.0, 8.0):
        a = O.flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, softcap=cap, n_threads=3)
        b = O.ref_flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, softcap=cap, n_threads=2)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tk,tv", [(O.Q4_1, O.Q4_1), (O.Q5_0, O.Q5_0), (O.Q5_1, O.Q5_1), (O.IQ4_NL, O.IQ4_NL), (O.Q5_1, O.IQ4_NL), (O.Q4_1, O.F16), (O.F16, O.Q5_0)])
@pytest.mark.parametrize("D,nh,nkv,n_kv,n_q", [(128, 8, 2, 300, 2), (64, 4, 4, 257, 3), (256, 4, 2, 130, 1), (96, 4, 2, 70, 2)])
def test_flash_attn_remaining_cache_types_bit_exact(tk, tv, D, nh, nkv, n_kv, n_q):
    """the other CLI-whitelisted cache types (common/arg.cpp:172-182): q4_1 / q5_1 meet Q as q8_1, q5_0 / iq4_nl as q8_0 (type_traits_cpu); head size 96 = three
    blocks: iq4_nl's two-blocks-at-a-time loop leaves one to its scalar tail"""
    rng = np.random.default_rng(n_kv * 13 + tk * 5 + tv)
    q = rng.uniform(-1, 1, (nh, n_q, D)).astype(np.float32)
    k = O.from_float(tk, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    v = O.from_float(tv, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    mask = np.zeros((n_q, n_kv), np.float16)
    for j in range(n_q):
        mask[j, n_kv - n_q + j + 1:] = -np.inf
    for kw in (dict(), dict(softcap=8.0), dict(max_bias=8.0)):
        a = O.flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, n_threads=3, **kw)
        b = O.ref_flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, n_threads=2, **kw)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tk", [O.F16, O.BF16])
def test_flash_attn_head_size_80_bit_exact(tk):
    """head size 80 (tests/test-backend-ops.cpp:4367): not a multiple of the 32-element steps of the vector dots, the leftovers run in double"""
    rng = np.random.default_rng(80 + tk)
    D, nh, nkv, n_kv, n_q = 80, 4, 4, 300, 3
    q = rng.uniform(-1, 1, (nh, n_q, D)).astype(np.float32)
    k = O.from_float(tk, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    v = O.from_float(tk, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    mask = np.zeros((n_q, n_kv), np.float16)
    mask[:, 280:] = -np.inf
    for mb in (0.0, 8.0):
        a = O.flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tk, scale=D ** -0.5, max_bias=mb, n_threads=3)
        b = O.ref_flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tk, scale=D ** -0.5, max_bias=mb, n_threads=2)
        assert np.array_equal(a, b)


def test_codecs_bit_exact():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((257, 128)) * rng.choice([1e-3, 1.0, 50.0], (257, 1))).astype(np.float32)
    x[::17] = np.round(x[::17] * 4) / 4
    x[9] = -0.0                                      # blocks of negative zeros (q4_0 / q5_0: d = +0 / -8 = -0, the quants of -0 * 0)
    x[3, :4] = [np.nan, np.inf, -np.inf, 0.0]
    for t in (O.F16, O.BF16, O.Q8_0, O.Q4_0, O.Q4_1, O.Q5_0, O.Q5_1, O.IQ4_NL):
        if t not in (O.F16, O.BF16):
            x[3, :4] = 0.0
        got, want = O.from_float(t, x), O.ref_cpy(t, x, n_threads=2)
        if t == O.IQ4_NL:                                # (all-zero iq4_nl blocks: the reference's indices are whatever its scratch held)
            got[9], want[9] = 0, 0
        assert np.array_equal(got, want)
        b = O.from_float(t, x)
        assert np.array_equal(O.to_float(t, b, 128).view(np.uint32), O.ref_to_float(t, b, 128).view(np.uint32))     # (bits: the row holds a NaN)
    assert np.array_equal(O.from_float(O.Q8_0, x, "ref"), O.ref_from_float_ref(O.Q8_0, x))


@pytest.mark.parametrize("kw", [dict(), dict(freq_base=10000.0, mode=2), dict(freq_scale=0.5, ext_factor=1.0, attn_factor=0.9,
                                                                            n_ctx_orig=2048, freq_base=10000.0)])
def test_rope_bit_exact(kw):
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (9, 8, 128)).astype(np.float32)
    pos = np.array([0, 1, 2, 63, 64, 1000, 4095, 8191, 32767], np.int32)
    ff = rng.uniform(1, 4, 64).astype(np.float32)
    for f in (None, ff):
        assert np.array_equal(O.rope(x, pos, f, n_dims=128, **kw), O.ref_rope(x, pos, f, n_dims=128, **kw))
