"""-m gpu: pkv_flash_attn_ext_flat — GGML_OP_FLASH_ATTN_EXT on the reference's own tensor layout, for the part of the operator's domain the paged
kernels do not take (r03): head sizes 80 / 96 / 256 (and 64 / 128 with bf16 rows), bf16 K / V, odd gqa ratios, strided (permuted) K / V views, ALiBi,
softcap, general masks.  The oracle (bit-exact against the compiled reference for exactly these shapes and types, tests/test_oracle_vs_ref.py) is the
checker: max-abs <= 1e-3 and NMSE <= 5e-4 (tests/test-backend-ops.cpp:3216); for f16 V the reference's f16 accumulator is itself up to ~1e-3 off exact
arithmetic on long rows, so — as for the paged decode kernel — the bar is "1e-3, or as far as the reference itself is from f64 (+3e-4)", and never
further from f64 than twice the reference."""
import numpy as np
import pytest
import torch

from gpu_util import dev, nmse

pytestmark = pytest.mark.gpu
T = {"f16": 1, "bf16": 30, "q8_0": 8, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "iq4_nl": 20}


def _case(O, rng, D, nh, nkv, n_kv, n_q, tk, tv, masked=True):
    q = rng.uniform(-1, 1, (nh, n_q, D)).astype(np.float32)
    k = O.from_float(tk, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    v = O.from_float(tv, rng.uniform(-1, 1, (nkv, n_kv, D)).astype(np.float32))
    mask = None
    if masked:
        mask = rng.uniform(-1, 1, (n_q, n_kv)).astype(np.float16)          # finite entries are additive (test-backend-ops fills the mask with noise)
        for j in range(n_q):
            mask[j, max(1, n_kv - n_q + j + 1):] = -np.inf                  # causal tail
        mask[:, rng.integers(0, max(1, n_kv // 2), 5)] = -np.inf            # holes, as after seq_rm
        mask[:, 0] = 0
    return q, k, v, mask


def _check(O, got, q, k, v, mask, tk, tv, D, **kw):
    ref = O.flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, n_threads=8, **kw)
    assert np.isfinite(got).all()
    assert nmse(got, ref) <= 5e-4
    # the yardstick: how far the reference itself is from exact arithmetic on these rows (its f16 accumulator for f16 V); ALiBi only shifts the logits,
    # the f64 restatement has no slopes: the plain run's distance serves there
    exact = O.flash_attn_f64(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, softcap=kw.get("softcap", 0.0))
    plain = ref if kw.get("max_bias", 0.0) == 0.0 else O.flash_attn_ext(q, k, v, mask, type_k=tk, type_v=tv, scale=D ** -0.5, n_threads=8)
    ref_err = np.abs(plain - exact).max()
    assert np.abs(got - ref).max() <= max(1e-3, ref_err + 3e-4), (np.abs(got - ref).max(), ref_err)
    if kw.get("max_bias", 0.0) == 0.0:
        assert np.abs(got - exact).max() <= max(2.0 * ref_err, 2e-4)


@pytest.mark.parametrize("tk,tv", [("f16", "f16"), ("bf16", "bf16"), ("q8_0", "q8_0"), ("q4_0", "q4_0"), ("bf16", "q8_0"), ("q4_0", "f16")])
@pytest.mark.parametrize("D,nh,nkv,n_kv,n_q", [(256, 4, 4, 512, 1), (256, 16, 4, 1024, 3), (256, 8, 2, 515, 35), (128, 4, 4, 512, 32), (64, 12, 4, 300, 2),
                                               (96, 6, 2, 257, 5), (32, 4, 1, 65, 1)])
def test_flat_matches_oracle(pkv, O, tk, tv, D, nh, nkv, n_kv, n_q):
    tk, tv = T[tk], T[tv]
    rng = np.random.default_rng(D * 7 + n_kv + tk + 3 * tv)
    q, k, v, mask = _case(O, rng, D, nh, nkv, n_kv, n_q, tk, tv)
    for kw in (dict(), dict(softcap=10.0), dict(max_bias=8.0)):
        got = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=tk, type_v=tv, scale=D ** -0.5, **kw)
        torch.cuda.synchronize()
        _check(O, got.cpu().numpy(), q, k, v, mask, tk, tv, D, **kw)
    got = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), None, type_k=tk, type_v=tv, scale=D ** -0.5)       # no mask: every cell is seen
    torch.cuda.synchronize()
    _check(O, got.cpu().numpy(), q, k, v, None, tk, tv, D)


@pytest.mark.parametrize("tk", ["f16", "bf16"])
def test_head_size_80(pkv, O, tk):
    """tests/test-backend-ops.cpp:4367: 80 is no multiple of the 32-element blocks, so only the 16-bit types exist at this size"""
    tk = T[tk]
    rng = np.random.default_rng(80 + tk)
    D, nh, nkv, n_kv, n_q = 80, 4, 4, 1024, 3
    q, k, v, mask = _case(O, rng, D, nh, nkv, n_kv, n_q, tk, tk)
    for kw in (dict(), dict(max_bias=8.0)):
        got = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=tk, type_v=tk, scale=D ** -0.5, **kw)
        torch.cuda.synchronize()
        _check(O, got.cpu().numpy(), q, k, v, mask, tk, tk, D, **kw)
    with pytest.raises(pkv.PkvError):                                       # quantized rows need whole blocks
        pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), None, type_k=8, type_v=8, scale=1.0)


@pytest.mark.parametrize("tk", ["bf16", "q8_0"])
def test_permuted_views(pkv, O, tk):
    """K / V as [n_kv][n_head_kv] rows (the layout of a cache tensor, and test-backend-ops' permute {0, 2, 1, 3}) seen through strides; q likewise"""
    tk = T[tk]
    rng = np.random.default_rng(5 + tk)
    D, nh, nkv, n_kv, n_q = 128, 8, 2, 333, 4
    q, k, v, mask = _case(O, rng, D, nh, nkv, n_kv, n_q, tk, tk)
    kd = dev(np.ascontiguousarray(k.transpose(1, 0, 2))).permute(1, 0, 2)   # memory [n_kv][n_head_kv][row], logical [n_head_kv][n_kv][row]
    vd = dev(np.ascontiguousarray(v.transpose(1, 0, 2))).permute(1, 0, 2)
    qd = dev(np.ascontiguousarray(q.transpose(1, 0, 2))).permute(1, 0, 2)
    a = pkv.flash_attn_ext_flat(qd, kd, vd, dev(mask), type_k=tk, type_v=tk, scale=D ** -0.5)
    b = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=tk, type_v=tk, scale=D ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    _check(O, a.cpu().numpy(), q, k, v, mask, tk, tk, D)


def test_row_without_visible_cell_and_bad_arguments(pkv, O):
    rng = np.random.default_rng(1)
    D, nh, nkv, n_kv, n_q = 64, 2, 1, 70, 2
    q, k, v, mask = _case(O, rng, D, nh, nkv, n_kv, n_q, 30, 30)
    mask[1, :] = -np.inf
    got = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=30, type_v=30, scale=D ** -0.5).cpu().numpy()
    assert np.array_equal(got[1], np.zeros_like(got[1])) and np.isfinite(got).all()
    ref = O.flash_attn_ext(q, k, v, mask, type_k=30, type_v=30, scale=D ** -0.5)
    assert np.abs(got[0] - ref[0]).max() <= 1e-3
    with pytest.raises(pkv.PkvError):
        pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), None, type_k=30, type_v=30, scale=1.0, max_bias=8.0)     # ALiBi without a mask
    with pytest.raises(pkv.PkvError):
        pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), None, type_k=12, type_v=30, scale=1.0)                    # q4_K is no KV type


@pytest.mark.parametrize("tk,tv", [("q4_1", "q4_1"), ("q5_0", "q5_0"), ("q5_1", "q5_1"), ("iq4_nl", "iq4_nl"), ("q5_1", "iq4_nl"), ("q4_1", "f16"), ("bf16", "q5_0")])
@pytest.mark.parametrize("D,nh,nkv,n_kv,n_q", [(128, 8, 2, 515, 3), (64, 4, 4, 300, 1), (256, 4, 2, 257, 2), (96, 6, 2, 130, 33)])
def test_remaining_cache_types(pkv, O, tk, tv, D, nh, nkv, n_kv, n_q):
    """q4_1 / q5_0 / q5_1 / iq4_nl (the rest of /root/reference/common/arg.cpp:172-182): K rows meet Q as q8_1 (q4_1, q5_1: + m_K * s_Q per block) or q8_0, V rows are
    dequantized as dequantize_row_* does; the oracle is bit-exact against the compiled reference for these types (tests/test_oracle_vs_ref.py)"""
    tk, tv = T[tk], T[tv]
    rng = np.random.default_rng(D + n_kv + 11 * tk + tv)
    q, k, v, mask = _case(O, rng, D, nh, nkv, n_kv, n_q, tk, tv)
    for kw in (dict(), dict(softcap=10.0), dict(max_bias=8.0)):
        got = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=tk, type_v=tv, scale=D ** -0.5, **kw)
        torch.cuda.synchronize()
        _check(O, got.cpu().numpy(), q, k, v, mask, tk, tv, D, **kw)


@pytest.mark.parametrize("t", ["q4_1", "q5_0", "q5_1", "iq4_nl"])
def test_kv_store_of_the_remaining_cache_types_is_bit_exact(pkv, O, t):
    """pkv_convert_f32 = GGML_OP_CPY f32 -> type, the KV store of such a cache: the same bytes as the reference's quantizers (the oracle's are bit-exact against
    oracle/_ref): values at three scales, exact ties, constant blocks; (all-zero iq4_nl blocks are left out: the reference leaves their indices undefined)"""
    t = T[t]
    rng = np.random.default_rng(40 + t)
    x = (rng.standard_normal((513, 128)) * rng.choice([1e-3, 1.0, 50.0], (513, 1))).astype(np.float32)
    x[::17] = np.round(x[::17] * 4) / 4
    x[5] = 0.75
    x[6, :32] = -2.0
    x[7, 32:64] = np.arange(32, dtype=np.float32) - 16
    x[8] = 0.0                                                       # all-zero blocks: q5_0 stores d = 0 / -16 = -0
    x[9] = -0.0
    want = O.from_float(t, x)
    dst = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
    pkv.convert_f32(dev(x), dst, t)
    torch.cuda.synchronize()
    got = dst.cpu().numpy().reshape(want.shape)
    if t == 20:                                                      # iq4_nl: all-zero blocks are not compared (see above); their scale must still be 0
        nz = np.abs(x.reshape(-1, 32)).max(axis=1) > 0
        w, g = want.reshape(-1, 18), got.reshape(-1, 18)
        assert np.array_equal(g[nz], w[nz]) and not g[~nz][:, :2].any() and (~nz).sum() >= 8
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("tk", ["bf16", "q8_0"])
def test_split_kv_of_a_decode_shaped_launch(pkv, O, tk):
    """few query rows over a long K / V: the launch is cut along the cells (partials + merge launch) when the caller lends a workspace — same result as the unsplit
    launch up to the order of the additions, same bars against the oracle; a range that is masked out entirely contributes nothing"""
    tk = T[tk]
    rng = np.random.default_rng(70 + tk)
    D, nh, nkv, n_kv, n_q = 128, 8, 2, 4099, 1
    q, k, v, mask = _case(O, rng, D, nh, nkv, n_kv, n_q, tk, tk)
    mask[:, 1024:2048] = -np.inf                                     # one whole share of the split sees nothing
    assert pkv.lib().pkv_flash_attn_ext_flat_workspace_bytes(n_q, nh, D, n_kv) > 0
    a = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=tk, type_v=tk, scale=D ** -0.5)
    b = pkv.flash_attn_ext_flat(dev(q), dev(k), dev(v), dev(mask), type_k=tk, type_v=tk, scale=D ** -0.5, split=False)
    torch.cuda.synchronize()
    assert np.abs(a.cpu().numpy() - b.cpu().numpy()).max() <= 2e-6
    _check(O, a.cpu().numpy(), q, k, v, mask, tk, tk, D)
