"""-m gpu: first slice of the rest of the decoder layer (pkv_rms_norm_f32, pkv_add_f32 / pkv_mul_f32, pkv_silu_f32, pkv_mul_mat)
through the C ABI against exact (f64) arithmetic on the same stored weights.  The reference's own criterion — NMSE vs its CPU
backend in test-backend-ops (1e-7 / 5e-4 for MUL_MAT) — is checked through the plugin in tests/test_gpu_reference_harness.py; here
the bar is exactness: the kernels keep activations in f32 (the reference's CPU path rounds them to f16 / q8_0 for f16 / q8_0
weights), so they must sit within f32 accumulation error of the f64 result."""
import numpy as np
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu


def test_rms_norm_add_mul_silu(pkv):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 2048)).astype(np.float32) * 3
    w = rng.standard_normal(2048).astype(np.float32)
    r = rng.standard_normal((37, 2048)).astype(np.float32)
    y = pkv.rms_norm(dev(x), 1e-5).cpu().numpy()
    ref = x / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + 1e-5)
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()
    assert np.array_equal(pkv.mul(dev(x), dev(w)).cpu().numpy(), x * w)          # row broadcast: the norm weight
    assert np.array_equal(pkv.add(dev(x), dev(r)).cpu().numpy(), x + r)          # same shape: the residual
    x3 = x.reshape(37, 2, 1024)
    b3 = r[0].reshape(1, 2, 1024)                                                # a block of rows that repeats
    assert np.array_equal(pkv.add(dev(x3), dev(b3)).cpu().numpy(), x3 + b3)
    s = pkv.silu(dev(x)).cpu().numpy()
    sref = x.astype(np.float64) / (1 + np.exp(-x.astype(np.float64)))
    assert np.abs(s - sref).max() <= 2e-6 * np.abs(sref).max()
    # the fused pairs are the two separate launches, bit for bit
    assert torch.equal(pkv.rms_norm_mul(dev(x), dev(w), 1e-5), pkv.mul(pkv.rms_norm(dev(x), 1e-5), dev(w)))
    assert torch.equal(pkv.silu_mul(dev(x), dev(r)), pkv.mul(pkv.silu(dev(x)), dev(r)))
    # ... and so is the residual add in front of the norm (register path: rows of 2048; loop path: rows of 1030 and of 9216 elements)
    for ne0, rows in ((2048, 37), (1030, 5), (9216, 3), (4096, 1)):
        a = dev(rng.standard_normal((rows, ne0)).astype(np.float32) * 2)
        b = dev(rng.standard_normal((rows, ne0)).astype(np.float32))
        w2 = dev(rng.standard_normal(ne0).astype(np.float32))
        want_sum = pkv.add(a, b)
        want = pkv.mul(pkv.rms_norm(want_sum, 1e-5), w2)
        assert torch.equal(pkv.rms_norm_mul(want_sum, w2, 1e-5), want)
        got_sum, got = pkv.add_rms_norm_mul(a, b, w2, 1e-5)
        assert torch.equal(got_sum, want_sum) and torch.equal(got, want)
        a2 = a.clone()
        got_sum, got = pkv.add_rms_norm_mul(a2, b, w2, 1e-5, in_place=True)
        assert got_sum.data_ptr() == a2.data_ptr() and torch.equal(a2, want_sum) and torch.equal(got, want)


@pytest.mark.parametrize("tname,M,K,N", [("f32", 64, 256, 1), ("f16", 300, 2048, 1), ("q8_0", 2048, 2048, 1), ("q8_0", 96, 5632, 3),
                                         ("f16", 32, 512, 19), ("q8_0", 257, 96, 9), ("q4_K", 4096, 4096, 1), ("q4_K", 130, 768, 11),
                                         # several row groups per workgroup (grid capped), 3.5 chunks of one staged column, a column
                                         # longer than one staging (super-chunks), and all of it with two columns
                                         ("q4_K", 14336, 4096, 1), ("q4_K", 1030, 14336, 2), ("q4_K", 70, 20480, 2), ("q4_K", 8200, 512, 2)])
def test_mul_mat(pkv, O, tname, M, K, N):
    t = {"f32": 0, "f16": 1, "q8_0": 8, "q4_K": 12}[tname]
    rng = np.random.default_rng(M + K + N)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((N, K)).astype(np.float32)
    if t == 12:      # random well-formed block_q4_K rows, checked against dequantize_row_q4_K (== the reference's, tests/test_gguf.py)
        from llama_cpp_test_amd import gguf_synth as G
        raw = G.random_q4_K(rng, M, K)
        wb, wd = np.frombuffer(raw, np.uint8).copy(), G.dequantize_q4_K(raw, K).astype(np.float64)
    elif t == 0:
        wb, wd = w.view(np.uint8), w.astype(np.float64)
    else:
        wb = O.from_float(t, w, variant="ref") if t == 8 else O.from_float(t, w)
        wd = O.to_float(t, wb, K).astype(np.float64)
    y = pkv.mul_mat(dev(np.ascontiguousarray(wb).reshape(-1)), t, M, K, dev(x)).cpu().numpy()
    ref = x.astype(np.float64) @ wd.T
    assert y.shape == (N, M)
    if N >= 16 and t in (1, 8, 12):
        # many columns: the MFMA product (csrc/pkv_gemm.hip) — weights and activations as f16, f32 accumulation: far inside the harness bar (NMSE 5e-4)
        assert float(((y - ref) ** 2).sum() / (ref ** 2).sum()) <= 2e-6
        return
    if t == 12 and N == 1 and M % 2 == 0 and K % 256 == 0:
        # ONE column of K-quant rows: the decode-step product (csrc/pkv_gemv.hip) — the column as the reference's q8_K integers (per 256), integer dots
        from test_gpu_gemv import quant_act_ref
        ref_q = wd @ quant_act_ref(x[0])[2]
        assert np.abs(y[0] - ref_q).max() <= 3e-6 * max(1.0, np.abs(ref_q).max()) * np.sqrt(K / 256)
        assert float(((y - ref) ** 2).sum() / (ref ** 2).sum()) <= 5e-4
        return
    if t == 12:
        # a few columns of q4_K rows: the activations quantized to int8 per 32-element sub-block (d = amax / 127, round-to-nearest-even), integer
        # dots — the reference's CPU backend does the same with coarser q8_K blocks of 256.  Tight against that
        # arithmetic in f64, and within the harness bar (NMSE 5e-4, we ask 1e-4) of the unquantized product.
        xb = x.reshape(N, K // 32, 32)
        amax = np.abs(xb).max(-1, keepdims=True)
        dq = (amax / np.float32(127.0)).astype(np.float32)
        idq = np.where(amax != 0, np.float32(127.0) / np.where(amax != 0, amax, 1), 0).astype(np.float32)
        xh = (dq.astype(np.float64) * np.rint((xb * idq).astype(np.float32))).reshape(N, K)
        ref_q = xh @ wd.T
        assert np.abs(y - ref_q).max() <= 3e-6 * max(1.0, np.abs(ref_q).max()) * np.sqrt(K / 256)
        assert float(((y - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-4
        return
    assert np.abs(y - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max()) * np.sqrt(K / 256)


@pytest.mark.parametrize("tname,Ms,K,N", [("q4_K", (4096, 1024, 1024), 4096, 1), ("q4_K", (14336, 14336), 4096, 1), ("q4_K", (70, 9, 257, 8), 768, 3),
                                          ("q8_0", (64, 96), 256, 2), ("q8_0", (2048, 256, 256), 2048, 1), ("q8_0", (5632, 5632), 2048, 3),
                                          # prompt ubatches (>= 16 columns): ONE MFMA launch over the row tiles of all matrices, every tile shape
                                          ("q8_0", (2048, 256, 256), 2048, 512), ("q4_K", (4096, 1024, 1024), 4096, 130), ("q4_K", (14336, 14336), 4096, 512),
                                          ("q4_K", (70, 9, 257, 8), 768, 33)])
def test_mul_mat_multi(pkv, tname, Ms, K, N):
    """pkv_mul_mat_multi == the separate pkv_mul_mat calls, bit for bit (same arithmetic, one launch for q4_K)"""
    t = {"q8_0": 8, "q4_K": 12}[tname]
    rng = np.random.default_rng(sum(Ms) + K)
    from llama_cpp_test_amd import gguf_synth as G
    x = dev(rng.standard_normal((N, K)).astype(np.float32))
    if t == 12:
        ws = [dev(np.frombuffer(G.random_q4_K(rng, M, K), np.uint8).copy()) for M in Ms]
    else:
        ws = [dev(np.frombuffer(G.quantize_q8_0((rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)), np.uint8).copy()) for M in Ms]
    got = pkv.mul_mat_multi(ws, t, Ms, K, x)
    for w, M, y in zip(ws, Ms, got):
        want = pkv.mul_mat(w, t, M, K, x)
        assert y.shape == (N, M) and torch.equal(y, want)


def test_soft_max_ext(pkv):
    """ggml_soft_max_ext (/root/reference/ggml/src/ggml-cpu/ggml-cpu.c:10222-10318): scale, f32 / f16 mask broadcast over heads, ALiBi slopes, -inf mask cells,
    rows that are not a multiple of the workgroup; against f64 arithmetic"""
    rng = np.random.default_rng(5)
    for (ne0, ne1, ne2, ne3), mtype, scale, max_bias in [((16, 2, 32, 1), None, 0.1, 0.0), ((1023, 15, 1, 1), np.float32, 1.0, 0.0), ((300, 7, 12, 2), np.float16, 0.125, 8.0),
                                                        ((4097, 3, 8, 1), np.float16, 0.0884, 0.0), ((32, 2, 32, 1), np.float32, 0.1, 8.0)]:
        x = rng.standard_normal((ne3, ne2, ne1, ne0)).astype(np.float32) * 4
        m = None
        if mtype is not None:
            m = rng.standard_normal((ne1 + 3, ne0)).astype(mtype)                 # padded rows, like GGML_KQ_MASK_PAD
            m[:, ne0 // 2:][rng.random((ne1 + 3, ne0 - ne0 // 2)) < 0.3] = -np.inf   # causal-style holes (never a whole row: column 0 stays finite)
        n_head_log2 = 1 << int(np.floor(np.log2(ne2)))
        m0, m1 = 2.0 ** (-max_bias / n_head_log2), 2.0 ** (-(max_bias / 2) / n_head_log2)
        slope = np.array([(m0 ** (h + 1) if h < n_head_log2 else m1 ** (2 * (h - n_head_log2) + 1)) if max_bias > 0 else 1.0 for h in range(ne2)])
        w = x.astype(np.float64) * np.float32(scale)
        if m is not None:
            w = w + slope[None, :, None, None] * m[:ne1].astype(np.float64)[None, None]
        e = np.exp(w - w.max(-1, keepdims=True))
        ref = e / e.sum(-1, keepdims=True)
        got = pkv.soft_max_ext(dev(x), None if m is None else dev(m), scale, max_bias).cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-6 and abs(got.sum(-1) - 1).max() <= 1e-5
        xin = dev(x)
        assert torch.equal(pkv.soft_max_ext(xin, None if m is None else dev(m), scale, max_bias, out=xin), dev(got))   # in place


def test_strided_cpy_and_mul_mat(pkv):
    """the generic ops of a -fa 0 graph (/root/reference/src/llama-graph.cpp:1201-1239, :1387-1394): strided copies (the store into the transposed V
    cache, kqv_merged -> cont) and batched / broadcast products over f16 views of the cache"""
    rng = np.random.default_rng(11)
    # V store of a -fa 0 step: transpose(v_cur [n_tok][n_embd]) -> view [n_embd][n_tok] of the [n_embd][n_ctx] f16 cache at column kv_head
    n_tok, n_embd, n_ctx, head = 5, 256, 64, 17
    v_cur = dev(rng.standard_normal((n_tok, n_embd)).astype(np.float32))
    cache = torch.zeros((n_embd, n_ctx), dtype=torch.float16, device="cuda")
    pkv.cpy_strided(v_cur.t(), cache[:, head:head + n_tok])
    want = torch.zeros_like(cache)
    want[:, head:head + n_tok] = v_cur.t().to(torch.float16)
    assert torch.equal(cache, want)
    # cont of a permuted f32 tensor, and f16 -> f32 of a strided slice
    a = dev(rng.standard_normal((3, 4, 6, 10)).astype(np.float32))
    out = torch.empty((3, 6, 4, 10), device="cuda")
    assert torch.equal(pkv.cpy_strided(a.permute(0, 2, 1, 3), out), a.permute(0, 2, 1, 3).contiguous())
    h = a.to(torch.float16)
    o2 = torch.empty((3, 4, 6, 5), device="cuda")
    assert torch.equal(pkv.cpy_strided(h[..., ::2], o2), h[..., ::2].float())
    # kq = k . q: k = f16 cache view [n_head_kv][n_kv][D] with cell-major rows, q = permuted f32 [n_head][n_tok][D], gqa 4
    D, n_kv, n_head, n_head_kv, n_tok = 64, 96, 8, 2, 7
    kc = dev((rng.standard_normal((n_kv, n_head_kv, D)) / 8).astype(np.float16))           # cache rows: cell x (head, d)
    qc = dev(rng.standard_normal((n_tok, n_head, D)).astype(np.float32))                   # Qcur: token x (head, d)
    kq = pkv.mul_mat_strided(kc.permute(1, 0, 2), qc.permute(1, 0, 2))[0]                  # [n_head][n_tok][n_kv]
    ref = np.einsum("hkd,htd->htk", kc.cpu().numpy().astype(np.float64).transpose(1, 0, 2)[np.arange(n_head) // (n_head // n_head_kv)],
                    qc.cpu().numpy().astype(np.float64).transpose(1, 0, 2))
    assert np.abs(kq.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    # kqv = v^T . p: v^T = transposed f16 cache [n_head_kv][D][n_ctx] restricted to n_kv columns, p f32 [n_head][n_tok][n_kv]
    vt = dev((rng.standard_normal((n_head_kv, D, 128)) / 8).astype(np.float16))
    p = torch.softmax(kq, -1)
    kqv = pkv.mul_mat_strided(vt[:, :, :n_kv], p)[0]                                       # [n_head][n_tok][D]
    ref2 = np.einsum("hdk,htk->htd", vt.cpu().numpy().astype(np.float64)[:, :, :n_kv][np.arange(n_head) // (n_head // n_head_kv)], p.cpu().numpy().astype(np.float64))
    assert np.abs(kqv.cpu().numpy() - ref2).max() <= 1e-5


@pytest.mark.parametrize("tname", ["f16", "q8_0", "q4_K", "q5_K", "q6_K"])
def test_mul_mat_batched_mfma(pkv, O, tname):
    """pkv_mul_mat with many activation columns (prompt processing) = the MFMA product of csrc/pkv_gemm.hip: weight tiles dequantized to f16 once per 128
    columns, activations rounded to f16, f32 accumulation.  Against f64 arithmetic on the same stored weights: NMSE far inside the harness bar (5e-4),
    every element within the f16 rounding of the activations; ragged M / N (tiles past the end), and the single-column / few-column paths agree with it."""
    from llama_cpp_test_amd import gguf_synth as G
    t = {"f16": 1, "q8_0": 8, "q4_K": 12, "q5_K": 13, "q6_K": 14}[tname]
    rng = np.random.default_rng(100 + t)
    for M, K, N in [(256, 512, 16), (384, 4096, 200), (130, 1024, 129), (1024, 2048, 512)]:
        if t in (12, 13, 14):
            raw = {12: G.random_q4_K, 13: G.random_q5_K, 14: G.random_q6_K}[t](rng, M, K)
            wb, wd = np.frombuffer(raw, np.uint8).copy(), {12: G.dequantize_q4_K, 13: G.dequantize_q5_K, 14: G.dequantize_q6_K}[t](raw, K).astype(np.float64).reshape(M, K)
        else:
            w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
            wb = O.from_float(t, w, variant="ref") if t == 8 else O.from_float(t, w)
            wd = np.asarray(O.to_float(t, wb, K), np.float64).reshape(M, K)
        x = rng.standard_normal((N, K)).astype(np.float32)
        y = pkv.mul_mat(dev(np.ascontiguousarray(wb).reshape(-1)), t, M, K, dev(x)).cpu().numpy()
        ref = x.astype(np.float64) @ wd.T
        assert y.shape == (N, M) and np.isfinite(y).all()
        err = float(((y - ref) ** 2).sum() / (ref ** 2).sum())
        assert err <= 2e-6, (M, K, N, err)
        # a handful of columns through the per-column kernels (or, q6_K, the same MFMA kernel): same product
        if M % 2 == 0:
            y1 = pkv.mul_mat(dev(np.ascontiguousarray(wb).reshape(-1)), t, M, K, dev(x[:1].copy())).cpu().numpy()
            assert float(((y1 - ref[:1]) ** 2).sum() / (ref[:1] ** 2).sum()) <= 5e-4


def test_q5_K_get_rows_and_single_column(pkv):
    """block_q5_K weights (r03): pkv_get_rows = dequantize_row_q5_K exactly; ONE activation column takes the decode-step product (csrc/pkv_gemv.hip MODE 13: the
    column as the reference's q8_K integers, integer sub-block dots — ggml_vec_dot_q5_K_q8_K's arithmetic): tight against f64 arithmetic on that rounded column,
    the harness' MUL_MAT bar (NMSE 5e-4) against the unrounded one; a few columns (2 .. 15) go through the MFMA product: within the f16 rounding of the columns"""
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(13)
    n_rows, K = 300, 1024
    raw = G.random_q5_K(rng, n_rows, K)
    w = dev(np.frombuffer(raw, np.uint8).copy())
    ids = np.array([5, 0, 299, 5, 17, 123], np.int32)
    got = pkv.get_rows(w, 13, dev(ids), K, n_rows).cpu().numpy()
    assert np.array_equal(got, G.dequantize_q5_K(raw, K)[ids])
    wd = G.dequantize_q5_K(raw, K).astype(np.float64)
    x = rng.standard_normal((3, K)).astype(np.float32)
    y = pkv.mul_mat(w, 13, n_rows, K, dev(x[:1].copy())).cpu().numpy()
    xb = x[0].reshape(-1, 256)                                  # quantize_row_q8_K: iscale = -127 / (signed value of largest magnitude), rint
    mx = xb[np.arange(len(xb)), np.abs(xb).argmax(-1)]
    isc = (np.float32(-127.0) / mx).astype(np.float32)
    xq = ((np.float32(1.0) / isc).astype(np.float64)[:, None] * np.rint((isc[:, None] * xb).astype(np.float32))).reshape(K)
    assert np.abs(y[0] - wd @ xq).max() <= 3e-6 * max(1.0, np.abs(wd @ xq).max()) * np.sqrt(K / 256)
    ref = x.astype(np.float64) @ wd.T
    assert float(((y - ref[:1]) ** 2).sum() / (ref[:1] ** 2).sum()) <= 5e-4
    y3 = pkv.mul_mat(w, 13, n_rows, K, dev(x)).cpu().numpy()
    assert float(((y3 - ref) ** 2).sum() / (ref ** 2).sum()) <= 2e-6
