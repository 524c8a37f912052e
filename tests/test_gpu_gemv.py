"""-m gpu: the decode-step products of csrc/pkv_gemv.hip through the C ABI — pkv_quantize_act, pkv_rms_norm_mul_q, pkv_gemv_q (q4_K / q6_K rows,
q5_K rows too; up to three matrices per launch, residual and SiLU(gate) * up epilogues) and pkv_get_rows — against exact (f64) arithmetic on the same
stored weights.  The weight rows are random well-formed block_q4_K / block_q6_K whose numpy dequantizers are bit-equal to the reference's
(tests/test_gguf.py).  Bars: tight (f32 accumulation) against the arithmetic the kernel states — the activation column as the reference's
q8_K integers (quantize_row_q8_K), integer dots per sub-block — and the reference harness' MUL_MAT bar (NMSE 5e-4, tests/test-backend-ops.cpp)
against the unquantized product."""
import numpy as np
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu


def quant_act_ref(x):
    """quantize_row_q8_K_ref in numpy: per 256 elements iscale = -127 / max (signed value of largest magnitude), q = rint(iscale * x), d = 1 / iscale"""
    K = x.shape[-1]
    xb = x.reshape(K // 256, 256)
    idx = np.abs(xb).argmax(-1)
    mx = xb[np.arange(len(xb)), idx]
    nz = mx != 0
    iscale = np.where(nz, np.float32(-127.0) / np.where(nz, mx, 1).astype(np.float32), 0).astype(np.float32)
    q = np.rint((iscale[:, None] * xb).astype(np.float32))
    d = np.where(nz, np.float32(1.0) / np.where(nz, iscale, 1).astype(np.float32), 0).astype(np.float32)
    return q.astype(np.int8), d, (d.astype(np.float64)[:, None] * q).reshape(K)


def weights(G, rng, t, M, K):
    raw = {12: G.random_q4_K, 13: G.random_q5_K, 14: G.random_q6_K}[t](rng, M, K)
    wd = {12: G.dequantize_q4_K, 13: G.dequantize_q5_K, 14: G.dequantize_q6_K}[t](raw, K).astype(np.float64)
    return dev(np.frombuffer(raw, np.uint8).copy()), wd


def check(y, ref_q, ref, K):
    assert np.abs(y - ref_q).max() <= 3e-6 * max(1.0, np.abs(ref_q).max()) * np.sqrt(K / 256)
    assert float(((y - ref) ** 2).sum() / (ref ** 2).sum()) <= 5e-4


def test_quantize_act_image(pkv, O):
    """the activation image holds the reference's q8_K integers and scales (quantize_row_q8_K_ref of the compiled reference, bit for bit)"""
    rng = np.random.default_rng(0)
    for K in (256, 4096, 14336):
        x = (rng.standard_normal(K) * rng.choice([1e-3, 1.0, 50.0], K)).astype(np.float32)
        if K > 256:
            x[256:512] = 0.0                                       # an all-zero block: d = 0, q = 0
        img = pkv.quantize_act(dev(x)).cpu().numpy()
        q, d, _ = quant_act_ref(x)
        got_q, got_d, got_bs = img[:K].view(np.int8), img[K:K + K // 64].view(np.float32), img[K + K // 64:].view(np.int32)
        # the device always takes max = +amax (the reference: the signed value of largest magnitude): d and the block's q may come out negated
        # TOGETHER; every quantity a product uses — d * q, d * bsum — is the same bits
        sgn = np.where(np.sign(got_d) == np.sign(d), 1, -1).astype(np.int32)
        assert np.array_equal(np.abs(got_d), np.abs(d))
        assert np.array_equal(got_q.reshape(-1, 256).astype(np.int32) * sgn[:, None], q.reshape(-1, 256).astype(np.int32))
        assert np.array_equal(got_bs.reshape(-1, 8) * sgn[:, None], q.astype(np.int32).reshape(K // 256, 8, 32).sum(-1))
        if O.ref_available():                                      # block_q8_K = {f32 d, i8 qs[256], i16 bsums[16]}
            O.ref_rope(np.zeros((1, 1, 64), np.float32), np.zeros(1, np.int32), n_dims=64)
            import ctypes, os
            L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libggml_ref.so"))
            rb = np.zeros((K // 256, 292), np.uint8)                 # (the reference leaves the bsums of an all-zero block unwritten)
            L.quantize_row_q8_K_ref(x.ctypes.data_as(ctypes.c_void_p), rb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(K))
            rd, rq = rb[:, :4].copy().view(np.float32)[:, 0], rb[:, 4:260].copy().view(np.int8).astype(np.int32)
            assert np.array_equal(rd.astype(np.float64)[:, None] * rq, got_d.astype(np.float64)[:, None] * got_q.reshape(-1, 256))
            rbs = rb[:, 260:].copy().view(np.int16).astype(np.int32).reshape(K // 256, 8, 2).sum(-1)
            assert np.array_equal(rd.astype(np.float64)[:, None] * rbs, got_d.astype(np.float64)[:, None] * got_bs.reshape(-1, 8))


@pytest.mark.parametrize("tname,M,K", [("q4_K", 4096, 4096), ("q4_K", 1024, 4096), ("q4_K", 4096, 14336), ("q4_K", 14336, 4096), ("q4_K", 6, 256),
                                       ("q4_K", 1030, 768), ("q6_K", 4096, 4096), ("q6_K", 1024, 4096), ("q6_K", 4096, 14336), ("q6_K", 128256, 4096),
                                       ("q6_K", 10, 512), ("q4_K", 512, 28672),
                                       # block_q5_K rows (Q5_K_M files): every launch shape of the kernel — 16 / 8 / 4 waves, two waves per row pair (ffn_down), no-loop and looped
                                       ("q5_K", 4096, 4096), ("q5_K", 1024, 4096), ("q5_K", 4096, 14336), ("q5_K", 14336, 4096), ("q5_K", 6, 256), ("q5_K", 1030, 768),
                                       ("q5_K", 128256, 4096), ("q5_K", 512, 28672)])
def test_gemv_single(pkv, tname, M, K):
    from llama_cpp_test_amd import gguf_synth as G
    t = {"q4_K": 12, "q5_K": 13, "q6_K": 14}[tname]
    rng = np.random.default_rng(M + K + t)
    if M > 20000:                                                  # the lm_head shape: a few thousand distinct rows repeated keep the host side quick
        wb0, wd0 = weights(G, rng, t, 2004, K)
        rep = M // 2004
        wb, wd = wb0.repeat(rep), np.tile(wd0, (rep, 1))
    else:
        wb, wd = weights(G, rng, t, M, K)
    x = rng.standard_normal(K).astype(np.float32)
    img = pkv.quantize_act(dev(x))
    y = torch.zeros(M, device="cuda")
    pkv.gemv_q([wb], [t], [M], K, img, [y])
    _, _, xh = quant_act_ref(x)
    check(y.cpu().numpy(), wd @ xh, wd @ x.astype(np.float64), K)
    # residual epilogue, in place on the residual
    r = rng.standard_normal(M).astype(np.float32)
    yr = dev(r.copy())
    pkv.gemv_q([wb], [t], [M], K, img, [yr], residual=yr)
    assert np.abs(yr.cpu().numpy() - (r + y.cpu().numpy())).max() <= 1e-6 * max(1.0, float(np.abs(r).max()))


@pytest.mark.parametrize("types,Ms,K", [(("q4_K", "q4_K", "q4_K"), (4096, 1024, 1024), 4096), (("q4_K", "q4_K", "q6_K"), (4096, 1024, 1024), 4096),
                                        (("q6_K", "q4_K"), (64, 130), 768), (("q4_K", "q6_K", "q4_K"), (2, 4, 6), 256),
                                        (("q5_K", "q5_K", "q6_K"), (4096, 1024, 1024), 4096), (("q5_K", "q5_K", "q5_K"), (4096, 1024, 1024), 4096),
                                        (("q6_K", "q5_K", "q4_K"), (64, 130, 32), 768), (("q5_K", "q6_K", "q5_K"), (2, 4, 6), 256), (("q4_K", "q5_K"), (4096, 4096), 4096)])   # (same launch shape for the group and for each matrix alone: a long-row product alone takes the two-waves-per-row-pair form, whose sums meet in another order)
def test_gemv_multi_equals_single(pkv, types, Ms, K):
    """products that share the activation image in ONE launch == the separate launches, bit for bit, mixed weight types included"""
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(sum(Ms) + K)
    ts = [{"q4_K": 12, "q5_K": 13, "q6_K": 14}[n] for n in types]
    ws = [weights(G, rng, t, M, K)[0] for t, M in zip(ts, Ms)]
    img = pkv.quantize_act(dev(rng.standard_normal(K).astype(np.float32)))
    outs = [torch.zeros(M, device="cuda") for M in Ms]
    pkv.gemv_q(ws, ts, Ms, K, img, outs)
    for w, t, M, y in zip(ws, ts, Ms, outs):
        one = torch.zeros(M, device="cuda")
        pkv.gemv_q([w], [t], [M], K, img, [one])
        assert torch.equal(y, one)


@pytest.mark.parametrize("tg,tu,M,K", [("q4_K", "q4_K", 14336, 4096), ("q4_K", "q6_K", 300, 1024), ("q5_K", "q5_K", 14336, 4096), ("q5_K", "q5_K", 300, 1024), ("q5_K", "q6_K", 300, 1024), ("q4_K", "q5_K", 1024, 4096)])
def test_gemv_swiglu(pkv, tg, tu, M, K):
    """gate + up + SiLU(gate) * up in one launch == silu(W_gate x) * (W_up x) of the separate products"""
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(M + K)
    t0, t1 = {"q4_K": 12, "q5_K": 13, "q6_K": 14}[tg], {"q4_K": 12, "q5_K": 13, "q6_K": 14}[tu]
    (wg, _), (wu, _) = weights(G, rng, t0, M, K), weights(G, rng, t1, M, K)
    img = pkv.quantize_act(dev(rng.standard_normal(K).astype(np.float32)))
    g, u, y = (torch.zeros(M, device="cuda") for _ in range(3))
    pkv.gemv_q([wg, wu], [t0, t1], [M, M], K, img, [g, u])
    pkv.gemv_q([wg, wu], [t0, t1], [M, M], K, img, [y], silu_mul=True)
    gd, ud = g.double().cpu().numpy(), u.double().cpu().numpy()
    want = gd / (1 + np.exp(-gd)) * ud
    assert np.abs(y.cpu().numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def test_rms_norm_mul_q(pkv):
    """rms_norm(x [+ res]) * w straight into the activation image == pkv_add_rms_norm_mul_f32 followed by pkv_quantize_act"""
    rng = np.random.default_rng(3)
    for K in (256, 2048, 4096, 8192):
        x, r, w = (dev(rng.standard_normal(K).astype(np.float32) * s) for s in (3.0, 1.0, 1.0))
        y = torch.empty(K, device="cuda")
        img = pkv.rms_norm_mul_q(x, w, 1e-5, pkv.act_image(K), y=y)
        want = pkv.rms_norm_mul(x.reshape(1, K), w, 1e-5).reshape(K)
        assert torch.equal(y, want) and torch.equal(img, pkv.quantize_act(want))
        xo = torch.empty(K, device="cuda")
        img2 = pkv.rms_norm_mul_q(x, w, 1e-5, pkv.act_image(K), res=r, x_out=xo)
        s_, n_ = pkv.add_rms_norm_mul(x.reshape(1, K), r.reshape(1, K), w, 1e-5)
        assert torch.equal(xo, s_.reshape(K)) and torch.equal(img2, pkv.quantize_act(n_.reshape(K)))


@pytest.mark.parametrize("tname", ["f32", "f16", "q8_0", "q4_K", "q6_K"])
def test_get_rows(pkv, O, tname):
    """ggml_get_rows: the token-embedding lookup; values = the reference's dequantization of the picked rows, bit for bit"""
    from llama_cpp_test_amd import gguf_synth as G
    t = {"f32": 0, "f16": 1, "q8_0": 8, "q4_K": 12, "q6_K": 14}[tname]
    rng = np.random.default_rng(t)
    n_rows, K = 300, 1024
    if t in (12, 14):
        raw = (G.random_q4_K if t == 12 else G.random_q6_K)(rng, n_rows, K)
        wb, wd = np.frombuffer(raw, np.uint8).copy(), (G.dequantize_q4_K if t == 12 else G.dequantize_q6_K)(raw, K)
    else:
        w = (rng.standard_normal((n_rows, K)) / 32).astype(np.float32)
        wb = w.view(np.uint8) if t == 0 else (O.from_float(t, w, variant="ref") if t == 8 else O.from_float(t, w))
        wd = w if t == 0 else O.to_float(t, wb, K)
    ids = np.array([5, 0, 299, 5, 17, 123], np.int32)
    out = pkv.get_rows(dev(np.ascontiguousarray(wb).reshape(-1)), t, dev(ids), K, n_rows).cpu().numpy()
    assert np.array_equal(out, np.asarray(wd).reshape(n_rows, K)[ids])


@pytest.mark.parametrize("K,norm", [(4096, True), (4096, False), (14336, False), (8192, True), (256, True)])
def test_gemv_in_kernel_column_equals_separate_launches(pkv, K, norm):
    """x (or rms_norm(x) * w) quantized in the product's own prologue == pkv_rms_norm_mul_q / pkv_quantize_act followed by the product, bit for bit"""
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(K + norm)
    M = 512
    wq, _ = weights(G, rng, 12, M, K)
    w6, _ = weights(G, rng, 14, M, K)
    x = dev(rng.standard_normal(K).astype(np.float32) * 2)
    nw = dev((1 + 0.1 * rng.standard_normal(K)).astype(np.float32)) if norm else None
    img = pkv.rms_norm_mul_q(x, nw, 1e-5, pkv.act_image(K)) if norm else pkv.quantize_act(x)
    want = [torch.zeros(M, device="cuda") for _ in range(2)]
    pkv.gemv_q([wq, w6], [12, 14], [M, M], K, img, want)
    got = [torch.zeros(M, device="cuda") for _ in range(2)]
    pkv.gemv_q([wq, w6], [12, 14], [M, M], K, None, got, x=x, norm_w=nw, eps=1e-5)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # q5_K rows beside q6_K ones (the wq / wk / wv group of a Q5_K_M file; MODE 1 of the kernel)
    w5, _ = weights(G, rng, 13, M, K)
    want5 = [torch.zeros(M, device="cuda") for _ in range(3)]
    pkv.gemv_q([w5, w6, w5], [13, 14, 13], [M, M, M], K, img, want5)
    got5 = [torch.zeros(M, device="cuda") for _ in range(3)]
    pkv.gemv_q([w5, w6, w5], [13, 14, 13], [M, M, M], K, None, got5, x=x, norm_w=nw, eps=1e-5)
    assert all(torch.equal(a, b) for a, b in zip(got5, want5)) and torch.equal(got5[1], want[1]) and torch.equal(got5[0], got5[2])
    # a residual epilogue in place on a SMALL product (more waves than row pairs: surplus waves must not add twice)
    r = dev(rng.standard_normal(6).astype(np.float32))
    w_small, _ = weights(G, rng, 12, 6, K)
    y1 = r.clone()
    pkv.gemv_q([w_small], [12], [6], K, img, [y1], residual=y1)
    y0 = torch.zeros(6, device="cuda")
    pkv.gemv_q([w_small], [12], [6], K, img, [y0])
    assert torch.equal(y1, r + y0)


def test_gemv_norm_outputs_are_materialised(pkv):
    """norm_out / col_out: the fused launch also writes rms_norm(x) and rms_norm(x) * w (one workgroup), the bits pkv_rms_norm_f32 / pkv_rms_norm_mul_f32 write —
    what the plugin hands to readers of the normalised column that sit in another graph of the scheduler"""
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(21)
    for K, M in ((512, 256), (4096, 1024), (8192, 64)):
        w, _ = weights(G, rng, 12, M, K)
        x = dev(rng.standard_normal(K).astype(np.float32) * 3)
        nw = dev((1 + 0.2 * rng.standard_normal(K)).astype(np.float32))
        y0, y1 = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
        no, co = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
        pkv.gemv_q([w], [12], [M], K, None, [y0], x=x, norm_w=nw, eps=1e-5)
        pkv.gemv_q([w], [12], [M], K, None, [y1], x=x, norm_w=nw, eps=1e-5, norm_out=no, col_out=co)
        assert torch.equal(y0, y1)
        assert torch.equal(no, pkv.rms_norm(x.reshape(1, K), 1e-5).reshape(K))
        assert torch.equal(co, pkv.rms_norm_mul(x.reshape(1, K), nw, 1e-5).reshape(K))
