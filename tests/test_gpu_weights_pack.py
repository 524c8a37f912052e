"""-m gpu: K-quant weight rows in the plane layout (csrc/pkv_wlayout.h, `type | PKV_TYPE_PACKED`): pkv_weights_pack against the host restatement and
back (byte-exact round trip = what the plugin's tensor_get hands out), and every consumer of a packed tensor — the decode-step products (all epilogues,
in-kernel column, mixed types), the few-column kernel, the MFMA product and get_rows — bit-identical to the same call on the GGUF-order tensor."""
import numpy as np
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu
T = {"q4_K": 12, "q5_K": 13, "q6_K": 14}


def raw_rows(G, rng, t, M, K):
    raw = {12: G.random_q4_K, 13: G.random_q5_K, 14: G.random_q6_K}[t](rng, M, K)
    return np.frombuffer(raw, np.uint8).copy()


@pytest.mark.parametrize("tname,M,K", [("q4_K", 64, 4096), ("q4_K", 7, 768), ("q4_K", 32, 14336), ("q6_K", 64, 4096), ("q6_K", 9, 5632), ("q6_K", 16, 14336),
                                       ("q5_K", 64, 4096), ("q5_K", 7, 768), ("q5_K", 32, 14336)])
def test_pack_matches_host_and_round_trips(pkv, tname, M, K):
    from llama_cpp_test_amd import gguf_synth as G
    t = T[tname]
    rng = np.random.default_rng(M + K)
    w = raw_rows(G, rng, t, M, K)
    pk = pkv.weights_pack(dev(w), t, M, K)
    assert np.array_equal(pk.cpu().numpy().reshape(M, -1), pkv.pack_rows_host(w.reshape(M, -1), t, K))
    back = pkv.weights_pack(pk, t, M, K, inverse=True)
    assert np.array_equal(back.cpu().numpy(), w)


@pytest.mark.parametrize("types,Ms,K", [(("q4_K",), (4096,), 4096), (("q6_K",), (4096,), 4096), (("q4_K",), (4096,), 14336), (("q6_K",), (4096,), 14336),
                                        (("q4_K", "q4_K", "q6_K"), (4096, 1024, 1024), 4096), (("q4_K", "q6_K", "q4_K"), (2, 4, 6), 256),
                                        (("q6_K", "q4_K"), (64, 130), 768), (("q4_K",), (1030,), 768), (("q6_K",), (16032,), 4096), (("q4_K",), (512,), 28672),
                                        (("q5_K",), (4096,), 4096), (("q5_K",), (4096,), 14336), (("q5_K", "q5_K", "q5_K"), (4096, 1024, 1024), 4096), (("q5_K", "q5_K", "q6_K"), (4096, 1024, 1024), 4096),
                                        (("q5_K", "q5_K", "q6_K"), (2, 4, 6), 256), (("q6_K", "q5_K", "q4_K"), (64, 130, 32), 768), (("q5_K",), (1030,), 768), (("q5_K",), (16032,), 4096), (("q5_K",), (512,), 28672)])
def test_gemv_packed_equals_gguf_order(pkv, types, Ms, K):
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(sum(Ms) + K)
    ts = [T[n] for n in types]
    ws = [dev(raw_rows(G, rng, t, M, K)) for t, M in zip(ts, Ms)]
    wp = [pkv.weights_pack(w, t, M, K) for w, t, M in zip(ws, ts, Ms)]
    tp = [t | pkv.PACKED for t in ts]
    x = dev(rng.standard_normal(K).astype(np.float32))
    img = pkv.quantize_act(x)
    a = [torch.zeros(M, device="cuda") for M in Ms]
    b = [torch.zeros(M, device="cuda") for M in Ms]
    pkv.gemv_q(ws, ts, Ms, K, img, a)
    pkv.gemv_q(wp, tp, Ms, K, img, b)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    if K <= 8192:                                                   # the in-kernel norm + quantize prologue and the residual epilogue
        nw = dev((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
        r = dev(rng.standard_normal(Ms[0]).astype(np.float32))
        ya, yb = r.clone(), r.clone()
        pkv.gemv_q(ws[:1], ts[:1], Ms[:1], K, None, [ya], x=x, norm_w=nw, eps=1e-5, residual=ya)
        pkv.gemv_q(wp[:1], tp[:1], Ms[:1], K, None, [yb], x=x, norm_w=nw, eps=1e-5, residual=yb)
        assert torch.equal(ya, yb)
    with pytest.raises(pkv.PkvError):                               # one launch = one layout
        if len(ws) > 1:
            pkv.gemv_q([ws[0], wp[1]], [ts[0], tp[1]], Ms[:2], K, img, a[:2])
        else:
            raise pkv.PkvError(-1, "single")


@pytest.mark.parametrize("t", [12, 13])
def test_swiglu_packed_equals_gguf_order(pkv, t):
    from llama_cpp_test_amd import gguf_synth as G
    rng = np.random.default_rng(5)
    M, K = 14336, 4096
    wg, wu = dev(raw_rows(G, rng, t, M, K)), dev(raw_rows(G, rng, t, M, K))
    x, nw = dev(rng.standard_normal(K).astype(np.float32)), dev((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    ya, yb = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
    pkv.gemv_q([wg, wu], [t, t], [M, M], K, None, [ya], x=x, norm_w=nw, eps=1e-5, silu_mul=True)
    pkv.gemv_q([pkv.weights_pack(wg, t, M, K), pkv.weights_pack(wu, t, M, K)], [t | pkv.PACKED] * 2, [M, M], K, None, [yb], x=x, norm_w=nw, eps=1e-5, silu_mul=True)
    assert torch.equal(ya, yb)


@pytest.mark.parametrize("tname,M,K,n_cols", [("q4_K", 1024, 4096, 1), ("q4_K", 1024, 4096, 3), ("q4_K", 1000, 768, 7), ("q4_K", 1024, 4096, 64), ("q4_K", 2048, 14336, 200),
                                              ("q6_K", 1024, 4096, 1), ("q6_K", 1024, 4096, 5), ("q6_K", 1024, 4096, 96), ("q6_K", 300, 1024, 33),
                                              # the third form of the MFMA product (raw planes by LDS-DMA): 64 x 128 tiles, 128 x 128 tiles (2048 rows x 3072 columns), a long K
                                              ("q4_K", 2048, 4096, 3072), ("q6_K", 2048, 4096, 3072), ("q6_K", 1024, 14336, 512), ("q6_K", 1000, 2048, 130),
                                              # block_q5_K: decode-step product, MFMA first form (3 / 33 columns), second form (96 / 200 / 3072 columns)
                                              ("q5_K", 1024, 4096, 1), ("q5_K", 1024, 4096, 3), ("q5_K", 300, 1024, 33), ("q5_K", 1024, 4096, 96), ("q5_K", 2048, 14336, 200), ("q5_K", 2048, 4096, 3072)])
def test_mul_mat_packed_equals_gguf_order(pkv, tname, M, K, n_cols):
    """pkv_mul_mat on a packed tensor: one column (decode product), a few (q4_K: the LDS-staged kernel), many (MFMA)"""
    from llama_cpp_test_amd import gguf_synth as G
    t = T[tname]
    rng = np.random.default_rng(M + K + n_cols)
    w = dev(raw_rows(G, rng, t, M, K))
    x = dev(rng.standard_normal((n_cols, K)).astype(np.float32))
    a = pkv.mul_mat(w, t, M, K, x)
    b = pkv.mul_mat(pkv.weights_pack(w, t, M, K), t | pkv.PACKED, M, K, x)
    assert torch.equal(a, b)
    if t == 12:
        a2 = pkv.mul_mat_multi([w, w], t, [M, M], K, x)
        wp = pkv.weights_pack(w, t, M, K)
        b2 = pkv.mul_mat_multi([wp, wp], t | pkv.PACKED, [M, M], K, x)
        assert torch.equal(a2[0], b2[0]) and torch.equal(a2[1], b2[1]) and torch.equal(a2[0], a)


@pytest.mark.parametrize("tname", ["q4_K", "q5_K", "q6_K"])
def test_get_rows_packed(pkv, tname):
    from llama_cpp_test_amd import gguf_synth as G
    t = T[tname]
    rng = np.random.default_rng(t)
    n_rows, K = 300, 1024
    w = dev(raw_rows(G, rng, t, n_rows, K))
    ids = dev(np.array([5, 0, 299, 5, 17, 123], np.int32))
    assert torch.equal(pkv.get_rows(w, t, ids, K, n_rows), pkv.get_rows(pkv.weights_pack(w, t, n_rows, K), t | pkv.PACKED, ids, K, n_rows))
