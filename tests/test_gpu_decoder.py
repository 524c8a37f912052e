"""-m gpu: the repo's own host loop of a llama decode step (llama.cpp_test_amd/decoder.py: own GGUF loader -> pkv_get_rows / pkv_rms_norm_mul_q /
pkv_gemv_q / pkv_attn_decode_step through the C ABI) against the final logits of the REFERENCE's llama_decode on the same model
(tests/golden/decoder_logits.npz, generator tests/golden/make_decoder_golden.py).  The model is a synthetic llama GGUF in the tensor-type mix of
a Q4_K_M file whose K-quant tensors are real Gaussian weights quantized by the compiled reference's own quantizers.

What is compared, and why the bars are what they are:
  * every product runs the reference's arithmetic (activation column -> the same q8_K integers, the same integer sub-block dots): with ONE
    cell in the cache the logits equal the reference's to 1e-9 NMSE (measured 9e-14) — `first_logits`;
  * with more cells the reference accumulates V in f16 in cell order (ggml-cpu.c:12374-12388), the split-KV kernel in f32: attention outputs
    differ by ~1e-7 NMSE, which flips a few int8 roundings in the NEXT activation quantization; a quantized pipeline amplifies that to its
    own rounding-noise level.  That level is measured: the reference's logits sit 0.9-1.5e-3 NMSE from exact (f64, unquantized-activation)
    arithmetic on the same stored weights (`exact__logits`, printed by the generator).  So: the own loop must be (a) as close to exact
    arithmetic as the reference is, (b) within twice that distance of the reference, (c) pick the same greedy token."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def nmse(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(((a - b) ** 2).sum() / (b ** 2).sum())


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("make_decoder_golden", os.path.join(HERE, "golden", "make_decoder_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def model_path(gen, O, tmp_path_factory):
    if not O.ref_available():
        pytest.skip("oracle/_ref not shipped: the model's K-quant tensors are produced by the compiled reference's quantizers")
    p = str(tmp_path_factory.mktemp("m") / "synthetic-llama-q4km.gguf")
    sha = gen.build(p)
    g = np.load(os.path.join(HERE, "golden", "decoder_logits.npz"))
    assert sha == bytes(g["sha256"]).hex(), "the synthetic model no longer regenerates bit for bit: rerun tests/golden/make_decoder_golden.py"
    return p


@pytest.mark.parametrize("kv", ["f16", "q8_0"])
def test_own_loop_reproduces_reference_logits(pkv, gen, model_path, kv):
    from llama_cpp_test_amd.decoder import LlamaDecoder
    g = np.load(os.path.join(HERE, "golden", "decoder_logits.npz"))
    tok, want = g["tokens"], g[f"{kv}__logits"]
    t = pkv.TYPE_BY_NAME[kv]
    d = LlamaDecoder(model_path, n_ctx=512, type_k=t, type_v=t)
    first = d.eval(int(tok[0]), 0).cpu().numpy()
    assert nmse(first, g[f"{kv}__first_logits"]) <= 1e-9                              # the reference's arithmetic, product by product
    d.clear()
    got = []
    for i in range(gen.N_PROMPT + gen.N_DECODE):                 # the prompt goes through the same single-token step (KV identical by construction)
        lg = d.eval(int(tok[i]), i)
        if i >= gen.N_PROMPT - 1:
            got.append(lg.cpu().numpy().copy())
    exact = g["exact__logits"]
    ref_noise = max(nmse(b, e) for b, e in zip(want, exact))                          # the reference's own distance to exact arithmetic
    assert len(got) == len(want) == len(exact) and 2e-4 < ref_noise < 5e-3
    assert max(nmse(a, e) for a, e in zip(got, exact)) <= 1.5 * ref_noise              # (a)
    assert max(nmse(a, b) for a, b in zip(got, want)) <= 2.0 * ref_noise               # (b)
    assert all(int(np.argmax(a)) == int(np.argmax(b)) for a, b in zip(got, want))     # (c) greedy sampling would pick the same tokens
    # the captured graph replays the same step: re-decode the last token (its cell is removed first) and compare bit for bit
    last = gen.N_PROMPT + gen.N_DECODE - 1
    eager = got[-1]
    d.capture()
    d.cache.seq_rm(0, last, -1)
    again = d.eval(int(tok[last]), last).cpu().numpy()
    assert np.array_equal(again, eager)
    d.close()
    # GGUF-order weights (pack_weights=False) give the same logits bit for bit: the plane layout permutes bytes, not arithmetic
    d2 = LlamaDecoder(model_path, n_ctx=256, type_k=t, type_v=t, pack_weights=False)
    for i in range(last + 1):
        lg2 = d2.eval(int(tok[i]), i)
    assert np.array_equal(lg2.cpu().numpy(), eager)
    d2.close()
    # the opt-in image path (the merge launch hands the wo product its q8_K column, pkv_attn_args::out_image): same integers, same logits
    d3 = LlamaDecoder(model_path, n_ctx=256, type_k=t, type_v=t)
    d3.use_attn_image()
    if d3.attn_img is not None:
        for i in range(last + 1):
            lg3 = d3.eval(int(tok[i]), i)
        assert np.array_equal(lg3.cpu().numpy(), eager)
    d3.close()


def test_weight_types_and_bytes(pkv, model_path):
    """the loader sees the Q4_K_M mix (q6_K output / attn_v / ffn_down of the `use_more_bits` layers) and counts the bytes one step reads"""
    from llama_cpp_test_amd.decoder import LlamaDecoder
    d = LlamaDecoder(model_path, n_ctx=256)
    P = pkv.PACKED                                                 # K-quant rows are re-laid inside their own bytes at upload (csrc/pkv_wlayout.h)
    assert d.w["output.weight"][1] == pkv.Q6_K | P and d.w["blk.0.attn_q.weight"][1] == pkv.Q4_K | P and d.w["token_embd.weight"][1] == pkv.Q4_K | P
    assert {d.w[f"blk.{l}.attn_v.weight"][1] for l in range(d.n_layer)} <= {pkv.Q4_K | P, pkv.Q6_K | P}
    total = sum(t.numel() for n, (t, _, _) in d.w.items())
    assert 0 < d.weight_bytes() < total
    d.close()


def _q8k_round(x):
    """the activation column as the products see it: quantize_row_q8_K per 256 elements, back in f64"""
    xb = np.asarray(x, np.float32).reshape(-1, 256)
    mx = xb[np.arange(len(xb)), np.abs(xb).argmax(-1)]
    isc = np.where(mx != 0, np.float32(-127.0) / np.where(mx != 0, mx, 1).astype(np.float32), 0).astype(np.float32)
    q = np.rint((isc[:, None] * xb).astype(np.float32))
    d = np.where(mx != 0, np.float32(1.0) / np.where(mx != 0, isc, 1).astype(np.float32), 0).astype(np.float64)
    return (d[:, None] * q).reshape(-1)


@pytest.mark.parametrize("wtype,k_mix", [(13, True), (13, False)])
def test_q5_K_model_first_token_against_exact_arithmetic(pkv, tmp_path, wtype, k_mix):
    """the tensor mix of a Q5_K_M file (block_q5_K + block_q6_K; r03: q5_K rows in the decode-step products, MODES 13 / 1 of csrc/pkv_gemv.hip, the wq / wk / wv
    group of mixed row types in one launch) through the own host loop: with ONE cell in the cache attention is the identity on V, so the step is a chain
    of products this test restates in f64 on the numpy-dequantized weights (dequantizers bit-equal to the reference's, tests/test_gguf.py) with each
    activation column rounded to the reference's q8_K integers — the logits must agree to f32-accumulation noise, eagerly and as a captured graph"""
    from llama_cpp_test_amd import gguf_synth as G
    from llama_cpp_test_amd.decoder import LlamaDecoder
    cfg = dict(n_layer=3, n_embd=1024, n_head=8, n_head_kv=2, n_ff=2048, n_vocab=256, n_ctx=256, rope_freq_base=500000.0, seed=90 + k_mix)
    path = str(tmp_path / "q5.gguf")
    G.synth_llama(path, wtype=wtype, k_mix=k_mix, **cfg)
    _, tensors, _ = G.read_gguf(path)

    def W(name):
        ne, t, _, data = tensors[name]
        t, raw = int(t), bytes(data)
        if t == 0:
            return np.frombuffer(raw, np.float32).astype(np.float64)
        deq = {12: G.dequantize_q4_K, 13: G.dequantize_q5_K, 14: G.dequantize_q6_K}[t]
        return np.asarray(deq(raw, int(ne[0])), np.float64).reshape(int(ne[1]), int(ne[0]))
    types = {int(tensors[n][1]) for n in tensors if n.endswith("weight") and len(tensors[n][0]) == 2}
    assert types == ({13, 14} if k_mix else {13})
    eps, tok = 1e-5, 7
    E, H, HK, D = cfg["n_embd"], cfg["n_head"], cfg["n_head_kv"], cfg["n_embd"] // cfg["n_head"]
    rms = lambda x, w: x / np.sqrt((x * x).mean() + eps) * w     # noqa: E731
    h = W("token_embd.weight")[tok].astype(np.float32).astype(np.float64)
    for il in range(cfg["n_layer"]):
        p = f"blk.{il}."
        xn = _q8k_round(rms(h, W(p + "attn_norm.weight")))
        v = (W(p + "attn_v.weight") @ xn).astype(np.float32).astype(np.float16).astype(np.float64)        # through the f16 cache
        attn = np.repeat(v.reshape(HK, D), H // HK, axis=0).reshape(-1)                                     # one cell: softmax == 1
        h = h + W(p + "attn_output.weight") @ _q8k_round(attn)
        xf = _q8k_round(rms(h, W(p + "ffn_norm.weight")))
        g_, u_ = W(p + "ffn_gate.weight") @ xf, W(p + "ffn_up.weight") @ xf
        h = h + W(p + "ffn_down.weight") @ _q8k_round(g_ / (1 + np.exp(-g_)) * u_)
    want = W("output.weight") @ _q8k_round(rms(h, W("output_norm.weight")))
    d = LlamaDecoder(path, n_ctx=256)
    assert {t & ~pkv.PACKED for (_, t, ne) in d.w.values() if len(ne) == 2} == types and d.w["blk.0.attn_q.weight"][1] == pkv.Q5_K | pkv.PACKED
    got = d.eval(tok, 0).cpu().numpy().copy()
    # (a value that lands on a rounding boundary of the NEXT column's q8_K image may flip one integer against the f64 chain: 1e-5 leaves room for that)
    assert nmse(got, want) <= 1e-5 and int(np.argmax(got)) == int(np.argmax(want))
    d.capture()
    d.cache.seq_rm(0, 0, -1)
    assert np.array_equal(d.eval(tok, 0).cpu().numpy(), got)
    d.close()
