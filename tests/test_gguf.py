"""Synthetic GGUF writer + minimal loader (llama.cpp_test_amd/gguf_synth.py) against the reference's own
GGUF parser / writer / model loader (oracle/_ref/libref_kv.so, driver oracle/ref_model_driver.cpp), and the real-model
attention fixture (tests/golden/model_attn.npz, produced by the reference's llama_decode) against the CPU oracle."""
import os

import numpy as np
import pytest

from oracle import ref_model as R


@pytest.fixture(scope="module")
def G(pkv):
    from llama_cpp_test_amd import gguf_synth
    return gguf_synth


TINY = dict(n_layer=2, n_embd=128, n_head=2, n_head_kv=1, n_ff=256, n_vocab=64, n_ctx=1024, rope_freq_base=10000.0, seed=4321)


def test_write_read_round_trip(G, tmp_path):
    p = str(tmp_path / "t.gguf")
    kv = [("general.architecture", G.T_STR, "llama"), ("a.u8", G.T_U8, 7), ("a.i64", G.T_I64, -5), ("a.f64", G.T_F64, 0.25),
          ("a.bool", G.T_BOOL, True), ("a.arr", G.T_ARR, (G.T_F32, [1.0, 2.5])), ("a.strs", G.T_ARR, (G.T_STR, ["x", "", "yz"]))]
    w = np.arange(64 * 3, dtype=np.float32).reshape(3, 64) / 7
    tensors = [("w.f32", (64, 3), G.GGML_F32, G.encode(w, G.GGML_F32)), ("w.q8", (64, 3), G.GGML_Q8_0, G.encode(w, G.GGML_Q8_0)),
               ("v.f16", (5,), G.GGML_F16, G.encode(np.ones(5), G.GGML_F16))]
    for align in (32, 64):
        G.write_gguf(p, kv, tensors, alignment=align)
        rkv, rt, ra = G.read_gguf(p)
        assert ra == align and rkv["a.i64"] == (G.T_I64, -5) and rkv["a.strs"][1] == (G.T_STR, ["x", "", "yz"])
        assert list(rt) == ["w.f32", "w.q8", "v.f16"]
        for (name, ne, t, data) in tensors:
            assert rt[name][0] == ne and rt[name][1] == t and bytes(rt[name][3]) == data and rt[name][2] % align == 0
    with pytest.raises(ValueError):
        open(p, "wb").write(b"GGML" + b"\0" * 32)
        G.read_gguf(p)


def test_q8_0_codec_matches_the_oracle_ref_quantizer(G, O):
    x = np.random.default_rng(0).uniform(-3, 3, (4, 256)).astype(np.float32)
    x[0, :32] = 0.0                                              # an all-zero block (d = 0)
    ours = np.frombuffer(G.quantize_q8_0(x), np.uint8)
    # quantize_row_q8_0_ref (the weight quantizer; pinned against the compiled reference in tests/test_oracle_vs_ref.py)
    assert np.array_equal(ours, O.from_float(8, x, variant="ref").reshape(-1))
    back = O.to_float(8, ours.reshape(4, -1), 256)
    assert np.abs(back - x).max() <= np.abs(x).max() / 127 * 0.55


def test_q4_K_blocks_dequantize_like_the_reference(G, O):
    """synthetic Q4_K weights are random well-formed block_q4_K rows; our dequantize_row_q4_K restatement (the checker of the
    device product) must equal the reference's own, bit for bit"""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    O.ref_rope(np.zeros((1, 1, 64), np.float32), np.zeros(1, np.int32), n_dims=64)      # any op: ggml_init fills the f16 table
    raw = G.random_q4_K(np.random.default_rng(3), 5, 1024)
    mine = G.dequantize_q4_K(raw, 1024)
    ref = O.ref_to_float(12, np.frombuffer(raw, np.uint8).reshape(5, -1), 1024)
    assert np.array_equal(mine, ref) and 0.5 / 32 < mine.std() < 2.0 / 32


def test_q5_K_blocks_dequantize_like_the_reference(G, O):
    """block_q5_K (Q5_K_M files; r03): dequantize_row_q5_K bit for bit"""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    O.ref_rope(np.zeros((1, 1, 64), np.float32), np.zeros(1, np.int32), n_dims=64)
    raw = G.random_q5_K(np.random.default_rng(5), 5, 1024)
    mine = G.dequantize_q5_K(raw, 1024)
    ref = O.ref_to_float(13, np.frombuffer(raw, np.uint8).reshape(5, -1), 1024)
    assert np.array_equal(mine, ref) and 0.5 / 32 < mine.std() < 2.0 / 32


def test_q6_K_blocks_dequantize_like_the_reference(G, O):
    """the same for block_q6_K (what a Q4_K_M file keeps attn_v / ffn_down / output in): dequantize_row_q6_K bit for bit"""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    O.ref_rope(np.zeros((1, 1, 64), np.float32), np.zeros(1, np.int32), n_dims=64)
    raw = G.random_q6_K(np.random.default_rng(4), 5, 1024)
    mine = G.dequantize_q6_K(raw, 1024)
    ref = O.ref_to_float(14, np.frombuffer(raw, np.uint8).reshape(5, -1), 1024)
    assert np.array_equal(mine, ref) and 0.5 / 32 < mine.std() < 2.0 / 32


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_kv.so not built (needs /root/reference)")
@pytest.mark.parametrize("wtype", ["f16", "q8_0", "f32", "q4_K"])
def test_reference_parser_reads_our_files_identically(G, tmp_path, wtype):
    """the reference's gguf_init_from_file sees exactly what our reader sees: every key, type, value, tensor shape, offset and
    data hash; and the reference's model loader accepts the file and decodes with it"""
    p = str(tmp_path / "m.gguf")
    cfg = dict(TINY, n_embd=256, n_head=4, n_head_kv=2, n_ff=512) if wtype == "q4_K" else TINY      # q4_K rows are multiples of 256
    G.synth_llama(p, wtype={"f16": G.GGML_F16, "q8_0": G.GGML_Q8_0, "f32": G.GGML_F32, "q4_K": G.GGML_Q4_K}[wtype], **cfg)
    assert G.describe(p) == R.gguf_describe(p)
    m = R.RefModel(p, n_ctx=256, n_batch=32)
    layers = m.decode([1, 2, 3], [0, 1, 2], [0, 0, 0])
    assert sorted(layers) == [0, 1] and np.isfinite(layers[1]["out"]).all()


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_kv.so not built (needs /root/reference)")
def test_our_reader_reads_the_reference_writers_file(G, tmp_path):
    p = str(tmp_path / "demo.gguf")
    R.gguf_write_demo(p)                                          # gguf_write_to_file: every scalar type, arrays, 3 tensors
    assert G.describe(p) == R.gguf_describe(p)
    kv, t, _ = G.read_gguf(p)
    assert kv["t.u64"] == (G.T_U64, 1 << 40) and kv["t.arr_str"][1][1] == ["alpha", "", "gamma delta"]
    assert np.array_equal(np.frombuffer(bytes(t["a.f32"][3]), np.float32), 0.5 * np.arange(21, dtype=np.float32) - 3.0)


def test_oracle_reproduces_the_real_model_attention(O):
    """tests/golden/model_attn.npz = Q/K/V projections and kqv_out captured inside the reference's llama_decode (2 sequences,
    batched prompt, interleaved decode, chunked prefill).  The oracle chain rope -> from_float -> flash_attn_ext over each
    token's own sequence must give the model's attention output BIT FOR BIT: the de-paged, per-sequence view is exactly what
    the reference's ring + KQ_mask compute (masked cells are skipped, the rest are visited in position order)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_attn.npz"))
    D = 64
    for name, t in (("f16", 1), ("q8_0", 8)):
        store = {}                                               # (layer, seq) -> [k rows], [v rows]
        for ci in range(int(g["n_calls"])):
            seq, pos = g[f"{name}__c{ci}__seq"], g[f"{name}__c{ci}__pos"]
            for il in (0, 1):
                q, k, v, out = (g[f"{name}__c{ci}__l{il}__{x}"] for x in ("q", "k", "v", "out"))
                qr = O.rope(q, pos, n_dims=D, freq_base=10000.0, n_ctx_orig=1024)
                kr = O.from_float(t, O.rope(k, pos, n_dims=D, freq_base=10000.0, n_ctx_orig=1024))
                vr = O.from_float(t, v)
                for i, s in enumerate(seq):
                    ks, vs = store.setdefault((il, int(s)), ([], []))
                    ks.append(kr[i]); vs.append(vr[i])
                    kc, vc = np.stack(ks).transpose(1, 0, 2), np.stack(vs).transpose(1, 0, 2)
                    ref = O.flash_attn_ext(qr[i][:, None, :], kc, vc, None, type_k=t, type_v=t, scale=D ** -0.5)[0]
                    assert np.array_equal(ref, out[i]), (name, ci, il, i)
