"""-m gpu: the UNMODIFIED reference runtime with the attention path on the MI355X.

oracle/_ref/libref_kv.so is the reference's llama library compiled in place (+ our driver).  Here it loads our ggml-backend
plugin exactly as GGML_BACKEND_PATH does (ggml_backend_load), loads a synthetic llama GGUF with n_gpu_layers = 99 — so the
KV cache lives in the plugin's buffers and the scheduler sends GGML_OP_CPY (KV store) and GGML_OP_FLASH_ATTN_EXT (and what else
the device claims) to it, while everything the device does not claim (mat-muls, norms) stays on the reference's CPU backend —
and runs llama_decode.  The same model and tokens on a CPU-only context of the same library are the expected values:
per-layer kqv_out NMSE <= 5e-4 (tests/test-backend-ops.cpp:3216), final logits NMSE <= 1e-4 with f16 KV and <= 5e-4 with
q8_0 KV (there the prefill kernel keeps Q in f16 where the reference quantizes Q to q8_0: different, smaller, rounding noise).
Skipped where the compiled reference or the plugin is absent."""
import os

import numpy as np
import pytest

from gpu_util import nmse
from oracle import ref_model as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "llama.cpp_test_amd", "libggml-mi355.so")
MODELS = {
    # head_dim 64, gqa 2, f16 weights, RoPE base 1e4 (TinyLlama-like)
    "tiny": dict(n_layer=2, n_embd=256, n_head=4, n_head_kv=2, n_ff=512, n_vocab=96, n_ctx=1024, rope_freq_base=10000.0, seed=77),
    # head_dim 128, gqa 4, Q8_0 weights, RoPE base 5e5 + rope_freqs table (Llama-3.1-like)
    "llama3": dict(n_layer=2, n_embd=1024, n_head=8, n_head_kv=2, n_ff=2048, n_vocab=128, n_ctx=2048, rope_freq_base=500000.0, seed=78,
                   rope_freqs=True, wtype=8),
    # the same geometry with Q4_K weights: the products of a layer go through the plugin's grouping rules (q / k / v in one launch with
    # K and V in a side buffer, gate / up in one launch) at prefill widths and at decode
    "llama3_q4k": dict(n_layer=2, n_embd=1024, n_head=8, n_head_kv=2, n_ff=2048, n_vocab=128, n_ctx=2048, rope_freq_base=500000.0, seed=79,
                       rope_freqs=True, wtype=12),
    # head size 256 (Gemma-like geometry in a llama graph), gqa 2, f16 weights: outside the paged kernels' geometries -> pkv_flash_attn_ext_flat
    "hs256": dict(n_layer=2, n_embd=1024, n_head=4, n_head_kv=2, n_ff=1024, n_vocab=96, n_ctx=1024, rope_freq_base=10000.0, seed=81),
    # ... and with the tensor mix of a Q5_K_M file (q5_K + q6_K; r03): q5_K WEIGHTS tensors packed at set_tensor; one column: the decode-step product (MODES 13 / 1 of csrc/pkv_gemv.hip, the wq / wk / wv group of mixed types in one launch), many: the MFMA product
    "llama3_q5km": dict(n_layer=2, n_embd=1024, n_head=8, n_head_kv=2, n_ff=2048, n_vocab=128, n_ctx=2048, rope_freq_base=500000.0, seed=80,
                        rope_freqs=True, wtype=13, k_mix=True),
}


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
@pytest.mark.parametrize("tk", [1, 8])
@pytest.mark.parametrize("model", ["tiny", "llama3", "llama3_q4k", "llama3_q5km"])
def test_llama_decode_with_attention_on_the_plugin(pkv, tmp_path, tk, model):
    from llama_cpp_test_amd import gguf_synth as G
    MODEL = MODELS[model]
    path = str(tmp_path / "m.gguf")
    G.synth_llama(path, **MODEL)
    rng = np.random.default_rng(5)
    # two sequences: both prompts in ONE batch, interleaved decode steps (cells of the two sequences alternate in the ring,
    # the KQ_mask separates them), then — f16 K only, the quantized K-shift graph needs a dequantizing CPY the device does not
    # claim — a context shift of sequence 0 (seq_rm + seq_add: the K-shift RoPE runs in place on the cache, in our buffers)
    # (sequence 0's prompt comes in three ubatches and ends beyond two 128-cell blocks: the plugin's paged mirrors of the cache
    # tensors are extended incrementally, ubatch by ubatch and token by token)
    calls = [([0] * 100, list(range(100))), ([0] * 100, list(range(100, 200))),
             ([0] * 70 + [1] * 33, list(range(200, 270)) + list(range(33)))] + [([0, 1], [270 + i, 33 + i]) for i in range(3)]
    if tk == 1:
        calls += ["shift", ([0, 1], [263, 36]), ([1, 0], [37, 264])]
    toks = [None if c == "shift" else rng.integers(0, MODEL["n_vocab"], len(c[0])) for c in calls]

    def run(gpu_layers):
        R.set_gpu_layers(gpu_layers)
        m = R.RefModel(path, n_ctx=512, n_batch=128, n_seq_max=2, flash_attn=True, type_k=tk, type_v=tk, n_threads=8)
        outs = []
        for c, t in zip(calls, toks):
            if c == "shift":
                m.seq_rm(0, 10, 20)                              # drop 10 tokens ...
                m.seq_add(0, 20, -1, -10)                        # ... and slide the rest down: applied by the next llama_decode
                continue
            sid, pos = c
            layers = m.decode(t, pos, sid)
            outs.append(([layers[il]["out"].copy() for il in sorted(layers)], m.logits(MODEL["n_vocab"]).copy()))
        return outs
    cpu = run(0)
    R.load_backend(PLUGIN)
    gpu = run(99)
    # the plugin's launch-merging rules (ADD->RMS_NORM->MUL, SILU..MUL, ROPE..CPY, products on shared activations pulled together, K / V
    # products redirected to a side buffer) change WHICH launches run, never the arithmetic: one launch per graph node gives the same bits
    os.environ["GGML_MI355_NO_FUSE"] = "1"
    try:
        plain = run(99)
    finally:
        del os.environ["GGML_MI355_NO_FUSE"]
    R.set_gpu_layers(0)
    for (lg, logit_g), (lp, logit_p) in zip(gpu, plain):
        assert np.array_equal(logit_g, logit_p)
        assert all(np.array_equal(a, b) for a, b in zip(lg, lp))
    for ci, ((lc, logit_c), (lg, logit_g)) in enumerate(zip(cpu, gpu)):
        for il, (a, b) in enumerate(zip(lc, lg)):
            assert np.isfinite(b).all()
            if os.environ.get("PKV_PRINT_NMSE"):
                print("call", ci, "layer", il, "kqv_out nmse %.3e" % nmse(b, a), flush=True)
                continue
            # Q4_K weights: the CPU backend quantizes activations to q8_K (blocks of 256), the device to int8 per 32.  The synthetic
            # q4_K blocks (uniform random 6-bit scales and mins) give weights several times wider than N(0, 1/fan_in): peaky softmaxes that
            # amplify that rounding difference — measured 2e-4 (layer 0) .. 3.4e-3 (layer 1) with f16 KV, up to 1.6e-2 with q8_0 KV, with
            # or without the merged launches (bit-identical, checked above).  For this model the comparison with the CPU run is a sanity
            # bound; the product itself is held to 1e-4 / exact int8 arithmetic in tests/test_gpu_layer_ops.py and to the harness bar.
            assert nmse(b, a) <= (5e-2 if MODEL.get("wtype") in (12, 13) else 5e-4), (ci, il)
        # Q8_0 weights: the CPU backend rounds the activations to q8_0 before every product (0.4 % of the block maximum per
        # element, 14 products deep), the device keeps them in f32: the runs differ by the CPU path's own quantization noise
        # (Q4_K weights: q8_K activations of 256 on the CPU, int8 per 32 on the device — same kind of difference)
        bar = 2e-2 if MODEL.get("wtype") in (12, 13) else (2e-3 if MODEL.get("wtype") == 8 else (1e-4 if tk == 1 else 5e-4))
        if os.environ.get("PKV_PRINT_NMSE"):
            print("call", ci, "logits nmse %.3e" % nmse(logit_g, logit_c), flush=True)
            continue
        assert nmse(logit_g, logit_c) <= bar, (ci, bar)


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
@pytest.mark.parametrize("model,tk", [("tiny", 30), ("llama3", 30), ("hs256", 1), ("hs256", 8), ("hs256", 30), ("tiny", 3), ("tiny", 6), ("tiny", 7), ("tiny", 20), ("hs256", 6)])   # (f16 weights: both runs quantize the same K / V values)
def test_llama_decode_outside_the_paged_geometries(pkv, tmp_path, model, tk):
    """r03: what the paged kernels do not take runs through pkv_flash_attn_ext_flat inside the unmodified reference runtime — a bf16 KV cache (`-ctk bf16
    -ctv bf16`, /root/reference/common/arg.cpp:172-182: the KV store is the plugin's f32 -> bf16 strided CPY, the cache tensors stay flat bf16 rows) and head
    size 256 (f16 / q8_0 / bf16 cache); and the remaining cache types of that list — q4_1 (3), q5_0 (6), q5_1 (7), iq4_nl (20): the KV store is pkv_convert_f32's
    block quantizers, bit-exact with the CPU backend's.  Expected values: the CPU-only run of the same library, per-layer kqv_out NMSE <= 5e-4, logits as in the test above."""
    from llama_cpp_test_amd import gguf_synth as G
    MODEL = MODELS[model]
    path = str(tmp_path / "m.gguf")
    G.synth_llama(path, **MODEL)
    rng = np.random.default_rng(15 + tk)
    calls = [([0] * 100, list(range(100))), ([0] * 70 + [1] * 33, list(range(100, 170)) + list(range(33)))] + [([0, 1], [170 + i, 33 + i]) for i in range(3)]
    toks = [rng.integers(0, MODEL["n_vocab"], len(c[0])) for c in calls]

    def run(gpu_layers):
        R.set_gpu_layers(gpu_layers)
        m = R.RefModel(path, n_ctx=512, n_batch=128, n_seq_max=2, flash_attn=True, type_k=tk, type_v=tk, n_threads=8)
        outs = []
        for (sid, pos), t in zip(calls, toks):
            layers = m.decode(t, pos, sid)
            outs.append(([layers[il]["out"].copy() for il in sorted(layers)], m.logits(MODEL["n_vocab"]).copy()))
        return outs
    cpu = run(0)
    R.load_backend(PLUGIN)
    gpu = run(99)
    R.set_gpu_layers(0)
    for ci, ((lc, logit_c), (lg, logit_g)) in enumerate(zip(cpu, gpu)):
        for il, (a, b) in enumerate(zip(lc, lg)):
            assert np.isfinite(b).all() and nmse(b, a) <= 5e-4, (ci, il, nmse(b, a))
        assert nmse(logit_g, logit_c) <= (2e-3 if MODEL.get("wtype") == 8 else 5e-4), (ci, nmse(logit_g, logit_c))


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
@pytest.mark.parametrize("model", ["tiny", "llama3"])
def test_llama_decode_without_flash_attention_on_the_plugin(pkv, tmp_path, model):
    """a13: a `-fa 0` graph (llama-bench's default, /root/reference/examples/llama-bench/llama-bench.cpp:190-215) with the KV cache in the plugin's
    buffers: kq = mul_mat(k, q) -> soft_max_ext(kq, mask, scale) -> mul_mat(v, kq) -> cont, V stored TRANSPOSED by a strided CPY
    (/root/reference/src/llama-graph.cpp:1201-1239, :1387-1394) — every one of those nodes on the MI355X (pkv_mul_mat_strided, pkv_soft_max_ext_f32,
    pkv_cpy_strided), checked against the CPU-only run of the same library like the flash-attention test above"""
    from llama_cpp_test_amd import gguf_synth as G
    MODEL = MODELS[model]
    path = str(tmp_path / "m.gguf")
    G.synth_llama(path, **MODEL)
    rng = np.random.default_rng(6)
    calls = [([0] * 90, list(range(90))), ([0] * 60 + [1] * 40, list(range(90, 150)) + list(range(40)))] + [([0, 1], [150 + i, 40 + i]) for i in range(3)]
    toks = [rng.integers(0, MODEL["n_vocab"], len(c[0])) for c in calls]

    def run(gpu_layers):
        R.set_gpu_layers(gpu_layers)
        m = R.RefModel(path, n_ctx=512, n_batch=128, n_seq_max=2, flash_attn=False, type_k=1, type_v=1, n_threads=8)
        outs = []
        for (sid, pos), t in zip(calls, toks):
            layers = m.decode(t, pos, sid)
            outs.append(([layers[il]["out"].copy() for il in sorted(layers)], m.logits(MODEL["n_vocab"]).copy()))
        return outs
    cpu = run(0)
    R.load_backend(PLUGIN)
    gpu = run(99)
    R.set_gpu_layers(0)
    for ci, ((lc, logit_c), (lg, logit_g)) in enumerate(zip(cpu, gpu)):
        for il, (a, b) in enumerate(zip(lc, lg)):
            assert np.isfinite(b).all() and nmse(b, a) <= 5e-4, (ci, il, nmse(b, a))
        assert nmse(logit_g, logit_c) <= (2e-3 if MODEL.get("wtype") == 8 else 1e-4), ci


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
@pytest.mark.parametrize("tk", [1, 8])
def test_cache_tensor_is_its_own_pool(pkv, tmp_path, tk):
    """f3: inside the reference runtime the KV cache tensors of a flash-attention context ARE the paged pools (no second copy): the plugin re-lays a
    cache tensor out as 128-token tiles at its first attention call, the KV store writes tiles, and ggml_backend_tensor_get / _set translate whole
    cell rows.  Checked through the reference's own session API: a sequence saved with llama_state_seq_get_data after decoding on the plugin must be
    the reference's byte format (its CPU context reads it and continues with the same logits as a CPU context that decoded the same tokens, up to the
    kernels' rounding), and restoring it into a FRESH plugin context (flat rows written before the first attention call, then re-laid out) must
    continue bit for bit like the context it was saved from.  (GGML_MI355_KV_MIRROR=1 keeps r01's duplicate pool for A/B.)"""
    from llama_cpp_test_amd import gguf_synth as G
    MODEL = MODELS["llama3"]
    path = str(tmp_path / "m.gguf")
    G.synth_llama(path, **MODEL)
    rng = np.random.default_rng(12)
    n_p = 150                                                          # beyond one 128-cell block
    toks = rng.integers(0, MODEL["n_vocab"], n_p + 3)

    def fresh(gpu_layers):
        R.set_gpu_layers(gpu_layers)
        return R.RefModel(path, n_ctx=512, n_batch=128, n_seq_max=1, flash_attn=True, type_k=tk, type_v=tk, n_threads=8)

    def feed(m, lo, hi):
        for a in range(lo, hi, 100):
            b = min(hi, a + 100)
            m.decode(toks[a:b], list(range(a, b)), [0] * (b - a))
        return m.logits(MODEL["n_vocab"]).copy()

    R.load_backend(PLUGIN)
    g = fresh(99)
    feed(g, 0, n_p)
    blob = g.state_seq_get(0)                                          # tiles -> reference rows
    cont = [feed(g, n_p + i, n_p + i + 1) for i in range(3)]
    g2 = fresh(99)
    assert g2.state_seq_set(blob, 0) == len(blob)                      # reference rows -> (flat, then) tiles
    cont2 = [feed(g2, n_p + i, n_p + i + 1) for i in range(3)]
    assert all(np.array_equal(a, b) for a, b in zip(cont, cont2))
    assert len(g2.state_seq_get(0)) > len(blob)                        # three more cells
    # the blob is the reference's format: a CPU context continues from it
    c = fresh(0)
    assert c.state_seq_set(blob, 0) == len(blob)
    cpu_cont = [feed(c, n_p + i, n_p + i + 1) for i in range(3)]
    for a, b in zip(cont, cpu_cont):
        assert nmse(a, b) <= 2e-3
    R.set_gpu_layers(0)


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
def test_q4_k_m_mix_through_the_plugin_against_exact_arithmetic(pkv, O, tmp_path):
    """end-to-end Q4_K parity of the plugin path on REAL weights.  The model of tests/golden/make_decoder_golden.py — Gaussian weights quantized
    by the compiled reference's own q4_K / q6_K quantizers in the tensor mix of a Q4_K_M file — decoded by the unmodified reference runtime with every
    layer on the MI355X: the 150-token prompt goes through the MFMA products (pkv_gemm.hip, q4_K and q6_K tiles) and the MFMA prefill attention, the six
    decode steps through the decode-step products (pkv_gemv.hip) and the paged decode kernel.  Bars as in tests/test_gpu_decoder.py: as close to exact
    (f64, unquantized-activation) arithmetic as the reference's CPU backend is, within twice that distance of the reference's logits, the same greedy token wherever the margin is above the noise (the prompt's products round
    the activations to f16 where the reference rounds them to q8_K: different noise of the same size)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_decoder_golden", os.path.join(ROOT, "tests", "golden", "make_decoder_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    path = str(tmp_path / "q4km.gguf")
    g = np.load(os.path.join(ROOT, "tests", "golden", "decoder_logits.npz"))
    assert gen.build(path) == bytes(g["sha256"]).hex()
    tok, want, exact = g["tokens"], g["f16__logits"], g["exact__logits"]
    R.load_backend(PLUGIN)
    R.set_gpu_layers(99)
    m = R.RefModel(path, n_ctx=512, n_batch=256, n_seq_max=1, flash_attn=True, type_k=1, type_v=1, n_threads=8)
    nv = gen.MODEL["n_vocab"]
    got = []
    m.decode(tok[:gen.N_PROMPT], list(range(gen.N_PROMPT)), [0] * gen.N_PROMPT)
    got.append(m.logits(nv).copy())
    for i in range(gen.N_DECODE):
        p = gen.N_PROMPT + i
        m.decode(tok[p:p + 1], [p], [0])
        got.append(m.logits(nv).copy())
    R.set_gpu_layers(0)
    ref_noise = max(nmse(b, e) for b, e in zip(want, exact))
    assert len(got) == len(want) and 2e-4 < ref_noise < 5e-3
    assert max(nmse(a, e) for a, e in zip(got, exact)) <= 1.5 * ref_noise
    assert max(nmse(a, b) for a, b in zip(got, want)) <= 2.0 * ref_noise
    for a, b in zip(got, want):                                       # the same greedy token wherever the reference's own top-2 margin is above the noise
        top = np.sort(b)[-2:]
        if top[1] - top[0] > 6.0 * float(np.sqrt(np.mean((a - b) ** 2))):
            assert int(np.argmax(a)) == int(np.argmax(b))


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
@pytest.mark.parametrize("tk", [1, 8])
def test_defrag_moves_cells_of_a_paged_cache_tensor(pkv, tmp_path, tk):
    """llama-cli / llama-server run with defrag_thold 0.1 (/root/reference/common/common.h); once kv_self->n >= 2048 and the share of holes exceeds
    it, llama_decode queues a defrag and the NEXT kv_self_update builds ggml_cpy nodes between cell ranges of k_l / v_l (/root/reference/src/llama-context.cpp:668-711)
    — after defrag_prepare has already moved the cell metadata, and with graph_compute's status ignored (:773).  With the cache tensor being its own paged pool those
    copies are gather -> staging -> scatter of whole cells (run_cpy_cells).  2300 prompt tokens, 1500 of them removed from the middle (f16 K: plus the context shift
    that slides the tail down — the in-place K-shift on tiles), then decode steps across the defrag: per-layer kqv_out and logits against the CPU-only run of the same
    library with the same defrag_thold."""
    from llama_cpp_test_amd import gguf_synth as G
    MODEL = MODELS["tiny"]
    path = str(tmp_path / "m.gguf")
    G.synth_llama(path, **MODEL)
    rng = np.random.default_rng(9)
    n_prompt, lo, hi = 2300, 300, 1800
    calls = [([0] * min(128, n_prompt - p), list(range(p, min(p + 128, n_prompt)))) for p in range(0, n_prompt, 128)]
    calls.append("holes")
    p0 = n_prompt - (hi - lo) if tk == 1 else n_prompt           # f16: the tail was slid down by the removed count
    calls += [([0], [p0 + i]) for i in range(6)]                 # the first of these requests the defrag, the second runs it
    toks = [None if c == "holes" else rng.integers(0, MODEL["n_vocab"], len(c[0])) for c in calls]

    def run(gpu_layers, thold=0.1):
        R.set_gpu_layers(gpu_layers)
        R.set_defrag_thold(thold)
        try:
            m = R.RefModel(path, n_ctx=2560, n_batch=128, n_seq_max=1, flash_attn=True, type_k=tk, type_v=tk, n_threads=8)
        finally:
            R.set_defrag_thold(-1.0)
        outs = []
        for c, t in zip(calls, toks):
            if c == "holes":
                m.seq_rm(0, lo, hi)
                if tk == 1:
                    m.seq_add(0, hi, -1, -(hi - lo))
                continue
            sid, pos = c
            layers = m.decode(t, pos, sid)
            if len(sid) == 1:
                outs.append(([layers[il]["out"].copy() for il in sorted(layers)], m.logits(MODEL["n_vocab"]).copy()))
        return outs
    # The expected values.  f16 KV: the CPU-only run of the same library with the same defrag_thold.  q8_0 KV: the reference's CPU backend cannot run this graph —
    # its same-type contiguous copy multiplies the ELEMENT count by the BLOCK size (ggml_compute_forward_dup_same_cont, /root/reference/ggml/src/ggml-cpu/ggml-cpu.c:
    # 3111-3124: ne = ggml_nelements, nb0 = ggml_type_size = 34 for q8_0), a 32-fold overrun of both views that corrupts the heap (SIGSEGV in the next threadpool
    # teardown, reproduced in the dev container without the plugin) — so the expected values are the plugin's own run WITHOUT defragmentation: the same cells in
    # another order, the same attention up to summation order.
    if tk == 1:
        want = run(0)
        R.load_backend(PLUGIN)
    else:
        R.load_backend(PLUGIN)
        want = run(99, thold=-1.0)
    gpu = run(99)
    R.set_gpu_layers(0)
    assert len(want) == len(gpu) == 6
    for ci, ((lc, logit_c), (lg, logit_g)) in enumerate(zip(want, gpu)):
        for il, (a, b) in enumerate(zip(lc, lg)):
            assert np.isfinite(b).all() and nmse(b, a) <= 5e-4, (ci, il, nmse(b, a))
        assert nmse(logit_g, logit_c) <= 1e-4, (ci, nmse(logit_g, logit_c))


@pytest.mark.skipif(not (R.available() and os.path.exists(PLUGIN)), reason="needs oracle/_ref/libref_kv.so and libggml-mi355.so")
def test_packed_weights_round_trip_through_tensor_get_and_set(pkv, tmp_path):
    """K-quant tensors of a WEIGHTS buffer are re-laid inside their own bytes by the plugin's set_tensor (plane layout, csrc/pkv_wlayout.h); ggml_backend_tensor_get must
    hand the GGUF's bytes back — whole tensors, row-aligned ranges and ranges that start and end inside a row — and a partial ggml_backend_tensor_set must land where the
    GGUF order says.  (That the products on such tensors equal the CPU run is the llama3_q4k case of test_llama_decode_with_attention_on_the_plugin.)"""
    from llama_cpp_test_amd import gguf_synth as G
    MODEL = dict(MODELS["llama3_q4k"], k_mix=True)                 # q4_K + the q6_K tensors of a Q4_K_M file
    path = str(tmp_path / "m.gguf")
    G.synth_llama(path, **MODEL)
    _, tensors, _ = G.read_gguf(path)
    R.load_backend(PLUGIN)
    R.set_gpu_layers(99)
    try:
        m = R.RefModel(path, n_ctx=256, n_batch=64, n_seq_max=1, flash_attn=True, type_k=1, type_v=1, n_threads=4)
        seen = set()
        for name in ("blk.0.attn_q.weight", "blk.0.attn_v.weight", "blk.1.ffn_down.weight", "blk.0.ffn_gate.weight", "blk.1.attn_output.weight"):
            ne, t, _, data = tensors[name]
            want = bytes(np.array(data).tobytes())
            got, on_dev = m.weight_get(name)
            assert on_dev, name                                  # the layer lives in the plugin's buffer
            assert got == want, name
            seen.add(int(t))
            rb = len(want) // int(ne[1])
            for off, n in ((rb * 3, rb * 2), (rb * 5 + 7, rb + 11), (1, 5), (len(want) - 9, 9)):
                assert m.weight_get(name, off, n)[0] == want[off:off + n], (name, off, n)
            patch = bytes(np.random.default_rng(len(name)).integers(0, 256, rb + 13, dtype=np.uint8))
            off = rb * 2 + 5
            assert m.weight_set(name, patch, off) == len(patch)
            assert m.weight_get(name)[0] == want[:off] + patch + want[off + len(patch):], name
            assert m.weight_set(name, want[off:off + len(patch)], off) == len(patch)      # restore
            assert m.weight_get(name)[0] == want
        assert seen == {12, 14}
    finally:
        R.set_gpu_layers(0)
