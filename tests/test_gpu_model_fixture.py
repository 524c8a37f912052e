"""-m gpu: the attention path inside a REAL model.  tests/golden/model_attn.npz holds, for every llama_decode call of a
scenario run by the reference itself on a synthetic random-weight llama GGUF (tests/golden/make_model_golden.py), each
layer's un-rotated Q/K/V projections and the attention output kqv_out.  The GPU path — block manager, RoPE, KV store in
f16 / q8_0, MFMA prefill, fused decode step — is driven by AttnPathRunner configured FROM THE GGUF (the minimal loader) and
must return the reference's attention outputs: batched ragged prompt (one sequence longer than a block), interleaved decode
steps, a chunked second prompt, another decode step.

Bar: NMSE <= 5e-4 (tests/test-backend-ops.cpp:3216).  Max-abs is noise-aware, as in tests/test_gpu_prefill.py: the fixture
also carries the exact (f64) attention over the cells the cache really holds, and the reference — which accumulates f16 V in f16
(ggml-cpu.c:12380-12395) and, with q8_0 K, quantizes Q to q8_0 — sits up to 4.8e-3 (f16) / 1.2e-2 (q8_0) away from it on this
model's O(1)-O(3) outputs.  The GPU result may differ from the reference by the reference's own distance to exact plus 1e-3, and
must be no further from exact than the reference is (plus 1e-3 where a RoPE'd K element lands on a quantization boundary under
device cosf/sinf).  tests/test_gguf.py checks the same fixture against the CPU oracle bit for bit."""
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, nmse

pytestmark = pytest.mark.gpu
MODEL = dict(n_layer=2, n_embd=128, n_head=2, n_head_kv=1, n_ff=256, n_vocab=64, n_ctx=1024, rope_freq_base=10000.0, seed=4321)


@pytest.mark.parametrize("name,t", [("f16", 1), ("q8_0", 8)])
def test_runner_reproduces_the_reference_models_attention(pkv, tmp_path, name, t):
    from llama_cpp_test_amd import gguf_synth as G
    from llama_cpp_test_amd.runner import AttnPathRunner
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_attn.npz"))
    path = str(tmp_path / "synthetic-llama.gguf")
    G.synth_llama(path, **MODEL)                                  # same hyper-parameters as the fixture's model (weights unused here)
    r = AttnPathRunner.from_gguf(path, type_k=t, type_v=t, n_blocks=6, n_seq_max=2, max_batch_tokens=256)
    assert (r.L, r.H, r.HK, r.D) == (2, 2, 1, 64)
    worst = 0.0
    for ci in range(int(g["n_calls"])):
        seq, pos = g[f"{name}__c{ci}__seq"], g[f"{name}__c{ci}__pos"]
        q, k, v = ([dev(g[f"{name}__c{ci}__l{il}__{x}"]) for il in range(2)] for x in ("q", "k", "v"))
        outs = r.decode(list(seq), list(pos), q, k, v)
        torch.cuda.synchronize()
        for il in range(2):
            ref = g[f"{name}__c{ci}__l{il}__out"]
            got = outs[il].cpu().numpy()
            assert np.isfinite(got).all()
            assert nmse(got, ref) <= 5e-4, (ci, il)
            exact = g[f"{name}__c{ci}__l{il}__exact"]
            ref_err = float(np.abs(ref - exact).max())
            worst = max(worst, float(np.abs(got - ref).max()))
            assert np.abs(got - ref).max() <= ref_err + 1e-3, (ci, il, ref_err)
            assert np.abs(got - exact).max() <= ref_err + 1e-3, (ci, il, ref_err)
    assert r.cache.seq_len(0) == 134 and r.cache.seq_len(1) == 64
    r.close()
