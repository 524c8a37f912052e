"""-m gpu: the fused decode step (pkv_attn_decode_step = RoPE(Q) + RoPE(K) + KV store + paged decode in one launch).

Bar: bit-identical to the two-call path (pkv_qkv_prepare + pkv_attn_decode) — outputs AND pool bytes — which is
itself pinned against the oracle elsewhere; plus a direct check against the oracle (rope -> from_float -> pool_store ->
attn_paged), attention output max-abs <= 1e-3 / NMSE <= 5e-4 (tests/test-backend-ops.cpp:3216)."""
import numpy as np
import pytest
import torch

from gpu_util import dev, forced_tiles, make_paged_case, nmse

pytestmark = pytest.mark.gpu
TYPES = {"f16": 1, "q8_0": 8, "q4_0": 2}


def _case(O, rng, lens, D, nh, nkv, tk, tv):
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tv)
    n = len(lens)
    q = rng.uniform(-1, 1, (n, nh, D)).astype(np.float32)
    kc = rng.uniform(-1, 1, (n, nkv, D)).astype(np.float32)
    vc = rng.uniform(-1, 1, (n, nkv, D)).astype(np.float32)
    pos = (sl - 1).astype(np.int32)
    return kp, vp, bt, sl, q, kc, vc, pos


def _two_call(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, *, nkv, tk, tv):
    dq, dkp, dvp = dev(q), dev(kp), dev(vp)
    last = sl - 1
    sb = bt[np.arange(len(sl)), last // 128].astype(np.int32)
    so = (last % 128).astype(np.int32)
    if rope is not None:
        pkv.qkv_prepare(dq, dev(kc), dev(vc), dkp, dvp, type_k=tk, type_v=tv, slot_block=dev(sb), slot_off=dev(so), block_size=128,
                        pos=dev(pos), rope=rope)
    else:
        pkv.kv_store(dev(kc), dev(vc), dkp, dvp, type_k=tk, type_v=tv, slot_block=dev(sb), slot_off=dev(so), block_size=128)
    D = q.shape[2]
    out = pkv.attn_decode(dq, dkp, dvp, dev(bt), dev(sl), n_head_kv=nkv, type_k=tk, type_v=tv, block_size=128, scale=D ** -0.5,
                          max_seq_len=int(sl.max()))
    torch.cuda.synchronize()
    return out.cpu().numpy(), dkp.cpu().numpy(), dvp.cpu().numpy()


def _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, *, nkv, tk, tv, shared_cache=False):
    dq, dkp, dvp = dev(q), dev(kp), dev(vp)
    D = q.shape[2]
    # shared_cache: the (cos, sin) rows are built once (pkv_rope_cache) and handed to the step, as a 32-layer loop does
    cache = pkv.rope_cache(dev(pos), rope, D) if (shared_cache and rope is not None) else None
    out = pkv.attn_decode_step(dq, dev(kc), dev(vc), None if cache is not None else dev(pos), rope, dkp, dvp, dev(bt), dev(sl),
                               n_head_kv=nkv, type_k=tk, type_v=tv, block_size=128, scale=D ** -0.5, max_seq_len=int(sl.max()),
                               rope_cache=cache)
    torch.cuda.synchronize()
    assert np.array_equal(dq.cpu().numpy(), q), "q must be left un-rotated"
    return out.cpu().numpy(), dkp.cpu().numpy(), dvp.cpu().numpy()


@pytest.mark.parametrize("tk,tv", [("f16", "f16"), ("q8_0", "q8_0"), ("q4_0", "q4_0"), ("q8_0", "f16"), ("f16", "q4_0")])
@pytest.mark.parametrize("D,nh,nkv", [(128, 32, 8), (128, 8, 1), (128, 4, 4), (64, 32, 4), (128, 4, 2), (64, 8, 8)])
@pytest.mark.parametrize("mode", [0, 2])
def test_step_bit_identical_to_two_calls(pkv, O, tk, tv, D, nh, nkv, mode):
    tk, tv = TYPES[tk], TYPES[tv]
    rng = np.random.default_rng(D + nh * 3 + tk + tv * 5 + mode)
    lens = [1, 2, 3, 8, 31, 127, 128, 129, 130, 255, 256, 257, 300, 641]   # new token first / last / middle of a block, of a V group
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, lens, D, nh, nkv, tk, tv)
    rope = pkv.rope_params(n_dims=D, mode=mode, freq_base=10000.0)
    a = _two_call(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=tk, tv=tv)
    b = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=tk, tv=tv, shared_cache=bool((D + nh + mode) % 4 == 0))
    assert np.isfinite(b[0]).all()
    assert np.array_equal(a[1], b[1]), "K pool bytes differ"
    assert np.array_equal(a[2], b[2]), "V pool bytes differ"
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), np.abs(a[0] - b[0]).max()


def test_step_without_rope_and_partial_rotation(pkv, O):
    rng = np.random.default_rng(5)
    D, nh, nkv = 128, 8, 2
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, [5, 128, 200, 513], D, nh, nkv, 8, 8)
    for rope in (None, pkv.rope_params(n_dims=64, mode=0, freq_base=500000.0, freq_scale=0.5, ext_factor=1.0, attn_factor=1.1,
                                       n_ctx_orig=4096)):
        a = _two_call(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=8, tv=8)
        for shared in (False, True):
            b = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=8, tv=8, shared_cache=shared)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))


def test_rope_cache_rows(pkv, O):
    """pkv_rope_cache == the (cos, sin) the oracle's rope applies: rotating the unit pair (1, 0) reads them back"""
    D, n = 128, 6
    pos = np.array([0, 1, 17, 4095, 16383, 100000], np.int32)
    for kw in (dict(n_dims=128, mode=0, freq_base=500000.0), dict(n_dims=64, mode=0, freq_base=10000.0, freq_scale=0.25, ext_factor=1.0,
                                                                   attn_factor=1.2, n_ctx_orig=4096)):
        rp = pkv.rope_params(**kw)
        got = pkv.rope_cache(dev(pos), rp, D).cpu().numpy()
        x = np.zeros((n, 1, D), np.float32)
        x[:, :, 0::2] = 1.0                                        # NORM pairs (1, 0) -> (cos, sin)
        ref = O.rope(x, pos, **kw)[:, 0]
        nd = kw["n_dims"]
        assert np.abs(got[:, :nd // 2, 0] - ref[:, 0:nd:2]).max() <= 2e-6
        assert np.abs(got[:, :nd // 2, 1] - ref[:, 1:nd:2]).max() <= 2e-6
        assert np.array_equal(got[:, nd // 2:, 0], np.ones_like(got[:, nd // 2:, 0])) and not got[:, nd // 2:, 1].any()


@pytest.mark.parametrize("tk", ["f16", "q8_0", "q4_0"])
def test_step_matches_oracle(pkv, O, tk):
    """the oracle's own chain: rope_f32 -> from_float -> pool_store -> flash_attn_ext over the gathered cells"""
    t = TYPES[tk]
    rng = np.random.default_rng(77 + t)
    D, nh, nkv = 128, 32, 8
    lens = [1, 129, 384, 1000]
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, lens, D, nh, nkv, t, t)
    rope = pkv.rope_params(n_dims=D, mode=0, freq_base=500000.0)
    got, gk, gv = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=t, tv=t)
    okw = dict(n_dims=D, mode=0, freq_base=500000.0)
    qr = O.rope(q, pos, **okw)
    kr = O.rope(kc, pos, **okw)
    last = sl - 1
    sb, so = bt[np.arange(len(sl)), last // 128].astype(np.int32), (last % 128).astype(np.int32)
    O.pool_store(kp, t, 0, O.from_float(t, kr), sb, so, D=D, block_size=128)
    O.pool_store(vp, t, 1, O.from_float(t, vc), sb, so, D=D, block_size=128)
    ref = O.attn_paged(qr[:, :, None, :], kp, vp, bt, sl, n_q=1, type_k=t, type_v=t, block_size=128, scale=D ** -0.5)[:, 0]
    assert np.array_equal(gv, vp), "V pool bytes differ from the oracle's store"
    # K went through RoPE on the device: bit-identical to the CPU's rows (libm-exact trig, r02)
    for s_, n in enumerate(lens):
        a = O.pool_gather(gk, t, 0, bt[s_], n, D=D, block_size=128)
        b = O.pool_gather(kp, t, 0, bt[s_], n, D=D, block_size=128)
        assert np.array_equal(a, b)
    assert np.abs(got - ref).max() <= 1e-3
    assert nmse(got, ref) <= 5e-4


@pytest.mark.parametrize("tk,tv", [("f16", "f16"), ("q8_0", "q8_0"), ("q4_0", "f16")])
@pytest.mark.parametrize("D,nh,nkv", [(128, 32, 8), (128, 8, 1), (64, 4, 4), (64, 8, 1), (128, 4, 2)])
def test_step_with_in_kernel_merge(pkv, O, tk, tv, D, nh, nkv):
    """PKV_ATTN_FUSED_COMBINE on the decode step: RoPE + KV store + attention + merge in ONE launch.  Same pools (bit-exact),
    outputs equal to the two-launch merge up to the summation order of the partials (<= 2e-6); the workspace's wrapping
    arrival tickets serve launch after launch, more than 32 splits take several merge passes."""
    tk, tv = TYPES[tk], TYPES[tv]
    rng = np.random.default_rng(D + nh + tk * 3 + tv)
    lens = [1, 127, 128, 129, 640, 4096, 4500, 2000]
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, lens, D, nh, nkv, tk, tv)
    rope = pkv.rope_params(n_dims=D, mode=0, freq_base=500000.0)
    with forced_tiles(1):     # (the two-launch path in its one-tile-per-workgroup form: the partition the in-kernel merge keeps)
        base = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=tk, tv=tv)
    ws = pkv.attn_workspace(len(lens), nh, D, int(sl.max()), 128)
    for it in range(3):
        dq, dkp, dvp = dev(q), dev(kp), dev(vp)
        out = pkv.attn_decode_step(dq, dev(kc), dev(vc), dev(pos), rope, dkp, dvp, dev(bt), dev(sl), n_head_kv=nkv, type_k=tk, type_v=tv,
                                   block_size=128, scale=D ** -0.5, max_seq_len=int(sl.max()), workspace=ws, flags=pkv.ATTN_FUSED_COMBINE)
        torch.cuda.synchronize()
        assert np.array_equal(dkp.cpu().numpy(), base[1]) and np.array_equal(dvp.cpu().numpy(), base[2]), it
        # quantized KV in a launch with >= 1024 workgroups: the two-launch kernel gives a wave 32 tokens, the in-kernel merge keeps 16 (other wave
        # maxima -> the f16 weights P * d_V round differently; both inside the oracle bars)
        assert np.abs(out.cpu().numpy() - base[0]).max() <= (2e-6 if (tk == 1 and tv == 1) else 1e-4), it


def test_step_rejects_unsupported(pkv, O):
    rng = np.random.default_rng(9)
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, [100], 128, 4, 4, 1, 1)
    with pytest.raises(pkv.PkvError):
        pkv.attn_decode_step(dev(q), dev(kc), dev(vc), dev(pos), None, dev(kp), dev(vp), dev(bt), dev(sl), n_head_kv=4, type_k=1,
                             type_v=1, block_size=128, scale=1.0, max_seq_len=100, logit_softcap=30.0)


def test_step_with_llama3_rope_frequency_factors(pkv, O):
    """Llama-3.1-style `rope_freqs` (ggml_rope_ext src[2], /root/reference/src/llama-model.cpp:11431-11443): the per-step RoPE cache
    carries them, so the fused step must still equal the two-call path bit for bit and the oracle within tolerance"""
    rng = np.random.default_rng(314)
    D, nh, nkv = 128, 32, 8
    lens = [5, 129, 700]
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, lens, D, nh, nkv, 8, 8)
    ff = (1.0 + 7.0 * rng.uniform(0, 1, D // 2) ** 3).astype(np.float32)          # factors between 1 and 8, like the llama-3 table
    rope = pkv.rope_params(n_dims=D, mode=0, freq_base=500000.0, freq_factors=dev(ff))
    a = _two_call(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=8, tv=8)
    for shared in (False, True):
        b = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=8, tv=8, shared_cache=shared)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    qr = O.rope(q, pos, ff, n_dims=D, freq_base=500000.0)
    kr = O.rope(kc, pos, ff, n_dims=D, freq_base=500000.0)
    last = sl - 1
    sb, so = bt[np.arange(len(sl)), last // 128].astype(np.int32), (last % 128).astype(np.int32)
    O.pool_store(kp, 8, 0, O.from_float(8, kr), sb, so, D=D, block_size=128)
    O.pool_store(vp, 8, 1, O.from_float(8, vc), sb, so, D=D, block_size=128)
    ref = O.attn_paged(qr[:, :, None, :], kp, vp, bt, sl, n_q=1, type_k=8, type_v=8, block_size=128, scale=D ** -0.5)[:, 0]
    assert np.abs(b[0] - ref).max() <= 1e-3 and nmse(b[0], ref) <= 5e-4


@pytest.mark.parametrize("D,nh,nkv", [(128, 32, 8), (64, 32, 4), (128, 4, 2)])
def test_step_leaves_the_activation_image_of_its_output(pkv, O, D, nh, nkv):
    """pkv_attn_args::out_image (r03): the merge launch also writes every out row as the q8_K activation image of the decode-step products — the
    bytes pkv_quantize_act makes of that row — and a product fed the image equals the product that quantizes the f32 row itself, bit for bit."""
    rng = np.random.default_rng(17 + D + nh)
    lens = [1, 129, 640, 4096, 0 + 300]
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, lens, D, nh, nkv, 1, 1)
    rope = pkv.rope_params(n_dims=D, mode=0, freq_base=10000.0)
    K = nh * D
    n = len(lens)
    nbytes = pkv.lib().pkv_act_image_bytes(K)
    img = torch.zeros((n, nbytes), dtype=torch.uint8, device="cuda")
    plain = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=1, tv=1, shared_cache=True)
    dq, dkp, dvp, dpos = dev(q), dev(kp), dev(vp), dev(pos)
    out = pkv.attn_decode_step(dq, dev(kc), dev(vc), None, rope, dkp, dvp, dev(bt), dev(sl), n_head_kv=nkv, type_k=1, type_v=1, block_size=128,
                               scale=D ** -0.5, max_seq_len=int(sl.max()), rope_cache=pkv.rope_cache(dpos, rope, D), out_image=img)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), plain[0].view(np.uint32))          # the f32 output is what it was
    for s in range(n):
        want = pkv.quantize_act(out[s].reshape(-1).contiguous())
        torch.cuda.synchronize()
        assert np.array_equal(img[s].cpu().numpy(), want.cpu().numpy()), s
    # a product on the image == the product on the f32 row (q4_K rows, residual epilogue: the wo product of a layer)
    if K % 256 == 0 and K >= 1024:
        from llama_cpp_test_amd import gguf_synth as G
        M = 512
        raw = np.frombuffer(G.random_q4_K(rng, M, K), np.uint8).copy()
        w = dev(raw)
        res = dev(rng.standard_normal(M).astype(np.float32))
        a = torch.zeros(M, device="cuda")
        b = torch.zeros(M, device="cuda")
        pkv.gemv_q([w], [12], [M], K, None, [a], residual=res, x=out[3].reshape(-1).contiguous())
        pkv.gemv_q([w], [12], [M], K, img[3], [b], residual=res, x=out[3].reshape(-1).contiguous())
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
