"""-m gpu: batched-prefill paged attention (MFMA, through the C ABI) against the CPU oracle.

Bar: f16 KV — NMSE <= 5e-4 vs the reference semantics (tests/test-backend-ops.cpp:3216); max-abs distance to the
reference no larger than the reference's own distance to exact (f64) arithmetic + 3e-4 (the reference accumulates f16-V
in f16: on short rows with O(1) outputs its own error reaches 1e-3), and at least as close to exact as the reference.  Quantized K: the Q rows take the
reference's q8_0 rounding (ggml-cpu.c:12343, r03), so the output is held to max-abs <= 1e-3 and NMSE <= 5e-4 against the reference semantics, and is never further
from the f64 result than the reference is by more than the f16 rounding of the MFMA operands."""
import numpy as np
import pytest
import torch

from gpu_util import dev, make_paged_case, nmse

pytestmark = pytest.mark.gpu


def run_prefill(pkv, O, rng, *, lens, n_qs, D, nh, nkv, tk, tv, softcap=0.0, bs=128):
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tv, block_size=bs)
    q_start = np.concatenate([[0], np.cumsum(n_qs)]).astype(np.int32)
    n_tok = int(q_start[-1])
    q = rng.uniform(-1, 1, (n_tok, nh, D)).astype(np.float32)
    out = pkv.attn_prefill(dev(q), dev(kp), dev(vp), dev(bt), dev(sl), dev(q_start), n_head_kv=nkv, type_k=tk, type_v=tv,
                           block_size=bs, scale=D ** -0.5, max_seq_len=int(sl.max()), max_q_per_seq=int(max(n_qs)),
                           logit_softcap=softcap)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    refs, exacts = [], []
    for s, (n, nq) in enumerate(zip(lens, n_qs)):
        kc = O.pool_gather(kp, tk, 0, bt[s], n, D=D, block_size=bs)
        vc = O.pool_gather(vp, tv, 1, bt[s], n, D=D, block_size=bs)
        qs = q[q_start[s]:q_start[s + 1]].transpose(1, 0, 2)               # [n_head][n_q][D]
        mask = np.zeros((nq, n), np.float16)
        for j in range(nq):
            mask[j, n - nq + j + 1:] = -np.inf
        refs.append(O.flash_attn_ext(qs, kc, vc, mask, type_k=tk, type_v=tv, scale=D ** -0.5, softcap=softcap, n_threads=8))
        exacts.append(O.flash_attn_f64(qs, kc, vc, mask, type_k=tk, type_v=tv, scale=D ** -0.5, softcap=softcap))
    return got, np.concatenate(refs), np.concatenate(exacts)


@pytest.mark.parametrize("D,nh,nkv", [(128, 32, 8), (128, 8, 1), (128, 4, 4), (64, 32, 4), (128, 4, 2), (128, 32, 2), (64, 32, 1)])   # (.., 32, 2) / (.., 32, 1): gqa 16 / 32
def test_prefill_f16_matches_oracle(pkv, O, D, nh, nkv):
    rng = np.random.default_rng(D + nh + nkv)
    lens = [1, 31, 32, 33, 100, 128, 129, 300, 512]          # whole prompt in one ubatch: n_q == seq_len
    got, ref, exact = run_prefill(pkv, O, rng, lens=lens, n_qs=lens, D=D, nh=nh, nkv=nkv, tk=1, tv=1)
    assert np.isfinite(got).all()
    ref_err = np.abs(ref - exact).max()
    assert np.abs(got - ref).max() <= ref_err + 3e-4
    assert nmse(got, ref) <= 5e-4
    assert np.abs(got - exact).max() <= max(ref_err, 2e-4)    # never further from exact than the reference is


def test_prefill_chunked_with_past(pkv, O):
    """second ubatch of a long prompt: n_q new tokens attend to seq_len - n_q cached ones plus themselves (causal)"""
    rng = np.random.default_rng(2)
    lens, n_qs = [600, 257, 1000, 64], [88, 1, 512, 33]
    got, ref, exact = run_prefill(pkv, O, rng, lens=lens, n_qs=n_qs, D=128, nh=8, nkv=2, tk=1, tv=1)
    ref_err = np.abs(ref - exact).max()
    assert np.abs(got - ref).max() <= ref_err + 3e-4 and nmse(got, ref) <= 5e-4
    assert np.abs(got - exact).max() <= max(ref_err, 2e-4)


@pytest.mark.parametrize("tk,tv", [(8, 8), (2, 2), (8, 1), (1, 8)])
def test_prefill_quantized_kv(pkv, O, tk, tv):
    rng = np.random.default_rng(10 * tk + tv)
    lens, n_qs = [200, 129, 64, 512], [200, 40, 64, 130]
    got, ref, exact = run_prefill(pkv, O, rng, lens=lens, n_qs=n_qs, D=128, nh=8, nkv=2, tk=tk, tv=tv)
    assert np.isfinite(got).all()
    assert nmse(got, ref) <= 5e-4
    assert np.abs(got - ref).max() <= 1e-3                   # quantized K: Q takes the reference's q8_0 rounding (r03)
    # K/V (and the q8_0-rounded Q) reach the MFMA as f16 operands (2^-11 relative rounding per element): never further from exact arithmetic than
    # the reference itself by more than that
    assert np.abs(got - exact).max() <= np.abs(ref - exact).max() + 5e-4


def test_prefill_equals_decode_for_single_query(pkv, O):
    """n_q == 1 per sequence: the MFMA prefill path and the split-KV decode path must agree"""
    rng = np.random.default_rng(8)
    lens = [300, 129, 640]
    D, nh, nkv = 128, 32, 8
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=1, type_v=1)
    q = rng.uniform(-1, 1, (3, nh, D)).astype(np.float32)
    a = pkv.attn_prefill(dev(q), dev(kp), dev(vp), dev(bt), dev(sl), dev(np.arange(4, dtype=np.int32)), n_head_kv=nkv, type_k=1,
                         type_v=1, block_size=128, scale=D ** -0.5, max_seq_len=640, max_q_per_seq=1)
    b = pkv.attn_decode(dev(q), dev(kp), dev(vp), dev(bt), dev(sl), n_head_kv=nkv, type_k=1, type_v=1, block_size=128,
                        scale=D ** -0.5, max_seq_len=640)
    torch.cuda.synchronize()
    assert np.abs(a.cpu().numpy() - b.cpu().numpy()).max() <= 2e-4


def test_prefill_pp512_properties(pkv, O):
    """BASELINE config 3 prompt shape (pp512, 8B geometry): causality — perturbing the K/V of position p only changes
    query rows >= p — and block-permutation invariance, without running the CPU on the full size"""
    rng = np.random.default_rng(5)
    D, nh, nkv, n = 128, 32, 8, 512
    kp, vp, bt, sl = make_paged_case(O, rng, lens=[n], D=D, n_head=nh, n_head_kv=nkv, type_k=1, type_v=1, spare_blocks=4)
    q = rng.uniform(-1, 1, (n, nh, D)).astype(np.float32)
    qs = dev(np.array([0, n], np.int32))

    def run(kpool, vpool, table):
        o = pkv.attn_prefill(dev(q), dev(kpool), dev(vpool), dev(table), dev(sl), qs, n_head_kv=nkv, type_k=1, type_v=1,
                             block_size=128, scale=D ** -0.5, max_seq_len=n, max_q_per_seq=n)
        torch.cuda.synchronize()
        return o.cpu().numpy()
    base = run(kp, vp, bt)
    perm = rng.permutation(kp.shape[0]).astype(np.int32)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    assert np.array_equal(base, run(kp[perm], vp[perm], inv[bt]))
    p = 300
    kp2 = kp.copy()
    rows = O.from_float(1, rng.uniform(-1, 1, (1, nkv, D)).astype(np.float32))
    O.pool_store(kp2, 1, 0, rows, bt[0][[p // 128]], np.array([p % 128], np.int32), D=D, block_size=128)
    pert = run(kp2, vp, bt)
    assert np.array_equal(pert[:p], base[:p])
    assert np.abs(pert[p:] - base[p:]).max() > 0
    # a handful of rows against the oracle
    kc = O.pool_gather(kp, 1, 0, bt[0], n, D=D, block_size=128)
    vc = O.pool_gather(vp, 1, 1, bt[0], n, D=D, block_size=128)
    for j in (0, 31, 32, 255, 511):
        mask = np.zeros((1, n), np.float16)
        mask[0, j + 1:] = -np.inf
        ref = O.flash_attn_ext(q[j][:, None, :], kc, vc, mask, type_k=1, type_v=1, scale=D ** -0.5)[0]
        assert np.abs(base[j] - ref).max() <= 1e-3


@pytest.mark.parametrize("tk", [1, 8, 2])
@pytest.mark.parametrize("max_bias,softcap", [(0.0, 0.0), (8.0, 0.0), (0.0, 10.0)])
def test_prefill_general_mask_noncausal(pkv, O, tk, max_bias, softcap):
    """the ggml_flash_attn_ext contract as tests/test-backend-ops.cpp:3197-3272 exercises it: every query sees every cell,
    an additive f16 mask with arbitrary values (and -inf holes), ALiBi slopes (max_bias), logit soft-cap"""
    rng = np.random.default_rng(int(tk * 100 + max_bias + softcap))
    D, nh, nkv = 128, 8, 2
    # n_kv 509: mask rows 1018 bytes apart — the kernel's one-cell-per-load path (rows off 8-byte boundaries); 512: its 8-byte loads
    for n_q, n_kv in ((1, 512), (3, 512), (35, 512), (35, 509)):
        kp, vp, bt, sl = make_paged_case(O, rng, lens=[n_kv], D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tk)
        q = rng.uniform(-1, 1, (n_q, nh, D)).astype(np.float32)
        mask = rng.uniform(-1, 1, (n_q, n_kv)).astype(np.float16)
        mask[:, rng.integers(0, n_kv, 20)] = -np.inf
        out = pkv.attn_prefill(dev(q), dev(kp), dev(vp), dev(bt), dev(sl), dev(np.array([0, n_q], np.int32)), n_head_kv=nkv,
                               type_k=tk, type_v=tk, block_size=128, scale=D ** -0.5, max_seq_len=n_kv, max_q_per_seq=n_q,
                               logit_softcap=softcap, mask=dev(mask), max_bias=max_bias, flags=pkv.ATTN_NONCAUSAL)
        torch.cuda.synchronize()
        kc = O.pool_gather(kp, tk, 0, bt[0], n_kv, D=D, block_size=128)
        vc = O.pool_gather(vp, tk, 1, bt[0], n_kv, D=D, block_size=128)
        ref = O.flash_attn_ext(q.transpose(1, 0, 2), kc, vc, mask, type_k=tk, type_v=tk, scale=D ** -0.5, max_bias=max_bias,
                               softcap=softcap)
        assert nmse(out.cpu().numpy(), ref) <= 5e-4, (n_q,)
        # f16 KV: the reference accumulates V in f16 (ggml-cpu.c:12374-12388), the kernel in f32; quantized K: same Q rounding, f32 V accumulation on both sides
        assert np.abs(out.cpu().numpy() - ref).max() <= (2e-3 if tk == 1 else 1e-3)


@pytest.mark.parametrize("tk,nkv,nh", [(1, 2, 8), (8, 1, 8), (1, 8, 32)])
def test_prefill_split_kv_equals_unsplit(pkv, O, tk, nkv, nh):
    """Small launches over a long past (a 512-token ubatch deep into a long prompt) share each query tile's KV range over
    several workgroups and merge the (m, l, O) partials.  The split is opportunistic — it needs the scratch
    pkv_attn_prefill_workspace_bytes asks for — so the same call with a 16-byte workspace runs unsplit.  Both round P to f16
    relative to their own running maximum, so they agree to the f16 weight quantum (1e-4 on O(0.1) outputs), not bit for
    bit; the split result itself must meet the oracle bar.  Ragged batch: one sequence shorter than a share."""
    rng = np.random.default_rng(tk + nkv)
    D = 128
    lens, n_qs = [3000, 70, 1500], [96, 64, 33]
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tk)
    q_start = np.concatenate([[0], np.cumsum(n_qs)]).astype(np.int32)
    q = rng.uniform(-1, 1, (int(q_start[-1]), nh, D)).astype(np.float32)
    kw = dict(n_head_kv=nkv, type_k=tk, type_v=tk, block_size=128, scale=D ** -0.5, max_seq_len=int(sl.max()), max_q_per_seq=max(n_qs))
    args = (dev(q), dev(kp), dev(vp), dev(bt), dev(sl), dev(q_start))
    split = pkv.attn_prefill(*args, **kw)                                     # default workspace: split-KV
    plain = pkv.attn_prefill(*args, workspace=torch.zeros(16, dtype=torch.uint8, device="cuda"), **kw)
    torch.cuda.synchronize()
    a, b = split.cpu().numpy(), plain.cpu().numpy()
    assert np.isfinite(a).all()
    assert np.abs(a - b).max() <= 1e-4
    for s, (n, nq) in enumerate(zip(lens, n_qs)):
        kc = O.pool_gather(kp, tk, 0, bt[s], n, D=D, block_size=128)
        vc = O.pool_gather(vp, tk, 1, bt[s], n, D=D, block_size=128)
        mask = np.zeros((nq, n), np.float16)
        for j in range(nq):
            mask[j, n - nq + j + 1:] = -np.inf
        qs = q[q_start[s]:q_start[s + 1]].transpose(1, 0, 2)
        exact = O.flash_attn_f64(qs, kc, vc, mask, type_k=tk, type_v=tk, scale=D ** -0.5)
        ref = O.flash_attn_ext(qs, kc, vc, mask, type_k=tk, type_v=tk, scale=D ** -0.5, n_threads=8)
        got = a[q_start[s]:q_start[s + 1]]                                   # oracle output is [n_q][n_head][D] too
        assert nmse(got, ref) <= 5e-4
        ref_err = np.abs(ref - exact).max()
        assert np.abs(got - exact).max() <= max(1e-3, ref_err + 3e-4)        # (quantized K: Q carries the reference's q8_0 rounding, as far from exact as the reference is)
        if tk != 1:
            assert np.abs(got - ref).max() <= 1e-3


def test_prefill_64_queries_per_wave_variant(pkv, O, monkeypatch):
    """the opt-in 64-queries-per-wave kernel (PKV_PREFILL_QB=2, DESIGN.md §3: measured, slower, kept as an experiment) computes what the
    32-query kernel computes: identical bits where both walk the same KV tiles per query, <= 1e-4 on ragged lengths (a 64-query tile
    reaches the diagonal in a different tile than a 32-query tile does, so the running maximum is updated in a different order)"""
    rng = np.random.default_rng(64)
    D, nh, nkv = 128, 32, 8
    lens = [1, 31, 33, 64, 65, 100, 700, 1000, 512, 512]
    n_qs = [1, 31, 33, 64, 65, 100, 300, 129, 512, 512]
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=1, type_v=1)
    q = rng.uniform(-1, 1, (sum(n_qs), nh, D)).astype(np.float32)
    qs = np.concatenate([[0], np.cumsum(n_qs)]).astype(np.int32)
    outs = []
    for qb in ("1", "2"):
        monkeypatch.setenv("PKV_PREFILL_QB", qb)
        o = pkv.attn_prefill(dev(q), dev(kp), dev(vp), dev(bt), dev(sl), dev(qs), n_head_kv=nkv, type_k=1, type_v=1, block_size=128,
                             scale=D ** -0.5, max_seq_len=max(lens), max_q_per_seq=max(n_qs))
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
    assert np.isfinite(outs[1]).all()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-4
    assert np.array_equal(outs[0][qs[8]:], outs[1][qs[8]:])      # the two whole pp512 prompts: bit for bit


@pytest.mark.parametrize("bs,tk", [(64, 1), (192, 1), (192, 8), (256, 2)])
def test_prefill_other_block_sizes(pkv, O, bs, tk):
    """blocks of 64, 192 (not a power of two: the kernel's divide path for the table index) and 256 tokens, chunked prompts over a past"""
    rng = np.random.default_rng(bs + tk)
    lens, n_qs = [1, 63, 190, 193, 500, 700], [1, 63, 100, 193, 37, 300]
    got, ref, exact = run_prefill(pkv, O, rng, lens=lens, n_qs=n_qs, D=128, nh=8, nkv=2, tk=tk, tv=tk, bs=bs)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= (np.abs(ref - exact).max() + 3e-4 if tk == 1 else 1e-3) and nmse(got, ref) <= 5e-4


@pytest.mark.parametrize("D,nh,nkv", [(128, 8, 2), (64, 32, 4), (128, 64, 8)])
@pytest.mark.parametrize("tk,tv", [(1, 1), (8, 8), (2, 8)])
def test_prefill_step_equals_prepare_plus_prefill(pkv, O, tk, tv, D, nh, nkv):
    """pkv_attn_prefill_step (RoPE(K) + KV store, then the attention with RoPE(Q) fused into its Q fetch; q untouched) == pkv_qkv_prepare (q rotated in
    place) + pkv_attn_prefill: same output bits, same pool bits — ragged ubatch over pasts, partial rotation (n_dims < D) included"""
    rng = np.random.default_rng(11 + tk + D + nh)
    bs = 128
    lens, n_qs = [1, 40, 200, 333, 512], [1, 40, 70, 333, 129]
    for n_dims in (D, D // 2):
        kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tv)
        q_start = np.concatenate([[0], np.cumsum(n_qs)]).astype(np.int32)
        n_tok = int(q_start[-1])
        q = rng.uniform(-1, 1, (n_tok, nh, D)).astype(np.float32)
        k = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
        v = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
        # the ubatch's tokens are the LAST n_q cells of each sequence
        cell = np.concatenate([np.arange(n - nq, n) for n, nq in zip(lens, n_qs)])
        seq = np.concatenate([np.full(nq, s) for s, nq in enumerate(n_qs)])
        sb, so = bt[seq, cell // bs].astype(np.int32), (cell % bs).astype(np.int32)
        pos = (cell + rng.integers(0, 3000)).astype(np.int32)
        rp = pkv.rope_params(n_dims=n_dims)
        kw = dict(n_head_kv=nkv, type_k=tk, type_v=tv, block_size=bs, scale=D ** -0.5, max_seq_len=max(lens), max_q_per_seq=max(n_qs))
        pools = [[dev(kp), dev(vp)], [dev(kp), dev(vp)]]
        q1 = dev(q)
        pkv.qkv_prepare(q1, dev(k), dev(v), pools[0][0], pools[0][1], type_k=tk, type_v=tv, slot_block=dev(sb), slot_off=dev(so), block_size=bs, pos=dev(pos), rope=rp)
        o1 = pkv.attn_prefill(q1, pools[0][0], pools[0][1], dev(bt), dev(sl), dev(q_start), **kw)
        q2 = dev(q)
        cache = pkv.rope_cache(dev(pos), rp, D)
        o2 = pkv.attn_prefill_step(q2, dev(k), dev(v), dev(pos), rp, pools[1][0], pools[1][1], dev(bt), dev(sl), dev(q_start), dev(sb), dev(so), rope_cache=cache, **kw)
        torch.cuda.synchronize()
        assert torch.equal(q2, dev(q))                                 # q is not modified
        assert torch.equal(pools[0][0], pools[1][0]) and torch.equal(pools[0][1], pools[1][1])
        assert torch.isfinite(o2).all() and torch.equal(o1, o2), n_dims
    with pytest.raises(pkv.PkvError):                                  # NEOX pairs are not fused
        pkv.attn_prefill_step(q2, dev(k), dev(v), dev(pos), pkv.rope_params(n_dims=D, mode=2), pools[1][0], pools[1][1], dev(bt), dev(sl), dev(q_start), dev(sb), dev(so), rope_cache=cache, **kw)


@pytest.mark.parametrize("tk", [1, 8])
@pytest.mark.parametrize("D,nh,nkv", [(128, 32, 8), (64, 8, 2), (128, 8, 1)])
def test_prefill_f16_q_in_f16_out(pkv, O, tk, D, nh, nkv):
    """PKV_ATTN_Q_F16 | PKV_ATTN_OUT_F16 (the own runner's boundary for prompt ubatches): an f16 Q is what the f32 call makes of the same values
    (for f16 K the reference rounds Q to f16 itself) — the f32 output is bit-identical when Q is f16-representable; the f16 output is that result
    rounded once.  Also through the split-KV mode (a small ubatch over a long past) and the fused step (RoPE(Q) in the Q fetch: the rotation
    starts from the f16 values)."""
    rng = np.random.default_rng(5 + tk + D + nkv)
    for lens, n_qs in (([300, 64, 513, 1], [300, 64, 130, 1]), ([3000], [96])):
        kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tk)
        q_start = np.concatenate([[0], np.cumsum(n_qs)]).astype(np.int32)
        qh = rng.uniform(-1, 1, (int(q_start[-1]), nh, D)).astype(np.float16)
        kw = dict(n_head_kv=nkv, type_k=tk, type_v=tk, block_size=128, scale=D ** -0.5, max_seq_len=int(sl.max()), max_q_per_seq=max(n_qs))
        args = (dev(kp), dev(vp), dev(bt), dev(sl), dev(q_start))
        base = pkv.attn_prefill(dev(qh.astype(np.float32)), *args, **kw).cpu().numpy()                      # f32 in, f32 out
        a = pkv.attn_prefill(dev(qh), *args, **kw)                                                          # f16 in, f32 out
        b = pkv.attn_prefill(dev(qh), *args, out_dtype=torch.float16, **kw)                                 # f16 in, f16 out
        c = pkv.attn_prefill(dev(qh.astype(np.float32)), *args, out_dtype=torch.float16, **kw)              # f32 in, f16 out
        torch.cuda.synchronize()
        assert a.dtype == torch.float32 and b.dtype == torch.float16 and c.dtype == torch.float16
        assert np.array_equal(a.cpu().numpy().view(np.uint32), base.view(np.uint32))
        want = base.astype(np.float16)
        assert np.array_equal(b.cpu().numpy().view(np.uint16), want.view(np.uint16))
        assert np.array_equal(c.cpu().numpy().view(np.uint16), want.view(np.uint16))
    # the fused prefill step on f16 q: same pools, output = the f32 call on the same (f16-representable) q
    lens = [200, 130]
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tk)
    q_start = np.array([0, 200, 330], np.int32)
    n_tok = 330
    qh = rng.uniform(-1, 1, (n_tok, nh, D)).astype(np.float16)
    kc = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
    vc = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
    pos = np.concatenate([np.arange(n) for n in lens]).astype(np.int32)
    sb = np.concatenate([bt[s][np.arange(n) // 128] for s, n in enumerate(lens)]).astype(np.int32)
    so = (pos % 128).astype(np.int32)
    rope = pkv.rope_params(n_dims=D, mode=0, freq_base=10000.0)
    res = []
    for qq, od in ((qh.astype(np.float32), None), (qh, torch.float16)):
        dkp, dvp, dpos = dev(kp), dev(vp), dev(pos)
        cache = pkv.rope_cache(dpos, rope, D)
        o = pkv.attn_prefill_step(dev(qq), dev(kc), dev(vc), dpos, rope, dkp, dvp, dev(bt), dev(sl), dev(q_start), dev(sb), dev(so), rope_cache=cache,
                                  n_head_kv=nkv, type_k=tk, type_v=tk, block_size=128, scale=D ** -0.5, max_seq_len=int(sl.max()), max_q_per_seq=200, out_dtype=od)
        torch.cuda.synchronize()
        res.append((o.cpu().numpy(), dkp.cpu().numpy(), dvp.cpu().numpy()))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[1][0].view(np.uint16), res[0][0].astype(np.float16).view(np.uint16))
    # and the decode path says no
    with pytest.raises(pkv.PkvError):
        pkv.attn_decode(dev(qh[:2]), dev(kp), dev(vp), dev(bt), dev(sl), n_head_kv=nkv, type_k=tk, type_v=tk, block_size=128, scale=1.0, max_seq_len=int(sl.max()))
