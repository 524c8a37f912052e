"""-m gpu: the multi-tile form of the paged decode kernel (r03): one workgroup walks several consecutive 128-token tiles with an
online softmax per wave; with one workgroup per (sequence, kv head) the output is written directly (no partials, no merge launch).

The library picks the tiles per workgroup from the launch shape; PKV_DECODE_NTILE (read per launch) forces it here so that every
form meets the same bars on the same ragged inputs: oracle max-abs <= 1e-3 and NMSE <= 5e-4 (tests/test-backend-ops.cpp:3216),
never worse than twice the reference's own distance to f64; the fused step (RoPE + KV store inside the launch) leaves the same
pool bytes whatever the tile count, and physical block order never changes a bit (the table is logical -> physical)."""
import numpy as np
import pytest
import torch

from gpu_util import dev, forced_tiles, make_paged_case, nmse
from test_gpu_decode_step import _case, _fused

pytestmark = pytest.mark.gpu
TYPES = {"f16": 1, "q8_0": 8, "q4_0": 2}


def _decode(pkv, q, kp, vp, bt, sl, nkv, tk, tv):
    D = q.shape[2]
    out = pkv.attn_decode(dev(q), dev(kp), dev(vp), dev(bt), dev(sl), n_head_kv=nkv, type_k=tk, type_v=tv, block_size=128,
                          scale=D ** -0.5, max_seq_len=max(1, int(sl.max())))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("tk,tv", [("f16", "f16"), ("q8_0", "q8_0"), ("q4_0", "q4_0"), ("q8_0", "f16"), ("f16", "q4_0")])
@pytest.mark.parametrize("D,nh,nkv", [(128, 32, 8), (128, 8, 1), (64, 32, 4), (128, 4, 2)])
def test_every_tile_count_matches_oracle(pkv, O, tk, tv, D, nh, nkv):
    tk, tv = TYPES[tk], TYPES[tv]
    rng = np.random.default_rng(11 + D + nh + tk * 3 + tv)
    lens = [0, 1, 2, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 640, 1000, 1024, 1153]   # empty, partial last tiles, 1..10 tiles
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=tk, type_v=tv)
    q = rng.uniform(-1, 1, (len(lens), nh, D)).astype(np.float32)
    ref = np.zeros_like(q)                                   # (the oracle is asked about the non-empty sequences only)
    ref[1:] = O.attn_paged(q[1:, :, None, :], kp, vp, bt[1:], sl[1:], n_q=1, type_k=tk, type_v=tv, block_size=128, scale=D ** -0.5, n_threads=8)[:, 0]
    # the yardstick for long f16-V sequences: the reference accumulates V in f16 (ggml-cpu.c:12374-12388) and drifts from exact arithmetic itself
    exact = np.zeros_like(q)
    for s in range(1, len(lens)):
        kc = O.pool_gather(kp, tk, 0, bt[s], lens[s], D=D, block_size=128)
        vc = O.pool_gather(vp, tv, 1, bt[s], lens[s], D=D, block_size=128)
        exact[s] = O.flash_attn_f64(q[s][:, None, :], kc, vc, None, type_k=tk, type_v=tv, scale=D ** -0.5)[0]
    ref_err = np.abs(ref - exact).reshape(len(lens), -1).max(axis=1)
    outs = {}
    for nt in (1, 2, 3, 4, 16):          # 16 >= the 10 tiles of the longest sequence: one workgroup per (sequence, head), direct output
        with forced_tiles(nt):
            got = _decode(pkv, q, kp, vp, bt, sl, nkv, tk, tv)
        outs[nt] = got
        assert np.isfinite(got).all(), nt
        assert np.array_equal(got[0], np.zeros_like(got[0])), "an empty sequence yields zeros"
        assert nmse(got[1:], ref[1:]) <= 5e-4, nt
        for s in range(1, len(lens)):
            e_ref, e_gpu = np.abs(got[s] - ref[s]).max(), np.abs(got[s] - exact[s]).max()
            assert e_ref <= max(1e-3, ref_err[s] + 3e-4), (nt, s, lens[s], e_ref, ref_err[s])     # 1e-3 against the reference, or as far as IT is from exact
            assert e_gpu <= max(2.0 * ref_err[s], 2e-4), (nt, s, lens[s], e_gpu, ref_err[s])       # never further from exact arithmetic than twice the reference
    # sequences of one tile see the same arithmetic in every form (f16 KV: quantized KV switches between 16 and 32 tokens per wave with the launch size)
    if tk == 1 and tv == 1:
        for nt in (2, 3, 4, 16):
            assert np.array_equal(outs[nt][1:5].view(np.uint32), outs[1][1:5].view(np.uint32)), nt


@pytest.mark.parametrize("tk,tv", [("f16", "f16"), ("q8_0", "q8_0"), ("q4_0", "q4_0")])
@pytest.mark.parametrize("G", [4, 8])
def test_fused_step_pool_bytes_and_outputs(pkv, O, tk, tv, G):
    """the new token is patched into the LAST tile a workgroup walks: same pool bytes as the single-tile form, outputs at the oracle's bars"""
    tk, tv = TYPES[tk], TYPES[tv]
    rng = np.random.default_rng(3 + tk + tv + G)
    D, nkv = 128, 2
    nh = nkv * G
    lens = [1, 2, 128, 129, 256, 257, 385, 512, 513, 700, 1024, 1025]   # new token first / last of a tile, of a V group; 1..9 tiles
    kp, vp, bt, sl, q, kc, vc, pos = _case(O, rng, lens, D, nh, nkv, tk, tv)
    rope = pkv.rope_params(n_dims=D, mode=0, freq_base=10000.0)
    with forced_tiles(1):
        base = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=tk, tv=tv, shared_cache=True)
    for nt in (2, 3, 16):
        with forced_tiles(nt):
            got = _fused(pkv, q, kc, vc, pos, rope, kp, vp, bt, sl, nkv=nkv, tk=tk, tv=tv, shared_cache=True)
        assert np.array_equal(base[1], got[1]), ("K pool bytes differ", nt)
        assert np.array_equal(base[2], got[2]), ("V pool bytes differ", nt)
        assert np.isfinite(got[0]).all()
        assert np.abs(got[0] - base[0]).max() <= 2e-4, nt        # (single-tile form is pinned against the oracle in test_gpu_decode_step.py)
    # and directly against the oracle, on the pools the step left behind
    ref = O.attn_paged(O.rope(q, pos, n_dims=D, freq_base=10000.0)[:, :, None, :], base[1], base[2], bt, sl, n_q=1, type_k=tk, type_v=tv, block_size=128, scale=D ** -0.5)[:, 0]
    assert np.abs(got[0] - ref).max() <= 1e-3 and nmse(got[0], ref) <= 5e-4


def test_block_permutation_is_bit_exact_in_every_form(pkv, O):
    rng = np.random.default_rng(9)
    D, nh, nkv = 128, 32, 8
    lens = [1153, 640, 1024, 77]
    kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=1, type_v=1)
    q = rng.uniform(-1, 1, (len(lens), nh, D)).astype(np.float32)
    nb = kp.shape[0]
    perm = rng.permutation(nb).astype(np.int32)               # new physical place of every block
    kp2, vp2 = np.empty_like(kp), np.empty_like(vp)
    kp2[perm], vp2[perm] = kp, vp
    bt2 = perm[bt]
    for nt in (None, 1, 3, 16):
        with forced_tiles(nt):
            a = _decode(pkv, q, kp, vp, bt, sl, nkv, 1, 1)
            b = _decode(pkv, q, kp2, vp2, bt2, sl, nkv, 1, 1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), nt


def test_library_choice_many_short_and_long(pkv, O):
    """the shapes the heuristic exists for, at reduced size: 40 sequences x <= 5 tiles (one workgroup per stream, no merge launch) and one
    sequence of 128 tiles at gqa 8 (several tiles per workgroup + merge)"""
    rng = np.random.default_rng(21)
    for lens, nh, nkv in (([int(x) for x in rng.integers(1, 640, 38)] + [640, 513], 32, 8), ([16384 - 5], 64, 8)):
        D = 128
        kp, vp, bt, sl = make_paged_case(O, rng, lens=lens, D=D, n_head=nh, n_head_kv=nkv, type_k=1, type_v=1)
        q = rng.uniform(-1, 1, (len(lens), nh, D)).astype(np.float32)
        with forced_tiles(None):
            got = _decode(pkv, q, kp, vp, bt, sl, nkv, 1, 1)
        ref = O.attn_paged(q[:, :, None, :], kp, vp, bt, sl, n_q=1, type_k=1, type_v=1, block_size=128, scale=D ** -0.5, n_threads=8)[:, 0]
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= 1e-3 and nmse(got, ref) <= 5e-4
