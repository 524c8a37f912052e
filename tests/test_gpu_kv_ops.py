"""-m gpu: KV store (+fused RoPE), RoPE, K-shift, reference-row I/O, block copies and the device side of the cache,
all through the C ABI, against the CPU oracle / golden vectors.

Bar: bytes written into the pools are BIT-EXACT with what the reference's ggml_cpy stores (f16 RN, q8_0 AVX2-branch
rounding, q4_0 ref) when the input is identical; RoPE is BIT-EXACT too since r02: the device restates the host libm's
cosf / sinf operation by operation (pkv_rowcodec.h: libm_cosf / libm_sinf; oracle/check_trig.c) and rounds the rotation as the pinned
reference build does (no FMA contraction)."""
import numpy as np
import pytest
import torch

from conftest import cases_of
from gpu_util import dev, nmse

pytestmark = pytest.mark.gpu
TYPES = (1, 8, 2)


def rope_kw(p):
    return dict(n_dims=int(p[0]), mode=int(p[1]), n_ctx_orig=int(p[2]), freq_base=float(p[3]), freq_scale=float(p[4]),
                ext_factor=float(p[5]), attn_factor=float(p[6]), beta_fast=float(p[7]), beta_slow=float(p[8]))


def test_rope_golden(pkv, golden):
    g = golden["rope"]
    for name in cases_of(g):
        kw = rope_kw(g[f"{name}__params"])
        ff = dev(g[f"{name}__ff"]) if f"{name}__ff" in g.files else None
        rp = pkv.rope_params(freq_factors=ff, **kw)
        x = dev(g[f"{name}__x"])
        out = pkv.rope_f32(x, dev(g[f"{name}__pos"]), rp)
        torch.cuda.synchronize()
        ref = g[f"{name}__out"]
        assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32)), name   # the reference's bits
        # in place + inverse gives the input back (times mscale^2 when YaRN / attn_factor scale the rotation)
        pkv.rope_f32(out, dev(g[f"{name}__pos"]), rp, inverse=True, out=out)
        m = kw["attn_factor"] * (1.0 + 0.1 * np.log(1.0 / kw["freq_scale"]) if kw["ext_factor"] != 0 else 1.0)
        back = out.cpu().numpy()
        nd = kw["n_dims"]
        assert np.abs(back[..., :nd] - g[f"{name}__x"][..., :nd] * m * m).max() <= 2e-5, name
        assert np.array_equal(back[..., nd:], g[f"{name}__x"][..., nd:]), name


@pytest.mark.parametrize("tk", TYPES)
@pytest.mark.parametrize("tv", TYPES)
@pytest.mark.parametrize("D", [128, 64])
def test_kv_store_bit_exact(pkv, O, golden, tk, tv, D):
    rng = np.random.default_rng(tk * 10 + tv + D)
    nkv, bs, nb, n_tok = 4, 128, 6, 200
    k = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
    v = (rng.standard_normal((n_tok, nkv, D)) * rng.choice([1e-3, 1.0, 30.0], (n_tok, 1, 1))).astype(np.float32)
    if D == 128:
        k[:48, 0, :] = golden["cpy"]["x"]                    # the reference's corner cases (zeros, ties, scales)
        v[:48, 1, :] = golden["cpy"]["x"]
    k[60, 1, :] = -0.0                                       # blocks of negative zeros: q4_0's scale is max / -8 with max = +0 (the initial value), i.e. -0
    v[61, 2, :32] = -0.0
    # ragged slot mapping: interleaved blocks, one skipped token (slot_block < 0)
    cells = rng.permutation(nb * bs)[:n_tok]
    sb, so = (cells // bs).astype(np.int32), (cells % bs).astype(np.int32)
    sb[7] = -1
    kp_d = torch.zeros(pkv.pool_bytes(tk, nb, nkv, bs, D), dtype=torch.uint8, device="cuda")
    vp_d = torch.zeros(pkv.pool_bytes(tv, nb, nkv, bs, D), dtype=torch.uint8, device="cuda")
    pkv.kv_store(dev(k), dev(v), kp_d, vp_d, type_k=tk, type_v=tv, slot_block=dev(sb), slot_off=dev(so), block_size=bs)
    torch.cuda.synchronize()
    kp, vp = O.pool_alloc(tk, nb, nkv, bs, D), O.pool_alloc(tv, nb, nkv, bs, D)
    O.pool_store(kp, tk, 0, O.from_float(tk, k), sb, so, D=D, block_size=bs)
    O.pool_store(vp, tv, 1, O.from_float(tv, v), sb, so, D=D, block_size=bs)
    assert np.array_equal(kp_d.cpu().numpy(), kp.reshape(-1))
    assert np.array_equal(vp_d.cpu().numpy(), vp.reshape(-1))


@pytest.mark.parametrize("tk", TYPES)
def test_kv_store_fused_rope(pkv, O, tk):
    """RoPE(K) fused into the store == oracle rope followed by the oracle codec, bit for bit (libm-exact device trig, r02)"""
    rng = np.random.default_rng(5 + tk)
    D, nkv, bs, n_tok = 128, 8, 128, 96
    k = rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32)
    pos = rng.integers(0, 8192, n_tok).astype(np.int32)
    ff = rng.uniform(1, 8, 64).astype(np.float32)
    sb, so = np.zeros(n_tok, np.int32), np.arange(n_tok, dtype=np.int32)
    kp_d = torch.zeros(pkv.pool_bytes(tk, 1, nkv, bs, D), dtype=torch.uint8, device="cuda")
    rp = pkv.rope_params(n_dims=128, freq_factors=dev(ff))
    pkv.kv_store(dev(k), None, kp_d, None, type_k=tk, type_v=1, slot_block=dev(sb), slot_off=dev(so), block_size=bs,
                 pos=dev(pos), rope=rp)
    torch.cuda.synchronize()
    got = O.to_float(tk, O.pool_gather(kp_d.cpu().numpy().reshape(1, nkv, -1), tk, 0, np.zeros(1, np.int32), n_tok, D=D,
                                       block_size=bs), D)
    roped = O.rope(k, pos, ff, n_dims=128)
    ref = O.to_float(tk, O.from_float(tk, roped), D).transpose(1, 0, 2)
    assert np.array_equal(got, ref)                          # RoPE + convert: the same stored values as the CPU reference, everywhere


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("is_v", [0, 1])
@pytest.mark.parametrize("D", [128, 64])
def test_rows_io_roundtrip_bit_exact(pkv, O, t, is_v, D):
    """reference-format rows in (set_tensor) -> pool bytes == oracle layout; rows out (get_tensor) == rows in"""
    rng = np.random.default_rng(t + is_v + D)
    nkv, bs, nb, n_tok = 2, 128, 4, 300
    rows = O.from_float(t, rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32))
    cells = rng.permutation(nb * bs)[:n_tok]
    sb, so = (cells // bs).astype(np.int32), (cells % bs).astype(np.int32)
    pool_d = torch.zeros(pkv.pool_bytes(t, nb, nkv, bs, D), dtype=torch.uint8, device="cuda")
    pkv.pool_scatter_rows(pool_d, dev(rows), type=t, is_v=is_v, slot_block=dev(sb), slot_off=dev(so), n_head_kv=nkv,
                          head_dim=D, block_size=bs)
    pool = O.pool_alloc(t, nb, nkv, bs, D)
    O.pool_store(pool, t, is_v, rows, sb, so, D=D, block_size=bs)
    assert np.array_equal(pool_d.cpu().numpy(), pool.reshape(-1))
    back = torch.zeros_like(dev(rows))
    pkv.pool_gather_rows(pool_d, back, type=t, is_v=is_v, slot_block=dev(sb), slot_off=dev(so), n_head_kv=nkv, head_dim=D,
                         block_size=bs)
    assert np.array_equal(back.cpu().numpy(), rows)


@pytest.mark.parametrize("tk", TYPES)
def test_k_shift_in_place(pkv, O, tk):
    """build_rope_shift: cached K rotated by delta in place; f16 directly, quantized via dequantize -> rope -> requantize"""
    rng = np.random.default_rng(21 + tk)
    D, nkv, bs, n = 128, 4, 128, 256
    rows = O.from_float(tk, rng.uniform(-1, 1, (n, nkv, D)).astype(np.float32))
    sb, so = (np.arange(n) // bs).astype(np.int32), (np.arange(n) % bs).astype(np.int32)
    pool = O.pool_alloc(tk, 2, nkv, bs, D)
    O.pool_store(pool, tk, 0, rows, sb, so, D=D, block_size=bs)
    pool_d = dev(pool.reshape(-1))
    cells = rng.permutation(n)[:100]
    delta = rng.integers(-64, 65, 100).astype(np.int32)
    delta[:5] = 0
    rp = pkv.rope_params(n_dims=128)
    pkv.k_shift(pool_d, type_k=tk, cell_block=dev(sb[cells]), cell_off=dev(so[cells]), cell_delta=dev(delta), n_head_kv=nkv,
                head_dim=D, block_size=bs, rope=rp)
    torch.cuda.synchronize()
    got = O.pool_gather(pool_d.cpu().numpy().reshape(pool.shape), tk, 0, np.arange(2, dtype=np.int32), n, D=D, block_size=bs)
    exp_f = O.to_float(tk, rows, D)                           # [n][nkv][D]
    shifted = O.rope(exp_f[cells], delta, n_dims=128)
    exp = rows.copy()
    exp[cells] = O.from_float(tk, shifted)
    nz = cells[delta != 0]
    same = np.ones(n, bool)
    same[nz] = False
    assert np.array_equal(got.transpose(1, 0, 2)[same], rows[same])          # untouched cells (and delta == 0) intact
    g, e = O.to_float(tk, got.transpose(1, 0, 2), D), O.to_float(tk, exp, D)
    assert np.array_equal(g, e)                              # dequantize -> rope(delta) -> requantize: the reference's values exactly


def test_copy_blocks(pkv):
    nkv, bs, D, nb = 8, 128, 128, 6
    pool = torch.randint(0, 255, (nb, pkv.pool_bytes(1, 1, nkv, bs, D)), dtype=torch.uint8, device="cuda")
    before = pool.clone()
    src, dst = np.array([0, 3, 5, -1], np.int32), np.array([1, 4, 5, 2], np.int32)
    pkv.copy_blocks(pool, type=1, src=dev(src), dst=dev(dst), n_head_kv=nkv, head_dim=D, block_size=bs)
    torch.cuda.synchronize()
    assert torch.equal(pool[1], before[0]) and torch.equal(pool[4], before[3])
    for b in (0, 2, 3, 5):
        assert torch.equal(pool[b], before[b])


@pytest.mark.parametrize("tk,tv", [(1, 1), (8, 8), (2, 8)])
def test_cache_device_update_cow_shift_defrag(pkv, O, tk, tv):
    """the whole cache object on the device: prefill, share (seq_cp), diverge (copy-on-write), context shift
    (seq_rm + seq_add -> K-shift), defrag; after every step what each sequence sees equals a dense oracle model."""
    rng = np.random.default_rng(100 + tk)
    D, nkv, nh, bs, L = 128, 2, 4, 128, 2
    rp = pkv.rope_params(n_dims=128)
    c = pkv.PagedKVCache(n_blocks=12, block_size=bs, n_seq_max=3, n_layer=L, n_head_kv=nkv, head_dim=D, type_k=tk, type_v=tv)
    model = {0: [], 1: []}                                    # seq -> list of (pos, k_row[L], v_row[L]) in f32 (post-codec)

    hip = __import__("ctypes").CDLL("libamdhip64.so")

    def pools(layer):
        import ctypes
        torch.cuda.synchronize()

        def grab(ptr, n):                                     # device -> host copy of a library-owned pool
            out = np.empty(n, np.uint8)
            assert hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(n), 2) == 0
            return out
        return (grab(c.k_pool(layer), pkv.pool_bytes(tk, 12, nkv, bs, D)).reshape(12, nkv, -1),
                grab(c.v_pool(layer), pkv.pool_bytes(tv, 12, nkv, bs, D)).reshape(12, nkv, -1))

    def store(seq, poss):
        n = len(poss)
        blk, off = c.find_slot([seq] * n, poss)
        c.update(rp)
        ks = rng.uniform(-1, 1, (L, n, nkv, D)).astype(np.float32)
        vs = rng.uniform(-1, 1, (L, n, nkv, D)).astype(np.float32)
        for l in range(L):
            pkv.kv_store(dev(ks[l]), dev(vs[l]), c.k_pool(l), c.v_pool(l), type_k=tk, type_v=tv, slot_block=dev(blk),
                         slot_off=dev(off), block_size=bs)
        torch.cuda.synchronize()
        kq = np.stack([O.to_float(tk, O.from_float(tk, ks[l]), D) for l in range(L)])
        vq = np.stack([O.to_float(tv, O.from_float(tv, vs[l]), D) for l in range(L)])
        for i, p in enumerate(poss):
            model[seq].append([p, kq[:, i], vq[:, i]])

    def check(tol_k):
        for seq, cells in model.items():
            sp = c.seq_slot_pos(seq)
            vis = [(i, p) for i, p in enumerate(sp) if p >= 0]
            assert sorted(p for _, p in vis) == sorted(m[0] for m in cells), seq
            if not vis:
                continue
            bt = c.seq_blocks(seq)
            by_pos = {}
            for m in cells:
                by_pos.setdefault(m[0], []).append(m)
            for l in range(L):
                kp, vp = pools(l)
                kc = O.to_float(tk, O.pool_gather(kp, tk, 0, bt, len(sp), D=D, block_size=bs), D)
                vc = O.to_float(tv, O.pool_gather(vp, tv, 1, bt, len(sp), D=D, block_size=bs), D)
                for i, p in vis:
                    cand = by_pos[p]
                    ok = any(np.abs(kc[:, i] - m[1][l]).max() <= tol_k and np.array_equal(vc[:, i], m[2][l]) for m in cand)
                    assert ok, (seq, l, i, p)

    store(0, list(range(200)))                                 # prefill: blocks 0,1
    check(0.0)
    c.seq_cp(0, 1, -1, -1)                                     # share
    model[1] = [[m[0], m[1], m[2]] for m in model[0]]
    store(1, [200, 201])                                       # diverge: CoW of the shared tail block
    store(0, [200])
    assert c.block_refcount(int(c.seq_blocks(0)[0])) == 2 and c.block_refcount(int(c.seq_blocks(0)[1])) == 1
    check(0.0)
    # context shift on sequence 0 (examples/main): drop [50, 90), slide the rest down by 40
    c.seq_rm(0, 50, 90)
    c.seq_add(0, 90, -1, -40)
    new = []
    for m in model[0]:
        if 50 <= m[0] < 90:
            continue
        if m[0] >= 90:
            k_sh = np.stack([O.rope(m[1][l][None], np.array([-40], np.int32), n_dims=128)[0] for l in range(L)])
            k_sh = np.stack([O.to_float(tk, O.from_float(tk, k_sh[l]), D) for l in range(L)])
            new.append([m[0] - 40, k_sh, m[2]])
        else:
            new.append(m)
    model[0] = new
    assert c.has_shift()
    c.update(rp)
    assert not c.has_shift()
    check(0.0)                                                 # sequence 1 must be untouched (CoW before the shift); the shifted K rows are the CPU's values exactly
    c.defrag()
    c.update(rp)
    assert c.seq_slot_pos(0).tolist() == list(range(161))
    check(0.0)
    c.close()


@pytest.mark.parametrize("tk,tv", [(1, 1), (8, 8), (2, 2)])
def test_qkv_prepare_equals_separate_ops(pkv, tk, tv):
    """the one-launch RoPE(Q)+RoPE(K)+store must write exactly what rope_f32 + kv_store(rope) write"""
    rng = np.random.default_rng(77)
    D, nh, nkv, bs, n_tok = 128, 32, 8, 128, 5
    q = dev(rng.uniform(-1, 1, (n_tok, nh, D)).astype(np.float32))
    k = dev(rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32))
    v = dev(rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32))
    pos = dev(np.array([0, 17, 4095, 4096, 9000], np.int32))
    sb, so = dev(np.array([1, 0, 1, 0, 1], np.int32)), dev(np.array([5, 9, 127, 0, 64], np.int32))
    rp = pkv.rope_params(n_dims=D)
    pools = [[torch.zeros(pkv.pool_bytes(t, 2, nkv, bs, D), dtype=torch.uint8, device="cuda") for t in (tk, tv)] for _ in range(2)]
    q1 = pkv.rope_f32(q, pos, rp)
    pkv.kv_store(k, v, pools[0][0], pools[0][1], type_k=tk, type_v=tv, slot_block=sb, slot_off=so, block_size=bs, pos=pos, rope=rp)
    q2 = q.clone()
    pkv.qkv_prepare(q2, k, v, pools[1][0], pools[1][1], type_k=tk, type_v=tv, slot_block=sb, slot_off=so, block_size=bs,
                    pos=pos, rope=rp)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2)
    assert torch.equal(pools[0][0], pools[1][0]) and torch.equal(pools[0][1], pools[1][1])


@pytest.mark.parametrize("tk,tv", [(1, 1), (8, 8), (2, 8)])
@pytest.mark.parametrize("D,mode,n_dims", [(128, 0, 128), (128, 2, 128), (64, 0, 64), (64, 2, 64), (128, 0, 64)])
def test_kv_store_per_token_kernel_equals_per_row_kernel(pkv, tk, tv, D, mode, n_dims, monkeypatch):
    """prefill ubatches (>= 64 tokens) take kv_store_tok_kernel (one workgroup per token, a token's RoPE angles evaluated once per wave); it must
    write the bits of the per-row kernel (PKV_KV_STORE_ROWS=1): NORM and NEOX pairs, partial rotation (n_dims < D), every cache type, and the same
    rotated Q; rope_f32 is the third witness for Q"""
    rng = np.random.default_rng(D + mode + n_dims + tk)
    nh, nkv, bs, n_tok = 8, 2, 128, 150
    q = dev(rng.uniform(-1, 1, (n_tok, nh, D)).astype(np.float32))
    k = dev(rng.uniform(-1, 1, (n_tok, nkv, D)).astype(np.float32))
    v = dev((rng.standard_normal((n_tok, nkv, D)) * rng.choice([1e-3, 1.0, 30.0], (n_tok, 1, 1))).astype(np.float32))
    pos = dev(rng.integers(0, 20000, n_tok).astype(np.int32))
    cells = rng.permutation(2 * bs)[:n_tok]
    sb, so = dev((cells // bs).astype(np.int32)), dev((cells % bs).astype(np.int32))
    rp = pkv.rope_params(n_dims=n_dims, mode=mode)
    res = []
    for rows in ("1", None):
        if rows:
            monkeypatch.setenv("PKV_KV_STORE_ROWS", rows)
        else:
            monkeypatch.delenv("PKV_KV_STORE_ROWS", raising=False)
        pools = [torch.zeros(pkv.pool_bytes(t, 2, nkv, bs, D), dtype=torch.uint8, device="cuda") for t in (tk, tv)]
        q2 = q.clone()
        pkv.qkv_prepare(q2, k, v, pools[0], pools[1], type_k=tk, type_v=tv, slot_block=sb, slot_off=so, block_size=bs, pos=pos, rope=rp)
        torch.cuda.synchronize()
        res.append((q2, pools[0], pools[1]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[1][0], pkv.rope_f32(q, pos, rp))
