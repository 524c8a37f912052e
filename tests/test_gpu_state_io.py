"""-m gpu: session state with K/V rows: gather-on-save / scatter-on-load through the block table, in the
reference's byte format (/root/reference/src/llama-kv-cache.cpp:848-1221).  Checker = oracle/kv_cells_ref.py's restatement
of the reference's flat cell ring ("parity unpinned", see tests/test_state_io.py).  Bar: byte-exact rows."""
import numpy as np
import pytest
import torch

from gpu_util import dev
from oracle.kv_cells_ref import UnifiedCells, state_read, state_write

pytestmark = pytest.mark.gpu
L, HK, D, BS = 2, 2, 128, 128


def _rows(O, rng, t, n):
    return O.from_float(t, rng.uniform(-1, 1, (n, HK, D)).astype(np.float32)).reshape(n, -1)    # [n][HK * row_size(D)]


def _store(pkv, c, tk, tv, blk, off, k_rows, v_rows):
    sb, so = dev(blk), dev(off)
    for il in range(L):
        for pool, t, is_v, rows in ((c.k_pool(il), tk, 0, k_rows[il]), (c.v_pool(il), tv, 1, v_rows[il])):
            pkv.pool_scatter_rows(pool, dev(rows), type=t, is_v=is_v, slot_block=sb, slot_off=so, n_head_kv=HK, head_dim=D, block_size=BS)
    torch.cuda.synchronize()


def _load(pkv, c, t, is_v, il, blk, off, row_bytes):
    out = torch.empty((len(blk), row_bytes), dtype=torch.uint8, device="cuda")
    pool = c.v_pool(il) if is_v else c.k_pool(il)
    pkv.pool_gather_rows(pool, out, type=t, is_v=is_v, slot_block=dev(blk), slot_off=dev(off), n_head_kv=HK, head_dim=D, block_size=BS)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _cells_of(c, s):
    """(block, off) of every logical cell of sequence s"""
    n = c.seq_len(s)
    b = c.seq_blocks(s)
    i = np.arange(n)
    return b[i // BS].astype(np.int32), (i % BS).astype(np.int32)


@pytest.mark.parametrize("tk,tv", [(1, 1), (8, 8), (2, 8)])
def test_state_round_trip_and_reference_view(pkv, O, tk, tv):
    rng = np.random.default_rng(40 + tk + tv)
    c = pkv.PagedKVCache(n_blocks=8, block_size=BS, n_seq_max=3, n_layer=L, n_head_kv=HK, head_dim=D, type_k=tk, type_v=tv)
    lens = {0: 200, 2: 131}
    stored = {}
    for s, n in lens.items():
        blk, off = c.find_slot([s] * n, list(range(n)))
        k_rows = [_rows(O, rng, tk, n) for _ in range(L)]
        v_rows = [_rows(O, rng, tv, n) for _ in range(L)]
        _store(pkv, c, tk, tv, blk, off, k_rows, v_rows)
        stored[s] = (k_rows, v_rows)
    blob = c.state_write(-1)
    assert len(blob) == pkv.lib().pkv_cache_state_size(c.h, -1)
    # (1) the reference ring reads the file: each sequence sees its rows, in order
    size = 8 * BS
    kr, vr = k_rows[0].shape[1], v_rows[0].shape[1]
    uc = UnifiedCells(size)
    fk = [np.zeros((size, kr), np.uint8) for _ in range(L)]
    fv = [np.zeros((size, vr), np.uint8) for _ in range(L)]
    assert state_read(uc, fk, fv, tk, tv, blob, -1)
    for s, n in lens.items():
        idx = [i for i, cell in enumerate(uc.cells) if s in cell.seq]
        assert [uc.cells[i].pos for i in idx] == list(range(n))
        for il in range(L):
            assert np.array_equal(fk[il][idx], stored[s][0][il]) and np.array_equal(fv[il][idx], stored[s][1][il])
    # (2) a fresh paged cache restores it: same rows through its own block tables, same attention result
    c2 = pkv.PagedKVCache(n_blocks=6, block_size=BS, n_seq_max=3, n_layer=L, n_head_kv=HK, head_dim=D, type_k=tk, type_v=tv)
    c2.find_slot([1] * 5, list(range(5)))
    c2.state_read(blob, -1)
    assert c2.seq_len(1) == 0
    q = dev(rng.uniform(-1, 1, (2, 4 * HK, D)).astype(np.float32))
    outs = []
    for cc in (c, c2):
        for s, n in lens.items():
            b, o = _cells_of(cc, s)
            for il in range(L):
                assert np.array_equal(_load(pkv, cc, tk, 0, il, b, o, kr), stored[s][0][il])
                assert np.array_equal(_load(pkv, cc, tv, 1, il, b, o, vr), stored[s][1][il])
        bt, sl = cc.block_table([0, 2])
        out = pkv.attn_decode(q, cc.k_pool(1), cc.v_pool(1), dev(bt), dev(sl), n_head_kv=HK, type_k=tk, type_v=tv, block_size=BS,
                              scale=D ** -0.5, max_seq_len=int(sl.max()))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    # (3) one sequence moved to another id of another cache
    c2.state_read(c.state_write(2), 1)
    b, o = _cells_of(c2, 1)
    assert len(b) == 131 and np.array_equal(_load(pkv, c2, tk, 0, 0, b, o, kr), stored[2][0][0])
    assert np.array_equal(_load(pkv, c2, tv, 1, 1, b, o, vr), stored[2][1][1])


@pytest.mark.parametrize("v_trans", [False, True])
def test_reference_written_file_with_interleaved_cells(pkv, O, v_trans):
    """a file as the reference's ring writes it after interleaved decoding (+ transposed f16 V of `-fa 0` runs)"""
    rng = np.random.default_rng(3 + v_trans)
    size = 4 * BS
    uc = UnifiedCells(size)
    uc.find_slot([0] * 130, list(range(130)))                    # a prompt longer than one block, shared by seq 1
    uc.advance_head(130)
    uc.seq_cp(0, 1, -1, -1)
    for step in range(20):
        uc.find_slot([0, 1], [130 + step, 130 + step])
        uc.advance_head(2)
    n_used = 170
    fk = [np.zeros((size, HK * D * 2), np.uint8) for _ in range(L)]
    fv = [np.zeros((size, HK * D * 2), np.uint8) for _ in range(L)]
    for il in range(L):
        fk[il][:n_used] = _rows(O, rng, 1, n_used)
        fv[il][:n_used] = _rows(O, rng, 1, n_used)
    blob = state_write(uc, fk, fv, 1, 1, -1, v_trans=v_trans)
    c = pkv.PagedKVCache(n_blocks=6, block_size=BS, n_seq_max=2, n_layer=L, n_head_kv=HK, head_dim=D, type_k=1, type_v=1)
    c.state_read(blob, -1)
    for s in (0, 1):
        idx = [i for i, cell in enumerate(uc.cells) if s in cell.seq]
        assert list(c.seq_slot_pos(s)) == [uc.cells[i].pos for i in idx]
        b, o = _cells_of(c, s)
        for il in range(L):
            assert np.array_equal(_load(pkv, c, 1, 0, il, b, o, HK * D * 2), fk[il][idx])
            assert np.array_equal(_load(pkv, c, 1, 1, il, b, o, HK * D * 2), fv[il][idx])
    assert c.seq_blocks(0)[0] == c.seq_blocks(1)[0] and c.block_refcount(c.seq_blocks(0)[0]) == 2   # the full prompt block is shared again


def test_reference_written_golden_files(pkv, O):
    """tests/golden/state_io.npz: session files produced by the REFERENCE's own state_write (compiled llama_kv_cache_unified,
    tests/golden/make_state_golden.py).  The paged cache must restore exactly the rows each sequence had in the ring — all KV
    types, and the transposed-V (`-fa 0`) file — and what it writes back must read as the same views."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "state_io.npz"))
    for name in g["names"]:
        tk, tv, v_trans, n_layer, hk, d, size = (int(x) for x in g[f"{name}__meta"])
        pos, nseq, seqs = g[f"{name}__cell_pos"], g[f"{name}__cell_nseq"], g[f"{name}__cell_seqs"]
        ring = {s: [i for i in range(size) if s in seqs[i][:nseq[i]]] for s in (0, 1)}
        c = pkv.PagedKVCache(n_blocks=4, block_size=BS, n_seq_max=2, n_layer=n_layer, n_head_kv=hk, head_dim=d, type_k=tk, type_v=tv)

        def check(cc, s, idx):
            n = cc.seq_len(s)
            assert list(cc.seq_slot_pos(s)) == [int(pos[i]) for i in idx], (name, s)
            b = cc.seq_blocks(s)
            i = np.arange(n)
            blk, off = dev(b[i // BS].astype(np.int32)), dev((i % BS).astype(np.int32))
            for il in range(n_layer):
                for is_v, t, key in ((0, tk, "k"), (1, tv, "v")):
                    exp = g[f"{name}__{key}{il}"][idx]
                    out = torch.empty(exp.shape, dtype=torch.uint8, device="cuda")
                    pool = cc.v_pool(il) if is_v else cc.k_pool(il)
                    pkv.pool_gather_rows(pool, out, type=t, is_v=is_v, slot_block=blk, slot_off=off, n_head_kv=hk, head_dim=d, block_size=BS)
                    torch.cuda.synchronize()
                    assert np.array_equal(out.cpu().numpy(), exp), (name, s, il, key)

        c.state_read(g[f"{name}__blob"].tobytes(), -1)
        for s in (0, 1):
            check(c, s, ring[s])
        # the single-sequence file, restored under another id of a fresh cache
        c1 = pkv.PagedKVCache(n_blocks=4, block_size=BS, n_seq_max=2, n_layer=n_layer, n_head_kv=hk, head_dim=d, type_k=tk, type_v=tv)
        c1.state_read(g[f"{name}__blob_seq1"].tobytes(), 0)
        check(c1, 0, ring[1])
        # what the paged cache writes, the (reference-pinned) ring restatement reads back to the same per-sequence rows
        # (sharing is per block on the paged side, so the removed range of seq 1 costs private copies: a larger ring)
        blob2 = c.state_write(-1)
        uc = UnifiedCells(2 * size)
        k2 = [np.zeros((2 * size, g[f"{name}__k{il}"].shape[1]), np.uint8) for il in range(n_layer)]
        v2 = [np.zeros((2 * size, g[f"{name}__v{il}"].shape[1]), np.uint8) for il in range(n_layer)]
        assert state_read(uc, k2, v2, tk, tv, blob2, -1)
        for s in (0, 1):
            idx2 = [i for i, cell in enumerate(uc.cells) if s in cell.seq]
            assert [uc.cells[i].pos for i in idx2] == [int(pos[i]) for i in ring[s]]
            for il in range(n_layer):
                assert np.array_equal(k2[il][idx2], g[f"{name}__k{il}"][ring[s]]) and np.array_equal(v2[il][idx2], g[f"{name}__v{il}"][ring[s]])


def test_state_read_rejects_mismatched_geometry(pkv, O):
    a = pkv.PagedKVCache(n_blocks=2, block_size=BS, n_seq_max=1, n_layer=L, n_head_kv=HK, head_dim=D, type_k=1, type_v=1)
    a.find_slot([0] * 3, [0, 1, 2])
    blob = a.state_write(-1)
    b = pkv.PagedKVCache(n_blocks=2, block_size=BS, n_seq_max=1, n_layer=L, n_head_kv=HK, head_dim=D, type_k=8, type_v=1)
    b.find_slot([0] * 2, [0, 1])
    with pytest.raises(pkv.PkvError, match="mismatched key type"):
        b.state_read(blob, -1)
    assert b.get_n_tokens() == 0                                 # cleared, like the reference after a failed restore
