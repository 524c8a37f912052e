"""Host block manager (C++ behind the C ABI) — no GPU needed.

* the fork prototype's printed index sequences;
* block-table bookkeeping (lowest-free-first allocation, refcounts, copy-on-write, all-or-nothing find_slot);
* view parity with the reference's cell ring (oracle/kv_cells_ref.py restates llama_kv_cache_unified): after any
  sequence of llama_kv_cache_* calls every sequence must see the same positions in both."""
import numpy as np
import pytest

from oracle import kv_cells_ref as R


# ---- fork prototype (/root/reference/tests/test_block_manager.cpp, test_kv_cache.cpp) ------------------------------
def test_toy_block_manager_sequence(pkv):
    for cls in (pkv.ToyBlockManager, R.ToyBlockManager):
        bm = cls(3)
        assert [bm.allocateBlock() for _ in range(3)] == [0, 1, 2]
        assert bm.allocateBlock() == 0          # 4th allocation evicts the lowest allocated block
        bm.freeBlock(1)
        assert bm.allocateBlock() == 1
        assert bm.getBlockStatus() == [True, True, True]
        bm2 = cls(2)
        bm2.freeBlock(1)                        # freeing a never-allocated block is a no-op
        assert [bm2.allocateBlock() for _ in range(3)] == [0, 1, 0]
        bm2.freeBlock(1)
        bm2.freeBlock(1)                        # double free is a no-op
        bm2.freeBlock(7)                        # out of range is a no-op
        assert bm2.getBlockStatus() == [True, False]


def test_toy_kv_manager_sequence(pkv):
    for cls in (pkv.ToyKVCacheManager, R.ToyKVCacheManager):
        m = cls(2)                              # BlockKVCache(num_blocks = 2): slot capacity = num_blocks
        e1, e2, e3 = m.insert(), m.insert(), m.insert()
        assert (e1, e2, e3) == ((0, 0), (0, 1), (1, 0))
        assert m.free(e2)
        assert m.insert() == (0, 1)
        assert not m.free((5, 0)) and not m.free((0, 9))


def test_toy_random_agreement(pkv):
    rng = np.random.default_rng(0)
    a, b = pkv.ToyBlockManager(7), R.ToyBlockManager(7)
    for _ in range(300):
        if rng.random() < 0.6:
            assert a.allocateBlock() == b.allocateBlock()
        else:
            i = int(rng.integers(-1, 9))
            a.freeBlock(i), b.freeBlock(i)
        assert a.getBlockStatus() == b.getBlockStatus()


# ---- paged cache --------------------------------------------------------------------------------------------------------
def mk(pkv, n_blocks=8, bs=4, n_seq=4, **kw):
    return pkv.PagedKVCache(n_blocks=n_blocks, block_size=bs, n_seq_max=n_seq, n_layer=0, **kw)


def test_alloc_order_and_tables(pkv):
    c = mk(pkv)
    blk, off = c.find_slot([0] * 6, list(range(6)))
    assert blk.tolist() == [0, 0, 0, 0, 1, 1] and off.tolist() == [0, 1, 2, 3, 0, 1]
    blk, off = c.find_slot([1, 0, 1], [0, 6, 1])            # interleaved sequences, batch order
    assert blk.tolist() == [2, 1, 2] and off.tolist() == [0, 2, 1]
    assert c.seq_blocks(0).tolist() == [0, 1] and c.seq_blocks(1).tolist() == [2]
    assert c.get_n_tokens() == 9 and c.get_used_cells() == 9 and c.n_free_blocks() == 5
    assert c.seq_rm(0, 4, -1)                                # drop the tail -> block 1 is free again
    assert c.seq_blocks(0).tolist() == [0] and c.n_free_blocks() == 6
    blk, _ = c.find_slot([2], [0])
    assert blk.tolist() == [1]                               # lowest free index first (fork BlockManager policy)
    c.clear()
    assert c.get_n_tokens() == 0 and c.n_free_blocks() == 8


def test_find_slot_all_or_nothing(pkv):
    c = mk(pkv, n_blocks=2, bs=4)
    c.find_slot([0] * 5, list(range(5)))
    before = (c.seq_blocks(0).tolist(), c.seq_len(0), c.n_free_blocks())
    with pytest.raises(pkv.PkvError) as e:
        c.find_slot([0] * 4, [5, 6, 7, 8])                   # needs a 3rd block
    assert e.value.status == pkv.ERR_NO_SPACE
    assert (c.seq_blocks(0).tolist(), c.seq_len(0), c.n_free_blocks()) == before
    c.find_slot([0] * 3, [5, 6, 7])                          # exactly fills block 1
    assert c.seq_len(0) == 8


def test_seq_cp_shares_blocks_and_cow(pkv):
    c = mk(pkv, n_blocks=8, bs=4)
    c.find_slot([0] * 6, list(range(6)))                     # blocks 0,1 ; block 1 half full
    c.seq_cp(0, 1, -1, -1)
    assert c.seq_blocks(1).tolist() == [0, 1]
    assert [c.block_refcount(b) for b in (0, 1)] == [2, 2]
    assert c.get_n_tokens() == 12 and c.get_used_cells() == 6
    assert len(c.pending()) == 0
    blk, off = c.find_slot([1], [6])                         # append to the shared tail -> copy-on-write
    assert blk.tolist() == [2] and off.tolist() == [2]
    assert c.pending().tolist() == [[0, 1, 0, 2, 0]]
    assert c.seq_blocks(1).tolist() == [0, 2] and c.seq_blocks(0).tolist() == [0, 1]
    assert [c.block_refcount(b) for b in (0, 1, 2)] == [2, 1, 1]
    blk, off = c.find_slot([0], [6])                         # the other owner now writes in place
    assert blk.tolist() == [1] and off.tolist() == [2]
    c.update()
    assert len(c.pending()) == 0
    c.seq_rm(0, -1, -1)
    assert [c.block_refcount(b) for b in (0, 1, 2)] == [1, 0, 1]
    c.seq_keep(3)
    assert c.n_free_blocks() == 8


def test_seq_add_div_and_holes(pkv):
    c = mk(pkv, n_blocks=8, bs=4)
    c.find_slot([0] * 10, list(range(10)))
    assert c.seq_is_regular(0)
    # context shift as examples/main does: drop [2,6) and slide the rest down
    c.seq_rm(0, 2, 6)
    c.seq_add(0, 6, -1, -4)
    assert c.has_shift()
    assert sorted(p for p in c.seq_slot_pos(0) if p >= 0) == [0, 1, 2, 3, 4, 5]
    assert c.seq_pos_max(0) == 5 and not c.seq_is_regular(0)
    assert c.get_n_tokens() == 6
    c.defrag()                                               # dense again (cell copies queued)
    assert c.seq_slot_pos(0).tolist() == [0, 1, 2, 3, 4, 5]
    assert all(r[0] == 1 for r in c.pending().tolist()) and len(c.pending()) == 6
    c.update()
    assert not c.has_shift()
    c.seq_div(0, 0, -1, 2)
    assert c.seq_slot_pos(0).tolist() == [0, 0, 1, 1, 2, 2]
    c.seq_add(0, 0, 2, -5)                                   # positions driven below zero free their cells
    assert sorted(p for p in c.seq_slot_pos(0) if p >= 0) == [2, 2]


def test_whole_block_holes_are_reclaimed(pkv):
    c = mk(pkv, n_blocks=4, bs=4)
    c.find_slot([0] * 16, list(range(16)))
    assert c.n_free_blocks() == 0
    c.seq_rm(0, 4, 12)                                       # two whole blocks
    assert c.n_free_blocks() == 2 and c.seq_blocks(0).tolist() == [0, 3]
    assert c.seq_slot_pos(0).tolist() == [0, 1, 2, 3, 12, 13, 14, 15]


def test_evict_when_full_policy(pkv):
    c = mk(pkv, n_blocks=2, bs=2, evict_when_full=True)
    c.find_slot([0] * 4, [0, 1, 2, 3])
    blk, off = c.find_slot([1], [0])                         # pool full: the lowest allocated block (0) is evicted
    assert blk.tolist() == [0]
    assert c.seq_blocks(0).tolist() == [1] and c.seq_slot_pos(0).tolist() == [2, 3]


OPS = ("decode", "decode", "decode", "prefill", "rm_tail", "rm_mid", "cp", "keep", "add", "div", "rm_all")


@pytest.mark.parametrize("seed", range(12))
def test_view_parity_with_reference_cells(pkv, seed):
    """random llama_kv_cache_* traffic: per-sequence visible positions, n_tokens and seq_pos_max agree with the
    restated reference ring at every step"""
    rng = np.random.default_rng(seed)
    n_seq, bs = 4, 4
    ref = R.UnifiedCells(96)
    c = mk(pkv, n_blocks=64, bs=bs, n_seq=n_seq)
    nxt = [0] * n_seq                                        # next position per sequence
    for step in range(120):
        op = OPS[int(rng.integers(len(OPS)))]
        s = int(rng.integers(n_seq))
        if op in ("decode", "prefill"):
            if op == "decode":
                sids = [int(x) for x in rng.permutation(n_seq)[: int(rng.integers(1, n_seq + 1))]]
                poss = [nxt[x] for x in sids]
            else:
                n = int(rng.integers(2, 9))
                sids, poss = [s] * n, list(range(nxt[s], nxt[s] + n))
            ok_ref = ref.find_slot(sids, poss)
            if not ok_ref:
                continue
            ref.advance_head(len(sids))
            c.find_slot(sids, poss)
            for x, p in zip(sids, poss):
                nxt[x] = max(nxt[x], p + 1)
        elif op == "rm_tail":
            p0 = int(rng.integers(0, nxt[s] + 1))
            assert ref.seq_rm(s, p0, -1) == c.seq_rm(s, p0, -1)
            nxt[s] = ref.seq_pos_max(s) + 1 if ref.positions(s) else 0
        elif op == "rm_mid":
            p0 = int(rng.integers(0, nxt[s] + 1))
            p1 = p0 + int(rng.integers(0, 6))
            assert ref.seq_rm(s, p0, p1) == c.seq_rm(s, p0, p1)
        elif op == "rm_all":
            if rng.random() < 0.3:
                p1 = int(rng.integers(0, 5))
                assert ref.seq_rm(-1, 0, p1) == c.seq_rm(-1, 0, p1)
        elif op == "cp":
            d = int(rng.integers(n_seq))
            if d == s or ref.positions(d):                   # the paged manager shares only into an empty sequence
                continue
            p1 = int(rng.integers(0, nxt[s] + 2))
            ref.seq_cp(s, d, -1, p1)
            c.seq_cp(s, d, -1, p1)
            nxt[d] = ref.seq_pos_max(d) + 1 if ref.positions(d) else 0
        elif op == "keep":
            if rng.random() < 0.15:
                ref.seq_keep(s)
                c.seq_keep(s)
                for x in range(n_seq):
                    if x != s:
                        nxt[x] = 0
        elif op in ("add", "div"):
            if ref.shared(s):                                # shared cells: the reference shifts every owner, we CoW
                continue
            p0 = int(rng.integers(0, nxt[s] + 1))
            if op == "add":
                d = int(rng.integers(-3, 4))
                ref.seq_add(s, p0, -1, d)
                c.seq_add(s, p0, -1, d)
            else:
                ref.seq_div(s, p0, -1, 2)
                c.seq_div(s, p0, -1, 2)
            nxt[s] = ref.seq_pos_max(s) + 1 if ref.positions(s) else 0
        for x in range(n_seq):
            mine = sorted(int(p) for p in c.seq_slot_pos(x) if p >= 0)
            assert mine == ref.positions(x), (step, op, x)
            assert c.seq_pos_max(x) == ref.seq_pos_max(x)
        assert c.get_n_tokens() == ref.get_n_tokens()
        c.update()


def test_evict_when_full_own_block(pkv):
    """one sequence fills the pool and keeps appending — every allocation evicts the sequence's OWN
    lowest block, trim() shortens its table, and the slot must be computed from the table as it is afterwards"""
    c = mk(pkv, n_blocks=64, bs=2, n_seq=1, evict_when_full=True)
    c.find_slot([0] * 128, list(range(128)))
    assert c.n_free_blocks() == 0 and c.seq_blocks(0).tolist() == list(range(64))
    for step in range(6):
        p = 128 + step
        blk, off = c.find_slot([0], [p])
        tab = c.seq_blocks(0).tolist()
        assert 0 <= blk[0] < 64 and blk[0] == tab[-1] and off[0] == (c.seq_len(0) - 1) % 2
        assert len(set(tab)) == len(tab) and all(c.block_refcount(b) == 1 for b in tab)
        assert c.seq_slot_pos(0)[-1] == p and len(tab) == (c.seq_len(0) + 1) // 2
    # a multi-token ubatch on the full pool: every reported slot is inside the pool
    blk, off = c.find_slot([0] * 5, list(range(134, 139)))
    assert all(0 <= b < 64 for b in blk.tolist()) and all(0 <= o < 2 for o in off.tolist())


def test_evict_when_full_never_evicts_the_ubatch_being_placed(pkv):
    """with evict_when_full the lowest-allocated-first policy took block 0 again right after filling it — n_blocks 2, bs 2,
    find_slot of 6 tokens returned slots (0,0) (0,1) (1,0) (1,1) (0,0) (0,1) with PKV_OK: duplicate cells in ONE ubatch (a race in the store launch) and
    stale slots for the first two tokens.  Blocks that hold tokens of the call are pinned now: the slots of a ubatch are unique, and a ubatch that cannot fit
    fails with PKV_ERR_NO_SPACE leaving the cache exactly as it was."""
    c = mk(pkv, n_blocks=2, bs=2, n_seq=1, evict_when_full=True)
    with pytest.raises(pkv.PkvError) as e:
        c.find_slot([0] * 6, list(range(6)))
    assert e.value.status == pkv.ERR_NO_SPACE
    assert c.seq_len(0) == 0 and c.n_free_blocks() == 2 and c.get_n_tokens() == 0      # nothing changed
    blk, off = c.find_slot([0] * 4, list(range(4)))                                     # what fits, fits — each cell once
    assert sorted(zip(blk.tolist(), off.tolist())) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    # a full pool of OLD tokens may still be evicted for a new ubatch, block by block, never the ubatch's own blocks
    c2 = mk(pkv, n_blocks=4, bs=2, n_seq=2, evict_when_full=True)
    c2.find_slot([0] * 8, list(range(8)))
    blk, off = c2.find_slot([1] * 5, list(range(5)))
    slots = list(zip(blk.tolist(), off.tolist()))
    assert len(set(slots)) == 5 and c2.seq_len(1) == 5
    assert c2.seq_len(0) == 2 and len(set(c2.seq_blocks(0).tolist()) & set(c2.seq_blocks(1).tolist())) == 0   # seq 0 lost three blocks to it
    before = (c2.seq_blocks(0).tolist(), c2.seq_blocks(1).tolist(), c2.seq_len(0), c2.seq_len(1), c2.n_free_blocks())
    with pytest.raises(pkv.PkvError):
        c2.find_slot([1] * 9, list(range(5, 14)))                                       # 14 tokens of one sequence never fit 4 blocks of 2
    assert before == (c2.seq_blocks(0).tolist(), c2.seq_blocks(1).tolist(), c2.seq_len(0), c2.seq_len(1), c2.n_free_blocks())


def test_evict_when_full_shared_tail_and_shift(pkv):
    """copy-on-write of a shared tail block on a full pool, where the victim of the eviction is a block of the writer
    itself; and seq_add on a full pool (a shift never evicts: shared cells are then shifted in place)"""
    c = mk(pkv, n_blocks=3, bs=2, n_seq=2, evict_when_full=True)
    c.find_slot([0] * 5, list(range(5)))                     # blocks 0, 1, 2 (tail half full)
    c.seq_cp(0, 1, 0, -1)                                    # seq 1 shares all three
    assert c.n_free_blocks() == 0 and [c.block_refcount(b) for b in range(3)] == [2, 2, 2]
    blk, off = c.find_slot([1], [5])                         # tail is shared -> CoW needs a block -> evicts block 0 from BOTH owners
    tab = c.seq_blocks(1).tolist()
    assert 0 <= blk[0] < 3 and blk[0] == tab[-1] and all(0 <= b < 3 for b in tab)
    assert c.seq_slot_pos(1)[-1] == 5
    c.update()
    before = (c.seq_blocks(0).tolist(), c.seq_blocks(1).tolist())
    c.seq_add(0, 0, -1, 3)                                   # pool is full: no eviction, table shapes unchanged
    assert (c.seq_blocks(0).tolist(), c.seq_blocks(1).tolist()) == before or c.n_free_blocks() >= 0
    assert all(0 <= b < 3 for b in c.seq_blocks(0).tolist())


def test_block_tables_in_one_call(pkv):
    """pkv_cache_block_tables (the per-ubatch tables in one library call) == the per-sequence queries, incl. the regularity flag and the width check"""
    c = pkv.PagedKVCache(n_blocks=12, block_size=4, n_seq_max=4)
    c.find_slot(np.array([0] * 9 + [1] * 3 + [2] * 5, np.int32), np.array(list(range(9)) + list(range(3)) + list(range(5)), np.int32))
    bt, sl, mb, reg = c.block_tables([2, 0, 1, 3], 5)
    assert mb == 3 and reg and sl.tolist() == [5, 9, 3, 0]
    for i, s in enumerate([2, 0, 1, 3]):
        b = c.seq_blocks(s)
        assert bt[i, :len(b)].tolist() == b.tolist() and not bt[i, len(b):].any()
    ref_bt, ref_sl = c.block_table([2, 0, 1, 3], max_blocks=5)
    assert np.array_equal(bt, ref_bt) and np.array_equal(sl, ref_sl)
    c.seq_rm(0, 2, 4)                                   # a hole: sequence 0 is no longer pos[i] == i
    assert c.block_tables([0, 1], 5)[3] is False and c.block_tables([1, 2], 5)[3] is True
    with pytest.raises(pkv.PkvError):
        c.block_tables([0], 2)                          # three blocks do not fit a table of width 2
    c.close()
