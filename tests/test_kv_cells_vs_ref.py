"""Pins oracle/kv_cells_ref.py (the Python restatement of the reference's KV cell ring that every block-manager and
session-state test checks against) to the REFERENCE ITSELF: llama_kv_cache_unified compiled in place from
/root/reference/src (oracle/_ref/libref_kv.so, driver oracle/ref_kv_driver.cpp).  Runs where that build exists (this
container); elsewhere the committed fixtures of tests/golden/make_state_golden.py carry the same evidence."""
import numpy as np
import pytest

from oracle import ref_kv
from oracle.kv_cells_ref import UnifiedCells, state_read, state_write

pytestmark = pytest.mark.skipif(not ref_kv.available(), reason="oracle/_ref/libref_kv.so not built (needs /root/reference)")

L, HK, D = 2, 2, 64
ROW = {1: HK * D * 2, 8: HK * D // 32 * 34, 2: HK * D // 32 * 18}


def _same_cells(r, uc):
    rc = r.cells()
    for i, c in enumerate(uc.cells):
        # the reference leaves a stale `pos` in cells it empties through seq_keep/seq_rm paths only when it says so: compare
        # what is observable — emptiness, and pos/delta/seq set of occupied cells
        assert sorted(c.seq) == rc[i][2], (i, sorted(c.seq), rc[i])
        if c.seq:
            assert (c.pos, c.delta) == rc[i][:2], (i, c.pos, c.delta, rc[i])
        else:
            assert rc[i][0] == c.pos, (i, c.pos, rc[i])


@pytest.mark.parametrize("seed", range(6))
def test_random_edit_sequences_match_the_reference(seed):
    rng = np.random.default_rng(seed)
    size, n_seq = 48, 4
    r = ref_kv.RefKvCache(n_layer=1, n_head_kv=HK, head_dim=D, type_k=1, type_v=1, kv_size=size)
    uc = UnifiedCells(size)
    nxt = [0] * n_seq
    for step in range(120):
        op = rng.choice(["decode", "decode", "prompt", "rm", "cp", "keep", "add", "div", "clear"], p=[.3, .2, .15, .1, .08, .02, .07, .05, .03])
        if op in ("decode", "prompt"):
            if op == "decode":
                sids = [int(s) for s in rng.permutation(n_seq)[:rng.integers(1, n_seq + 1)]]
            else:
                sids = [int(rng.integers(n_seq))] * int(rng.integers(2, 9))
            poss = []
            for s in sids:
                poss.append(nxt[s]); nxt[s] += 1
            a, b = r.find_slot(sids, poss), uc.find_slot(sids, poss)
            assert a == b, (step, op)
            if a:
                r.advance_head(len(sids)); uc.advance_head(len(sids))
            else:
                for s in sids:
                    nxt[s] -= 1
        elif op == "rm":
            s = int(rng.integers(-1, n_seq)); p0 = int(rng.integers(-1, 12)); p1 = int(rng.choice([-1, p0 + int(rng.integers(1, 8))]))
            assert r.seq_rm(s, p0, p1) == uc.seq_rm(s, p0, p1)
        elif op == "cp":
            a, b = (int(x) for x in rng.permutation(n_seq)[:2]); p0 = int(rng.integers(-1, 6)); p1 = int(rng.choice([-1, p0 + int(rng.integers(1, 10))]))
            r.seq_cp(a, b, p0, p1); uc.seq_cp(a, b, p0, p1)
            nxt[b] = max(nxt[b], nxt[a])
        elif op == "keep":
            s = int(rng.integers(n_seq)); r.seq_keep(s); uc.seq_keep(s)
        elif op == "add":
            s = int(rng.integers(n_seq)); p0 = int(rng.integers(0, 8)); d = int(rng.integers(-2, 4))
            r.seq_add(s, p0, -1, d); uc.seq_add(s, p0, -1, d)
            nxt[s] = max(0, nxt[s] + d)
        elif op == "div":
            s = int(rng.integers(n_seq)); r.seq_div(s, 0, -1, 2); uc.seq_div(s, 0, -1, 2)
        else:
            r.clear(); uc.clear()
        _same_cells(r, uc)
        assert r.head == uc.head, (step, op)
        assert r.get_used_cells() == uc.get_used_cells() and r.get_n_tokens() == uc.get_n_tokens(), (step, op)
        assert r.has_shift == uc.has_shift
        for s in range(n_seq):
            assert r.seq_pos_max(s) == uc.seq_pos_max(s)
        if step % 10 == 0:                                       # the session file of this state, byte for byte
            for sid in (-1, int(rng.integers(n_seq))):
                k = [r.tensor_get(0, False, size * ROW[1]).reshape(size, -1)]
                v = [r.tensor_get(0, True, size * ROW[1]).reshape(size, -1)]
                assert state_write(uc, k, v, 1, 1, sid) == r.state_write(sid), (step, sid)


@pytest.mark.parametrize("tk,tv,fa", [(1, 1, True), (8, 8, True), (2, 8, True), (1, 1, False)])
def test_state_files_are_byte_identical_and_cross_readable(tk, tv, fa):
    """rows included; fa=False = transposed V (v_trans, f16 only)"""
    rng = np.random.default_rng(tk * 7 + tv + fa)
    size = 40
    r = ref_kv.RefKvCache(n_layer=L, n_head_kv=HK, head_dim=D, type_k=tk, type_v=tv, kv_size=size, flash_attn=fa)
    assert r.v_trans == (not fa)
    uc = UnifiedCells(size)
    for c in (r, uc):
        c.find_slot([0] * 9, list(range(9))); c.advance_head(9)
        c.seq_cp(0, 1, -1, -1)
        for t in range(6):
            c.find_slot([0, 1], [9 + t, 9 + t]); c.advance_head(2)
        c.seq_rm(1, 3, 5)
    k = [rng.integers(0, 256, (size, ROW[tk]), dtype=np.uint8) for _ in range(L)]
    v = [rng.integers(0, 256, (size, ROW[tv]), dtype=np.uint8) for _ in range(L)]
    for il in range(L):
        r.tensor_set(il, False, k[il])
        # the reference's transposed V tensor is [n_embd][kv_size]; our row view of it is the transpose
        r.tensor_set(il, True, v[il] if fa else np.ascontiguousarray(v[il].view(np.uint16).T))
    for sid in (-1, 0, 1):
        blob = r.state_write(sid)
        assert state_write(uc, k, v, tk, tv, sid, v_trans=not fa) == blob
        # the restatement reads the reference's file ...
        uc2 = UnifiedCells(size)
        k2, v2 = [np.zeros_like(x) for x in k], [np.zeros_like(x) for x in v]
        assert state_read(uc2, k2, v2, tk, tv, blob, -1 if sid == -1 else 2)
        # ... exactly as the reference does
        r2 = ref_kv.RefKvCache(n_layer=L, n_head_kv=HK, head_dim=D, type_k=tk, type_v=tv, kv_size=size, flash_attn=fa)
        assert r2.state_read(blob, -1 if sid == -1 else 2)
        _same_cells(r2, uc2)
        assert r2.head == uc2.head and r2.get_used_cells() == uc2.get_used_cells()
        for il in range(L):
            assert np.array_equal(r2.tensor_get(il, False, size * ROW[tk]).reshape(size, -1), k2[il])
            got_v = r2.tensor_get(il, True, size * ROW[tv])
            got_v = got_v.reshape(size, -1) if fa else np.ascontiguousarray(got_v.view(np.uint16).reshape(-1, size).T).view(np.uint8)
            assert np.array_equal(got_v, v2[il])
    # a file the reference rejects, the restatement rejects too (sequence file read as a whole-cache file is fine; the
    # converse — a whole-cache file into one sequence — is the "seq_id-agnostic" error, :1025-1028)
    whole = r.state_write(-1)
    assert not ref_kv.RefKvCache(n_layer=L, n_head_kv=HK, head_dim=D, type_k=tk, type_v=tv, kv_size=size, flash_attn=fa).state_read(whole, 1)
    assert not state_read(UnifiedCells(size), [np.zeros_like(x) for x in k], [np.zeros_like(x) for x in v], tk, tv, whole, 1)


def test_debug_view_matches_the_reference(pkv):
    """llama_kv_cache_view_update (:1364-1433).  A single sequence laid out from block 0 occupies the same cell indices in the
    paged cache as in the ring, so the whole view — stats, positions (+ pending delta), sequence lists — must coincide."""
    size, bs = 64, 16
    r = ref_kv.RefKvCache(n_layer=1, n_head_kv=HK, head_dim=D, type_k=1, type_v=1, kv_size=size)
    c = pkv.PagedKVCache(n_blocks=size // bs, block_size=bs, n_seq_max=3, n_layer=0)
    r.find_slot([0] * 40, list(range(40))); r.advance_head(40)
    c.find_slot([0] * 40, list(range(40)))
    for x in (r, c):
        x.seq_rm(0, 5, 9)                                         # a hole (not a whole block: the paged table keeps it)
        x.seq_add(0, 20, -1, 3)                                   # pending shift shows up as pos + delta
    a, b = r.view(3), c.view(3)
    for k in ("n_cells", "token_count", "used_cells", "max_contiguous", "max_contiguous_idx"):
        assert a[k] == b[k], (k, a[k], b[k])
    used = a["cells_sequences"][:, 0] >= 0
    assert np.array_equal(used, b["cells_sequences"][:, 0] >= 0)
    assert np.array_equal(a["cells"][used], b["cells"][used])
    assert np.array_equal(a["cells_sequences"], b["cells_sequences"])
