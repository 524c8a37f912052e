"""Session state: pkv_cache_state_write / _read speak the byte format of
llama_kv_cache_unified::state_write / state_read (/root/reference/src/llama-kv-cache.cpp:848-1221).

CPU part: bookkeeping-only caches (n_layer = 0) against the Python restatement of the reference's cell ring
(oracle/kv_cells_ref.py).  The restatement is pinned to the reference twice: tests/test_kv_cells_vs_ref.py drives the
compiled llama_kv_cache_unified beside it (dev container), and tests/golden/state_io.npz holds session files the
reference's own state_write produced (checked below, everywhere).  What must hold: files written by either side restore on the other side to the same view — for every sequence
the same multiset AND order of positions — and the paged side re-shares blocks that hold the same cells."""
import struct

import numpy as np
import pytest

from oracle.kv_cells_ref import UnifiedCells, state_read, state_write


def _paged(pkv, n_blocks=16, bs=4, n_seq=4, **kw):
    return pkv.PagedKVCache(n_blocks=n_blocks, block_size=bs, n_seq_max=n_seq, n_layer=0, **kw)


def _seq_positions(c, s):
    return [p for p in c.seq_slot_pos(s) if p >= 0]


def _ref_positions(uc, s):
    return [c.pos for c in uc.cells if s in c.seq]


def test_single_sequence_file_is_byte_identical_to_reference(pkv):
    c, uc = _paged(pkv), UnifiedCells(64)
    pos = list(range(11))
    c.find_slot([0] * 11, pos)
    uc.find_slot([0] * 11, pos)
    for sid in (0, -1):
        assert c.state_write(sid) == state_write(uc, [], [], 1, 1, sid)
    # the layout itself: u32 count, (i32 pos, u32 n_seq_id[, i32 id])*, u32 v_trans, u32 n_layer
    raw = c.state_write(-1)
    assert struct.unpack_from("<I", raw, 0)[0] == 11
    assert struct.unpack_from("<iIi", raw, 4) == (0, 1, 0)
    assert struct.unpack_from("<II", raw, len(raw) - 8) == (0, 0)


def test_reference_ring_restores_into_block_tables(pkv):
    """interleaved decode (token k of the ubatch -> cell head+k, :658-667) + a shared prompt: per-sequence views survive"""
    uc = UnifiedCells(64)
    uc.find_slot([0] * 9, list(range(9)))                        # prompt of seq 0
    uc.advance_head(9)
    uc.seq_cp(0, 1, -1, -1)                                      # shared by seq 1 and 2
    uc.seq_cp(0, 2, 0, 8)
    for step in range(5):                                        # three sequences decode, interleaved cells
        uc.find_slot([0, 1, 2], [9 + step, 9 + step, 8 + step])
        uc.advance_head(3)
    uc.seq_rm(2, 9, 11)                                          # a hole in seq 2
    data = state_write(uc, [], [], 1, 1, -1)
    c = _paged(pkv)
    c.find_slot([3] * 5, list(range(5)))                         # stale content that the restore must drop
    c.state_read(data, -1)
    for s in range(4):
        assert _seq_positions(c, s) == _ref_positions(uc, s), s
    assert c.get_n_tokens() == uc.get_n_tokens()
    # sharing is per block: prompt cell 8 sits in a block whose other cells differ between seq 0 and seq 1 -> one private copy
    assert c.get_used_cells() == uc.get_used_cells() + 1
    # blocks that hold the same cells are shared again: the first two 4-cell blocks of the prompt by all three sequences
    b0, b1, b2 = c.seq_blocks(0), c.seq_blocks(1), c.seq_blocks(2)
    assert list(b0[:2]) == list(b1[:2]) == list(b2[:2])
    assert c.block_refcount(b0[0]) == 3 and c.block_refcount(b0[1]) == 3
    # and the other way: what the paged cache writes, the reference ring reads back to the same views
    uc2 = UnifiedCells(64)
    assert state_read(uc2, [], [], 1, 1, c.state_write(-1), -1)
    for s in range(4):
        assert sorted(_ref_positions(uc2, s)) == sorted(_ref_positions(uc, s))
    assert uc2.get_n_tokens() == uc.get_n_tokens() and uc2.get_used_cells() == c.get_used_cells()


def test_sequence_file_moves_between_caches_and_ids(pkv):
    a = _paged(pkv)
    a.find_slot([1] * 7, [3, 4, 5, 6, 7, 8, 9])                  # positions need not start at 0
    a.find_slot([0] * 3, [0, 1, 2])
    blob = a.state_write(1)
    assert struct.unpack_from("<I", blob, 0)[0] == 7 and struct.unpack_from("<iI", blob, 4) == (3, 0)
    b = _paged(pkv)
    b.find_slot([2] * 6, list(range(6)))
    b.state_read(blob, 2)                                        # replaces whatever seq 2 held (:1007)
    assert _seq_positions(b, 2) == [3, 4, 5, 6, 7, 8, 9]
    uc = UnifiedCells(32)
    assert state_read(uc, [], [], 1, 1, blob, 5)
    assert _ref_positions(uc, 5) == [3, 4, 5, 6, 7, 8, 9]
    with pytest.raises(pkv.PkvError):                            # a whole-cache file is not a sequence file (:1025-1028)
        b.state_read(a.state_write(-1), 1)
    assert _seq_positions(b, 1) == []


def test_restore_failures_leave_the_cache_cleared(pkv):
    a = _paged(pkv, n_blocks=8)
    a.find_slot([0] * 20, list(range(20)))
    blob = a.state_write(-1)
    small = _paged(pkv, n_blocks=2)                              # 8 cells < 20
    small.find_slot([0] * 3, [0, 1, 2])
    with pytest.raises(pkv.PkvError):
        small.state_read(blob, -1)
    assert small.get_n_tokens() == 0 and small.n_free_blocks() == 2
    b = _paged(pkv)
    b.find_slot([1] * 3, [0, 1, 2])
    with pytest.raises(pkv.PkvError):
        b.state_read(blob[:30], -1)                              # truncated
    assert b.get_n_tokens() == 0
    bad = bytearray(blob)
    struct.pack_into("<i", bad, 12, 99)                          # first cell's seq id out of range
    with pytest.raises(pkv.PkvError):
        b.state_read(bytes(bad), -1)
    wrong_layers = blob[:-4] + struct.pack("<I", 3)              # n_layer 3 != 0
    with pytest.raises(pkv.PkvError):
        b.state_read(wrong_layers, -1)
    assert b.get_n_tokens() == 0


def test_pending_device_work_blocks_the_save(pkv):
    c = _paged(pkv)
    c.find_slot([0] * 6, list(range(6)))
    c.seq_add(0, 2, -1, 5)                                       # K-shift pending: rows and positions disagree until update()
    with pytest.raises(pkv.PkvError):
        c.state_write(-1)
    c.update()
    assert struct.unpack_from("<I", c.state_write(-1), 0)[0] == 6


def _golden_ring(g, name):
    tk, tv, v_trans, n_layer, hk, d, size = (int(x) for x in g[f"{name}__meta"])
    uc = UnifiedCells(size)
    for i, c in enumerate(uc.cells):
        c.pos = int(g[f"{name}__cell_pos"][i])
        c.seq = set(int(x) for x in g[f"{name}__cell_seqs"][i][:g[f"{name}__cell_nseq"][i]])
    uc.used = sum(1 for c in uc.cells if c.seq)
    k = [g[f"{name}__k{il}"] for il in range(n_layer)]
    v = [g[f"{name}__v{il}"] for il in range(n_layer)]
    return uc, k, v, tk, tv, bool(v_trans), size


def test_restatement_reproduces_the_reference_written_files():
    """tests/golden/state_io.npz: files written by the reference's state_write (tests/golden/make_state_golden.py)"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "state_io.npz"))
    for name in g["names"]:
        uc, k, v, tk, tv, v_trans, size = _golden_ring(g, name)
        for sid, key in ((-1, "blob"), (1, "blob_seq1")):
            blob = g[f"{name}__{key}"].tobytes()
            assert state_write(uc, k, v, tk, tv, sid, v_trans=v_trans) == blob, (name, sid)
            uc2 = UnifiedCells(size)
            k2, v2 = [np.zeros_like(x) for x in k], [np.zeros_like(x) for x in v]
            assert state_read(uc2, k2, v2, tk, tv, blob, sid)
            assert _ref_positions(uc2, 1) == _ref_positions(uc, 1)
            if sid == -1:
                assert all(np.array_equal(a, b) for a, b in zip(k + v, k2 + v2))


def test_restatement_round_trip_with_rows():
    """the restatement against itself, rows included, both V layouts (keeps the GPU tests' checker honest)"""
    rng = np.random.default_rng(1)
    uc = UnifiedCells(16)
    uc.find_slot([0, 1, 0, 1, 0], [0, 0, 1, 1, 2])
    k = [rng.integers(0, 256, (16, 64), dtype=np.uint8) for _ in range(2)]
    v = [rng.integers(0, 256, (16, 64), dtype=np.uint8) for _ in range(2)]
    for vt in (False, True):
        blob = state_write(uc, k, v, 1, 1, -1, v_trans=vt)
        uc2 = UnifiedCells(16)
        k2, v2 = [np.zeros_like(x) for x in k], [np.zeros_like(x) for x in v]
        assert state_read(uc2, k2, v2, 1, 1, blob, -1)
        assert all(np.array_equal(a[:5], b[:5]) for a, b in zip(k + v, k2 + v2))
        assert _ref_positions(uc2, 0) == [0, 1, 2] and _ref_positions(uc2, 1) == [0, 1]
