"""b2 of SURVEY §8(b): the paged block manager behind the REFERENCE'S OWN memory interface, compiled.

tests/kv_iface/llama-kv-cache-paged.h declares `struct llama_kv_cache_paged : public llama_kv_cache` (/root/reference/src/llama-kv-cache.h:17-26,
/root/reference/src/llama-memory.h:7-21) over the C ABI of libmi355_paged_kv.so; `make -C oracle ref` compiles it against the reference's
headers where they lie, next to llama_kv_cache_unified, into oracle/_ref/libkv_iface.so.  Here both objects are driven through the SAME
`llama_kv_cache *` calls — the reference's llama_kv_cache_* wrappers (/root/reference/src/llama-kv-cache.cpp:1227-1332) — with the random
edit traffic of tests/test_block_manager.py, and must expose the same per-sequence views after every call.

No GPU needed (the paged object is created with n_layer = 0: bookkeeping only).  Exists where the reference was compiled (this
container); elsewhere tests/test_block_manager.py holds the same traffic against the restated ring (pinned by test_kv_cells_vs_ref.py)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libkv_iface.so")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="oracle/_ref/libkv_iface.so not built (needs /root/reference)")

_lib = None


def lib():
    global _lib
    if _lib is None:
        C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libggml_ref.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(PATH)
        L.kvi_create.restype = C.c_void_p
        for f in ("kvi_free", "kvi_clear", "kvi_seq_cp", "kvi_seq_keep", "kvi_seq_add", "kvi_seq_div", "kvi_defrag"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


class Kv:
    """a `llama_kv_cache *` (kind 0: llama_kv_cache_unified, kind 1: llama_kv_cache_paged); every method is one call on that pointer"""

    def __init__(self, kind, *, kv_size, n_seq_max=4, block_size=4):
        self.h = C.c_void_p(lib().kvi_create(kind, 2, 64, kv_size, n_seq_max, block_size))
        assert self.h, "construction failed"
        self.kind = kind

    def __del__(self):
        if getattr(self, "h", None):
            lib().kvi_free(self.h); self.h = None

    def find_slot(self, sids, poss):
        s, p = np.ascontiguousarray(sids, np.int32), np.ascontiguousarray(poss, np.int32)
        b, o = np.zeros(len(s), np.int32), np.zeros(len(s), np.int32)
        ok = lib().kvi_find_slot(self.h, len(s), s.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        return bool(ok), b, o

    def positions(self, s):
        out = np.zeros(4096, np.int32)
        n = lib().kvi_seq_positions(self.h, s, out.ctypes.data_as(C.c_void_p), len(out))
        return out[:n].tolist()

    def __getattr__(self, name):
        f = getattr(lib(), "kvi_" + name)
        return lambda *a: f(self.h, *a)


def test_both_objects_answer_through_the_base_pointer():
    for kind in (0, 1):
        k = Kv(kind, kv_size=64)
        assert k.n_tokens() == 0 and k.used_cells() == 0 and k.seq_pos_max(0) == 0   # an empty sequence answers 0 (llama-kv-cache.cpp:431-441)
        assert k.can_shift() == 1 and k.can_edit() == 1          # llama_memory_i::get_can_edit == get_can_shift (llama-kv-cache.h:25)
        ok, _, _ = k.find_slot([0] * 5 + [1] * 3, list(range(5)) + list(range(3)))
        assert ok and k.n_tokens() == 8 and k.used_cells() == 8
        assert k.positions(0) == [0, 1, 2, 3, 4] and k.positions(1) == [0, 1, 2]
        assert k.seq_rm(0, 3, -1) == 1 and k.positions(0) == [0, 1, 2] and k.seq_pos_max(0) == 2
        k.seq_cp(1, 2, -1, -1)
        assert k.positions(2) == [0, 1, 2] and k.n_tokens() == 9 and k.used_cells() == 6
        k.seq_keep(2)
        assert k.positions(0) == [] and k.positions(1) == [] and k.positions(2) == [0, 1, 2]
        k.seq_add(2, 1, -1, 4)
        assert k.positions(2) == [0, 5, 6]
        k.seq_div(2, 0, -1, 2)
        assert k.positions(2) == [0, 2, 3]
        k.defrag()
        k.clear()
        assert k.n_tokens() == 0 and k.positions(2) == []


def test_find_slot_failure_leaves_both_untouched():
    for kind in (0, 1):
        k = Kv(kind, kv_size=8, block_size=4)
        assert k.find_slot([0] * 6, list(range(6)))[0]
        before = (k.n_tokens(), k.used_cells(), k.positions(0))
        assert not k.find_slot([0] * 4, [6, 7, 8, 9])[0]        # 10 cells in a cache of 8: "failed to find KV cache slot" (llama-context.cpp:1303)
        assert (k.n_tokens(), k.used_cells(), k.positions(0)) == before


def test_paged_slots_follow_the_block_table():
    k = Kv(1, kv_size=32, block_size=4)
    ok, b, o = k.find_slot([0] * 6, list(range(6)))
    assert ok and b.tolist() == [0, 0, 0, 0, 1, 1] and o.tolist() == [0, 1, 2, 3, 0, 1]
    ok, b, o = k.find_slot([1, 0, 1], [0, 6, 1])
    assert ok and b.tolist() == [2, 1, 2] and o.tolist() == [0, 2, 1]
    assert k.n_free_blocks() == 5


OPS = ("decode", "decode", "decode", "prefill", "rm_tail", "rm_mid", "cp", "keep", "add", "div", "rm_all", "clear")


@pytest.mark.parametrize("seed", range(10))
def test_same_virtual_calls_same_views(seed):
    """random llama_kv_cache_* traffic on both objects: visible positions per sequence, n_tokens, seq_pos_max and every return value agree
    after every call.  Excluded, as in test_block_manager.py: seq_cp into a non-empty sequence and seq_add / seq_div on cells shared by several
    sequences (the reference shifts every owner; the paged manager copies on write: DESIGN.md §4, deliberate deviation)."""
    rng = np.random.default_rng(1000 + seed)
    n_seq = 4
    u, g = Kv(0, kv_size=96, n_seq_max=n_seq), Kv(1, kv_size=256, n_seq_max=n_seq, block_size=4)
    nxt = [0] * n_seq
    for step in range(150):
        op = OPS[int(rng.integers(len(OPS)))]
        s = int(rng.integers(n_seq))
        if op in ("decode", "prefill"):
            if op == "decode":
                sids = [int(x) for x in rng.permutation(n_seq)[: int(rng.integers(1, n_seq + 1))]]
                poss = [nxt[x] for x in sids]
            else:
                n = int(rng.integers(2, 9))
                sids, poss = [s] * n, list(range(nxt[s], nxt[s] + n))
            if not u.find_slot(sids, poss)[0]:
                continue                                         # the ring is the smaller of the two: skip what it cannot place
            assert g.find_slot(sids, poss)[0]
            for x, p in zip(sids, poss):
                nxt[x] = max(nxt[x], p + 1)
        elif op == "rm_tail":
            p0 = int(rng.integers(0, nxt[s] + 1))
            assert u.seq_rm(s, p0, -1) == g.seq_rm(s, p0, -1)
        elif op == "rm_mid":
            p0 = int(rng.integers(0, nxt[s] + 1)); p1 = p0 + int(rng.integers(0, 6))
            assert u.seq_rm(s, p0, p1) == g.seq_rm(s, p0, p1)
        elif op == "rm_all":
            if rng.random() < 0.3:
                p1 = int(rng.integers(0, 5))
                assert u.seq_rm(-1, 0, p1) == g.seq_rm(-1, 0, p1)
        elif op == "cp":
            d = int(rng.integers(n_seq))
            if d == s or u.positions(d):
                continue
            p1 = int(rng.integers(0, nxt[s] + 2))
            u.seq_cp(s, d, -1, p1); g.seq_cp(s, d, -1, p1)
        elif op == "keep":
            if rng.random() < 0.15:
                u.seq_keep(s); g.seq_keep(s)
        elif op in ("add", "div"):
            if u.seq_shared(s):
                continue
            p0 = int(rng.integers(0, nxt[s] + 1))
            if op == "add":
                d = int(rng.integers(-3, 4))
                u.seq_add(s, p0, -1, d); g.seq_add(s, p0, -1, d)
            else:
                u.seq_div(s, p0, -1, 2); g.seq_div(s, p0, -1, 2)
        elif op == "clear":
            if rng.random() < 0.1:
                u.clear(); g.clear()
        for x in range(n_seq):
            pu = u.positions(x)
            assert pu == g.positions(x), (step, op, x)
            assert u.seq_pos_max(x) == g.seq_pos_max(x), (step, op, x)
            nxt[x] = pu[-1] + 1 if pu else 0
        assert u.n_tokens() == g.n_tokens(), (step, op)
