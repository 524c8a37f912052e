"""The fork prototype's toy layer, replayed from its own test programs (which assert nothing — the expectations below are
what their semantics, recovered in SURVEY.md §8(a-toy), must print).  Sources of the prototype are absent from the
reference tree: beyond these sequences this layer is "parity unpinned"."""
import ctypes as C

import numpy as np


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class BKC:
    def __init__(self, L, nb, bs, h, d):
        self.L, self.nb, self.be = L, nb, bs * h * d
        L.pkv_toy_bkc_create.restype = C.c_void_p
        L.pkv_toy_bkc_free.restype = None
        for f in ("pkv_toy_bkc_init_cache", "pkv_toy_bkc_write_to_cache", "pkv_toy_bkc_swap_blocks", "pkv_toy_bkc_copy_blocks",
                  "pkv_toy_attention_fwd", "pkv_toy_attention_paged_fwd", "pkv_toy_attention_real_fwd"):
            getattr(L, f).restype = None
        self.h = C.c_void_p(L.pkv_toy_bkc_create(nb, bs, h, d))
        L.pkv_toy_bkc_init_cache(self.h)

    def __del__(self):
        self.L.pkv_toy_bkc_free(self.h)

    def get(self, v=0):
        n = self.L.pkv_toy_bkc_get(self.h, v, None, 0)
        out = np.empty(n, np.float32)
        self.L.pkv_toy_bkc_get(self.h, v, ptr(out), n)
        return out.reshape(-1, self.be)

    def write(self, k, v, m):
        m = np.ascontiguousarray(m, np.int32)
        self.L.pkv_toy_bkc_write_to_cache(self.h, ptr(f32(k)), ptr(f32(v)), ptr(m), len(m))

    def insert(self, k, v):
        b, s = C.c_int(), C.c_int()
        assert self.L.pkv_toy_bkc_insert_kv_block(self.h, ptr(f32(k)), ptr(f32(v)), C.byref(b), C.byref(s)) == 0
        return b.value, s.value


def test_block_kv_cache_program(pkv):
    """tests/test_block_kv_cache.cpp: num_blocks 4, block_size 2, heads 1, head_size 4 (be = 8)"""
    L = pkv.lib()
    c1, c2 = BKC(L, 4, 2, 1, 4), BKC(L, 4, 2, 1, 4)
    j = np.arange(8)
    k1 = np.stack([(i + 1) * 10 + j for i in range(2)]); v1 = np.stack([(i + 1) * 100 + j for i in range(2)])
    c1.write(k1, v1, [1, 3])
    exp1 = np.zeros((4, 8), np.float32); exp1[1], exp1[3] = k1[0], k1[1]
    assert np.array_equal(c1.get(0), exp1)
    assert np.array_equal(c1.get(1)[3], v1[1])
    k2 = np.stack([(i + 1) * 20 + j for i in range(2)]); v2 = np.stack([(i + 1) * 200 + j for i in range(2)])
    c2.write(k2, v2, [0, 2])
    exp2 = np.zeros((4, 8), np.float32); exp2[0], exp2[2] = k2[0], k2[1]
    assert np.array_equal(c2.get(0), exp2)
    # swapBlocks(cache2, {0, 2}): THIS block i <-> OTHER block map[i]  => c1[0]<->c2[0], c1[1]<->c2[2]
    m = np.array([0, 2], np.int32)
    L.pkv_toy_bkc_swap_blocks(c1.h, c2.h, ptr(m), 2)
    a1, a2 = exp1.copy(), exp2.copy()
    a1[0], a2[0] = exp2[0].copy(), exp1[0].copy()
    a1[1], a2[2] = exp2[2].copy(), exp1[1].copy()
    assert np.array_equal(c1.get(0), a1) and np.array_equal(c2.get(0), a2)
    # copyBlocks(cache2, {2}): same-index copy src[2] -> this[2]
    m = np.array([2], np.int32)
    L.pkv_toy_bkc_copy_blocks(c1.h, c2.h, ptr(m), 1)
    a1[2] = a2[2]
    assert np.array_equal(c1.get(0), a1)


def test_kv_cache_program(pkv):
    """tests/test_kv_cache.cpp: insertKVBlock x3, free the 2nd, insert again -> (0,0) (0,1) (1,0) then (0,1)"""
    c = BKC(pkv.lib(), 2, 4, 2, 3)
    k = np.arange(24, dtype=np.float32); v = k * 2
    k2, v2 = k * 3, k * 4
    e1, e2, e3 = c.insert(k, v), c.insert(k2, v2), c.insert(k, v)
    assert (e1, e2, e3) == ((0, 0), (0, 1), (1, 0))
    assert pkv.lib().pkv_toy_bkc_free_kv_block(c.h, e2[0], e2[1]) == 0
    assert c.insert(k2, v2) == (0, 1)
    kc = c.get(0)          # storage index = physical_block * slot_capacity + slot, slot capacity = num_blocks = 2
    assert np.array_equal(kc[0], k) and np.array_equal(kc[1], k2) and np.array_equal(kc[2], k)


def test_attention_programs(pkv):
    """tests/test_block_kv_attention.cpp and test_paged_attention.cpp: the dummy attention returns the query; the K/V block lands in its slot"""
    L = pkv.lib()
    c = BKC(L, 4, 2, 1, 4)
    q = np.arange(1, 9, dtype=np.float32)
    k = (np.arange(8, dtype=np.float32) + 1) * 10; v = (np.arange(8, dtype=np.float32) + 1) * 100
    out = np.zeros(8, np.float32)
    m = np.array([1], np.int32)
    L.pkv_toy_attention_fwd(ptr(q), 8, c.h, ptr(k), ptr(v), ptr(m), 1, ptr(out))
    assert np.array_equal(out, q) and np.array_equal(c.get(0)[1], k) and np.array_equal(c.get(1)[1], v)
    c2 = BKC(L, 2, 4, 2, 3)
    q = np.arange(25, dtype=np.float32)
    k = np.arange(1, 25, dtype=np.float32); v = k * 2
    out = np.zeros(25, np.float32)
    L.pkv_toy_attention_paged_fwd(ptr(q), 25, c2.h, ptr(k), ptr(v), ptr(out), 10)      # pages of 10, 10, 5
    assert np.array_equal(out, q) and np.array_equal(c2.get(0)[0], k)
    # blockKVAttentionRealFwd: scalar 4-key softmax
    q = np.array([0.5, -1.0, 2.0], np.float32); k4 = np.array([1, 2, 3, 4], np.float32); v4 = np.array([10, 20, 30, 40], np.float32)
    out = np.zeros(3, np.float32)
    L.pkv_toy_attention_real_fwd(ptr(q), 3, ptr(k4), ptr(v4), ptr(out))
    s = q[:, None] * k4[None]
    p = np.exp(s - s.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    assert np.allclose(out, p @ v4, rtol=1e-6)
