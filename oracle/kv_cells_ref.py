"""Pure-Python restatement of the reference's KV-cell bookkeeping.  TEST INFRASTRUCTURE ONLY (small cases).

`UnifiedCells` follows llama_kv_cache_unified for transformer (non-recurrent) models, /root/reference/src/llama-kv-cache.cpp:
clear :158-171, seq_rm :173-242, seq_cp :244-293, seq_keep :295-325, seq_add :327-383, seq_div :385-429,
seq_pos_max :431-441, find_slot (non-recurrent branch) :622-672, get_n_tokens :133-141, get_used_cells (`used`).
It is the oracle for WHAT each sequence can see (the multiset of positions per sequence) — the paged block manager
must expose the same view through its block tables.

`ToyBlockManager` / `ToyKVCacheManager` restate the fork prototype recovered in SURVEY.md §8(a-toy)
(/root/reference/tests/test_block_manager.cpp, /root/reference/tests/test_kv_cache.cpp); the prototype's sources are
absent from the reference tree and its tests assert nothing, so beyond the index sequences those tests print this
part is "parity unpinned".
"""
from __future__ import annotations

INT_MAX = 2 ** 31 - 1


class Cell:
    __slots__ = ("pos", "delta", "seq")

    def __init__(self):
        self.pos, self.delta, self.seq = -1, 0, set()


class UnifiedCells:
    def __init__(self, size: int):
        self.size, self.head, self.used = size, 0, 0
        self.cells = [Cell() for _ in range(size)]
        self.has_shift = False

    def clear(self):
        for c in self.cells:
            c.pos, c.seq = -1, set()
        self.head = self.used = 0

    def find_slot(self, seq_ids, poss) -> bool:
        n = len(seq_ids)
        if n > self.size:
            return False
        tested = 0
        while True:
            if self.head + n > self.size:
                tested += self.size - self.head
                self.head = 0
                continue
            found = True
            for i in range(n):
                if self.cells[self.head + i].pos >= 0:
                    found = False
                    self.head += i + 1
                    tested += i + 1
                    break
            if found:
                break
            if tested >= self.size:
                return False
        for k in range(n):
            c = self.cells[self.head + k]
            c.pos = poss[k]
            c.seq.add(seq_ids[k])
        self.used += n
        return True

    def advance_head(self, n):  # llama_context::decode :1347-1352
        self.head += n
        if self.head >= self.size:
            self.head = 0

    @staticmethod
    def _range(p0, p1):
        return (0 if p0 < 0 else p0), (INT_MAX if p1 < 0 else p1)

    def seq_rm(self, seq_id, p0, p1) -> bool:
        p0, p1 = self._range(p0, p1)
        new_head = self.size
        for i, c in enumerate(self.cells):
            if p0 <= c.pos < p1:
                if seq_id < 0:
                    c.seq.clear()
                elif seq_id in c.seq:
                    c.seq.discard(seq_id)
                else:
                    continue
                if not c.seq:
                    if c.pos >= 0:
                        self.used -= 1
                    c.pos = -1
                    if new_head == self.size:
                        new_head = i
        if new_head != self.size and new_head < self.head:
            self.head = new_head
        return True

    def seq_cp(self, src, dst, p0, p1):
        if src == dst:
            return
        p0, p1 = self._range(p0, p1)
        self.head = 0
        for c in self.cells:
            if src in c.seq and p0 <= c.pos < p1:
                c.seq.add(dst)

    def seq_keep(self, seq_id):
        new_head = self.size
        for i, c in enumerate(self.cells):
            if seq_id not in c.seq:
                if c.pos >= 0:
                    self.used -= 1
                c.pos = -1
                c.seq.clear()
                if new_head == self.size:
                    new_head = i
            else:
                c.seq = {seq_id}
        if new_head != self.size and new_head < self.head:
            self.head = new_head

    def seq_add(self, seq_id, p0, p1, delta):
        if delta == 0:
            return
        p0, p1 = self._range(p0, p1)
        if p0 == p1:
            return
        new_head = self.size
        for i, c in enumerate(self.cells):
            if seq_id in c.seq and p0 <= c.pos < p1:
                self.has_shift = True
                c.pos += delta
                c.delta += delta
                if c.pos < 0:
                    if c.seq:
                        self.used -= 1
                    c.pos = -1
                    c.seq.clear()
                    if new_head == self.size:
                        new_head = i
        self.head = new_head if new_head != self.size else 0

    def seq_div(self, seq_id, p0, p1, d):
        if d == 1:
            return
        p0, p1 = self._range(p0, p1)
        if p0 == p1:
            return
        for c in self.cells:
            if seq_id in c.seq and p0 <= c.pos < p1:
                self.has_shift = True
                old = c.pos
                c.pos //= d
                c.delta += c.pos - old

    def seq_pos_max(self, seq_id) -> int:
        r = 0
        for c in self.cells:
            if seq_id in c.seq:
                r = max(r, c.pos)
        return r

    def get_n_tokens(self) -> int:
        return sum(len(c.seq) for c in self.cells)

    def get_used_cells(self) -> int:
        return self.used

    # what a sequence attends to
    def positions(self, seq_id):
        return sorted(c.pos for c in self.cells if seq_id in c.seq)

    def shared(self, seq_id) -> bool:
        return any(seq_id in c.seq and len(c.seq) > 1 for c in self.cells)


# ---- session state: llama_kv_cache_unified::state_write / state_read (:848-1221) over UnifiedCells + flat K/V rows ----
# k_rows / v_rows: per layer, a numpy uint8 array [size][row bytes] = the flat k_l[il] / v_l[il] tensors (one row per cell).
# "parity unpinned": libllama cannot be compiled from a few files here, so this restatement is pinned by reading the
# reference source only (format constants and order cited line by line), not by running it.
import struct

import numpy as np


def state_write(uc: "UnifiedCells", k_rows, v_rows, type_k: int, type_v: int, seq_id: int = -1, v_trans: bool = False) -> bytes:
    ranges, count, begin = [], 0, uc.size                        # :852-871
    for i, c in enumerate(uc.cells):
        if (seq_id == -1 and c.seq) or (seq_id in c.seq):       # !is_empty() || has_seq_id
            count += 1
            if begin == uc.size:
                begin = i
        elif begin != uc.size:
            ranges.append((begin, i))
            begin = uc.size
    if begin != uc.size:
        ranges.append((begin, uc.size))
    out = [struct.pack("<I", count)]                              # :880
    for a, b in ranges:                                           # state_write_meta :904-921
        for i in range(a, b):
            c = uc.cells[i]
            ids = sorted(c.seq) if seq_id == -1 else []
            out.append(struct.pack("<iI", c.pos, len(ids)))
            out += [struct.pack("<i", s) for s in ids]
    n_layer = len(k_rows)
    out.append(struct.pack("<II", 1 if v_trans else 0, n_layer))   # :927-928
    for il in range(n_layer):                                     # keys :934-952
        out.append(struct.pack("<iQ", type_k, k_rows[il].shape[1]))
        out += [k_rows[il][a:b].tobytes() for a, b in ranges]
    for il in range(n_layer):
        if not v_trans:                                           # :954-973
            out.append(struct.pack("<iQ", type_v, v_rows[il].shape[1]))
            out += [v_rows[il][a:b].tobytes() for a, b in ranges]
        else:                                                     # :974-1000 (f16 V stored [n_embd][kv_size])
            n_embd = v_rows[il].shape[1] // 2
            out.append(struct.pack("<iII", type_v, 2, n_embd))
            vt = v_rows[il].view(np.uint16)                       # [size][n_embd] rows; the transposed tensor is vt.T
            for j in range(n_embd):
                out += [vt[a:b, j].tobytes() for a, b in ranges]
    return b"".join(out)


def state_read(uc: "UnifiedCells", k_rows, v_rows, type_k: int, type_v: int, data: bytes, seq_id: int = -1) -> bool:
    """restores cells and rows in place; returns False where the reference returns false (the caller then clears)"""
    off = 0

    def rd(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, data, off)
        off += struct.calcsize(fmt)
        return v
    (count,) = rd("<I")
    if seq_id != -1:                                              # :1004-1042
        uc.seq_rm(seq_id, -1, -1)
        poss = []
        for _ in range(count):
            pos, n = rd("<iI")
            if n != 0:
                return False
            poss.append(pos)
        if count and not uc.find_slot([seq_id] * count, poss):
            return False
    else:                                                         # :1043-1090
        if count > uc.size:
            return False
        uc.clear()
        for i in range(count):
            pos, n = rd("<iI")
            c = uc.cells[i]
            c.pos = pos
            for _ in range(n):
                (s,) = rd("<i")
                if s < 0:
                    return False
                c.seq.add(s)
        uc.head, uc.used = 0, count
    v_trans, n_layer = rd("<II")                                  # state_read_data :1105-1221
    if n_layer != len(k_rows) or count > uc.size:
        return False
    h = uc.head
    for il in range(n_layer):
        t, row = rd("<iQ")
        if t != type_k or row != k_rows[il].shape[1]:
            return False
        k_rows[il][h:h + count] = np.frombuffer(data, np.uint8, count * row, off).reshape(count, row)
        off += count * row
    for il in range(n_layer):
        if not v_trans:
            t, row = rd("<iQ")
            if t != type_v or row != v_rows[il].shape[1]:
                return False
            v_rows[il][h:h + count] = np.frombuffer(data, np.uint8, count * row, off).reshape(count, row)
            off += count * row
        else:
            t, el, n_embd = rd("<iII")
            if t != type_v or el != 2 or n_embd * 2 != v_rows[il].shape[1]:
                return False
            vt = v_rows[il].view(np.uint16)
            for j in range(n_embd):
                vt[h:h + count, j] = np.frombuffer(data, np.uint16, count, off)
                off += count * 2
    return True


class ToyBlockManager:
    def __init__(self, n):
        self.used = [False] * n

    def allocateBlock(self):
        for i, u in enumerate(self.used):
            if not u:
                self.used[i] = True
                return i
        for i, u in enumerate(self.used):  # evictBlock: lowest allocated index, cleared then re-marked
            if u:
                return i
        return -1

    def freeBlock(self, i):
        if 0 <= i < len(self.used):
            self.used[i] = False

    def getBlockStatus(self):
        return list(self.used)


class ToyKVCacheManager:
    def __init__(self, cap):
        self.cap, self.blocks = cap, [[False] * cap]

    def insert(self):
        for b, blk in enumerate(self.blocks):
            for s, u in enumerate(blk):
                if not u:
                    blk[s] = True
                    return b, s
        self.blocks.append([False] * self.cap)
        self.blocks[-1][0] = True
        return len(self.blocks) - 1, 0

    def free(self, e):
        b, s = e
        if 0 <= b < len(self.blocks) and 0 <= s < self.cap:
            self.blocks[b][s] = False
            return True
        return False
