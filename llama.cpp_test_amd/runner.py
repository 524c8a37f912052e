"""Host-side decode loop for the attention path — the part of `llama_context::decode`
(/root/reference/src/llama-context.cpp:1155-1487) that touches the KV cache, re-expressed over the paged block manager:

    kv_self_update()          -> PagedKVCache.update()            (:1293)
    kv_self->find_slot(ubatch) -> PagedKVCache.find_slot()         (:1301)   + update() for queued copy-on-write copies
    set_inputs(KQ_mask, pos)  -> block_table / seq_lens / q_start (:1330; the dense mask is gone)
    per layer: rope(Q), rope(K), cpy K/V -> cache, flash_attn_ext  (/root/reference/src/llama-model.cpp:4091-4105)
                              -> attn_decode_step (one token per sequence: everything in one launch + merge, the
                                 (cos, sin) rows built once per ubatch) | qkv_prepare + attn_prefill

Everything model-side (projections, FFN, sampling) is outside this path: the caller supplies the per-layer Q/K/V
activations as device tensors (benchmarks use random ones, like llama-bench uses random tokens).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import (ATTN_NONCAUSAL, PagedKVCache, PkvError, RopeParams, attn_decode, attn_decode_step, attn_prefill,  # noqa: F401
               attn_prefill_step, attn_prefill_workspace_bytes, attn_step_layers, attn_workspace, qkv_prepare, rope_cache, rope_params)


class AttnPathRunner:
    def __init__(self, *, n_layer: int, n_head: int, n_head_kv: int, head_dim: int, type_k: int, type_v: int, n_blocks: int,
                 n_seq_max: int, block_size: int = 128, rope: RopeParams | None = None, max_batch_tokens: int = 2048,
                 max_seq_len: int | None = None, device: int = 0):
        import torch
        self.torch = torch
        self.L, self.H, self.HK, self.D, self.BS = n_layer, n_head, n_head_kv, head_dim, block_size
        self.tk, self.tv = type_k, type_v
        self.cache = PagedKVCache(n_blocks=n_blocks, block_size=block_size, n_seq_max=n_seq_max, n_layer=n_layer,
                                  n_head_kv=n_head_kv, head_dim=head_dim, type_k=type_k, type_v=type_v, device=device)
        self.rope = rope if rope is not None else rope_params(n_dims=head_dim)
        self.scale = head_dim ** -0.5
        self.max_seq_len = max_seq_len or n_blocks * block_size
        # scratch of the split-KV decode (one query token per sequence, so at most n_seq_max tokens) and of the prefill
        # kernel's split-KV mode (small ubatches over a long past)
        self.ws = attn_workspace(n_seq_max, n_head, head_dim, self.max_seq_len, block_size,
                                 min_bytes=attn_prefill_workspace_bytes(min(max_batch_tokens, 2048), n_head, head_dim))
        self.max_batch_tokens = max_batch_tokens
        # prefill ubatches as ONE fused step per layer (K/V store + attention with RoPE(Q) in its Q fetch) where the rope pairs are NORM (llama)
        self.fused_prefill = (self.rope.mode & 2) == 0

    @classmethod
    def from_gguf(cls, path: str, *, type_k: int, type_v: int, n_blocks: int, n_seq_max: int, **kw):
        """geometry and RoPE parameters from a llama-architecture GGUF (the keys llama_model::load_hparams reads,
        /root/reference/src/llama-model.cpp:406-521) — the minimal loader half of SURVEY §8(f) row 2"""
        from .gguf_synth import read_gguf
        meta, _, _ = read_gguf(path)

        def get(key, default=None):
            return meta[key][1] if key in meta else default
        arch = get("general.architecture")
        if arch != "llama":
            raise PkvError(-4, f"unsupported architecture {arch!r} (llama only)")
        n_embd, n_head = get("llama.embedding_length"), get("llama.attention.head_count")
        hd = get("llama.attention.key_length", n_embd // n_head)
        rope = rope_params(n_dims=get("llama.rope.dimension_count", hd), mode=0, n_ctx_orig=get("llama.context_length", 8192),
                           freq_base=get("llama.rope.freq_base", 10000.0))
        return cls(n_layer=get("llama.block_count"), n_head=n_head, n_head_kv=get("llama.attention.head_count_kv", n_head),
                   head_dim=hd, type_k=type_k, type_v=type_v, n_blocks=n_blocks, n_seq_max=n_seq_max, rope=rope, **kw)

    # llama_kv_self_* pass-throughs used by drivers
    def clear(self):
        self.cache.clear()

    def seq_rm(self, seq_id, p0, p1):
        return self.cache.seq_rm(seq_id, p0, p1)

    def seq_cp(self, src, dst, p0, p1):
        self.cache.seq_cp(src, dst, p0, p1)

    def plan(self, seq_id: Sequence[int], pos: Sequence[int], *, with_rope_cache: bool = True):
        """host half of one ubatch: slot placement + the small device tables.  Tokens of one sequence must be adjacent
        (llama_sbatch simple split keeps batch order; batched-bench and llama-bench batches satisfy this)."""
        torch = self.torch
        sid = np.asarray(seq_id, np.int32)
        p = np.asarray(pos, np.int32)
        n_tok = len(sid)
        assert 0 < n_tok <= self.max_batch_tokens
        self.cache.update(self.rope)                               # pending K-shift / copies first (kv_self_update)
        blk, off = self.cache.find_slot(sid, p)                     # raises PkvError(ERR_NO_SPACE) like a failed find_slot
        self.cache.update(self.rope)                               # copy-on-write copies queued by find_slot
        # runs of equal seq_id -> sequences of this ubatch
        starts = [0] + (np.flatnonzero(sid[1:] != sid[:-1]) + 1).tolist() + [n_tok]
        seqs = [int(sid[s]) for s in starts[:-1]]
        assert len(set(seqs)) == len(seqs), "tokens of one sequence must be adjacent in the ubatch"
        n_q = np.diff(starts)
        bt, sl, mb, regular = self.cache.block_tables(seqs, -(-self.max_seq_len // self.BS))
        if not regular:
            raise PkvError(-5, "irregular sequence (holes / shifted cells): call defrag() + update() first")
        bt = bt[:, :max(1, mb)]
        all_decode = bool((n_q == 1).all())
        # the ubatch's six small tables in ONE upload (each H2D copy is ~5 us of the stream's time: 6 of them were 4 % of a 32-sequence decode step)
        parts = [np.asarray(blk, np.int32).ravel(), np.asarray(off, np.int32).ravel(), p, np.asarray(sl, np.int32), np.asarray(starts, np.int32),
                 np.ascontiguousarray(bt, np.int32).ravel()]
        offs, tot = [], 0
        for a in parts:
            offs.append(tot)
            tot += -(-a.size // 4) * 4                               # 16-byte aligned pieces
        # ... from PINNED host memory, asynchronously: a pageable source makes the copy a synchronisation point of the stream — the host then builds the next ubatch's
        # tables and starts its launches only after the previous ubatch has drained.  With all layers of a ubatch behind ONE call into the library (layers()) the
        # host is a fraction of a decode step ahead of the device instead.  A small ring of pinned buffers, each guarded by the event of the copy that last read it.
        slot = self._pin_slot(tot)
        host = slot["np"]
        host[:tot] = 0
        for a, o in zip(parts, offs):
            host[o:o + a.size] = a
        dall = torch.empty(tot, dtype=torch.int32, device="cuda")
        dall.copy_(slot["t"][:tot], non_blocking=True)
        slot["ev"].record()
        d_blk, d_off, dpos, d_sl, d_qs, d_bt = (dall[o:o + a.size] for a, o in zip(parts, offs))
        return dict(n_tok=n_tok, n_seq=len(seqs), max_q=int(n_q.max()), all_decode=all_decode,
                    max_seq_len=int(sl.max()), slot_block=d_blk, slot_off=d_off, pos=dpos, block_table=d_bt.view(bt.shape[0], bt.shape[1]),
                    seq_lens=d_sl, q_start=d_qs,
                    # ggml_rope_cache_init once per ubatch, shared by every layer's fused decode step
                    # ggml_rope_cache_init once per ubatch, shared by every layer's fused step (decode: attn_decode_step; prefill, NORM pairs: attn_prefill_step)
                    rope_cache=rope_cache(dpos, self.rope, self.D) if (with_rope_cache and (all_decode or self.fused_prefill)) else None)

    def _pin_slot(self, n: int):
        torch = self.torch
        ring = getattr(self, "_pins", None)
        if ring is None:
            ring = self._pins = {"i": 0, "slots": []}
        if len(ring["slots"]) < 8:
            t = torch.empty(max(4096, 2 * n), dtype=torch.int32).pin_memory()
            ring["slots"].append({"t": t, "np": t.numpy(), "ev": torch.cuda.Event()})
            return ring["slots"][-1]
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        sl = ring["slots"][ring["i"]]
        sl["ev"].synchronize()                                   # the copy that read this buffer eight plans ago
        if sl["t"].numel() < n:
            t = torch.empty(2 * n, dtype=torch.int32).pin_memory()
            sl["t"], sl["np"] = t, t.numpy()
        return sl

    def layer(self, il: int, plan, q, k_cur, v_cur, out=None):
        """device half, one layer: q f32 [n_tok][n_head][D], k_cur/v_cur f32 [n_tok][n_head_kv][D], all un-rotated.
        A prompt ubatch through the fused step may hand q over as f16 and take `out` back as f16 (PKV_ATTN_Q_F16 / PKV_ATTN_OUT_F16: half the
        boundary traffic; for f16 K the reference rounds Q to f16 itself).
        Decode ubatches (one token per sequence) and fused prefill ubatches (NORM rope pairs) leave q untouched; prefill ubatches with
        NEOX pairs (fused_prefill False) rotate it in place."""
        c = self.cache
        kw = dict(n_head_kv=self.HK, type_k=self.tk, type_v=self.tv, block_size=self.BS, scale=self.scale,
                  max_seq_len=plan["max_seq_len"], out=out, workspace=self.ws)
        if plan["all_decode"]:
            return attn_decode_step(q, k_cur, v_cur, plan["pos"], self.rope, c.k_pool(il), c.v_pool(il), plan["block_table"],
                                    plan["seq_lens"], rope_cache=plan["rope_cache"], **kw)
        if self.fused_prefill and plan["rope_cache"] is not None:   # q stays un-rotated: RoPE(Q) happens in the attention kernel's Q fetch
            return attn_prefill_step(q, k_cur, v_cur, plan["pos"], self.rope, c.k_pool(il), c.v_pool(il), plan["block_table"], plan["seq_lens"],
                                     plan["q_start"], plan["slot_block"], plan["slot_off"], rope_cache=plan["rope_cache"], max_q_per_seq=plan["max_q"], **kw)
        if q.dtype != self.torch.float32:
            raise PkvError(-4, "f16 q needs the fused prefill step (NORM rope pairs): pkv_qkv_prepare rotates an f32 q in place")
        qkv_prepare(q, k_cur, v_cur, c.k_pool(il), c.v_pool(il), type_k=self.tk, type_v=self.tv, slot_block=plan["slot_block"],
                    slot_off=plan["slot_off"], block_size=self.BS, pos=plan["pos"], rope=self.rope)
        return attn_prefill(q, c.k_pool(il), c.v_pool(il), plan["block_table"], plan["seq_lens"], plan["q_start"],
                            max_q_per_seq=plan["max_q"], **kw)

    def layers(self, plan, q: List, k_cur: List, v_cur: List, out: List):
        """device half, ALL layers of a planned ubatch in one trip through the binding (pkv_attn_step_layers: the fused decode step, or the fused prefill step, per
        layer — what layer() launches, without n_layer Python calls of ~18 us each).  Falls back to layer() where the fused steps do not apply."""
        fused = plan["rope_cache"] is not None and (plan["all_decode"] or self.fused_prefill)
        if not fused:
            return [self.layer(il, plan, q[il], k_cur[il], v_cur[il], out[il]) for il in range(self.L)]
        c = self.cache
        if getattr(self, "_pools", None) is None:
            self._pools = ([c.k_pool(il) for il in range(self.L)], [c.v_pool(il) for il in range(self.L)])
        pre = not plan["all_decode"]
        return attn_step_layers(q, k_cur, v_cur, out, self._pools[0], self._pools[1], plan["pos"], self.rope, plan["block_table"], plan["seq_lens"],
                                rope_cache=plan["rope_cache"], n_head_kv=self.HK, type_k=self.tk, type_v=self.tv, block_size=self.BS, scale=self.scale,
                                max_seq_len=plan["max_seq_len"], workspace=self.ws, q_start=plan["q_start"] if pre else None, max_q_per_seq=plan["max_q"] if pre else 1,
                                slot_block=plan["slot_block"] if pre else None, slot_off=plan["slot_off"] if pre else None)

    def decode(self, seq_id, pos, q: List, k_cur: List, v_cur: List, out: List | None = None):
        """one ubatch through all layers; q/k_cur/v_cur: per-layer device tensors"""
        plan = self.plan(seq_id, pos)
        return [self.layer(il, plan, q[il], k_cur[il], v_cur[il], None if out is None else out[il]) for il in range(self.L)]

    # ---- the same ubatch with the device half replayed from a hipGraph ---------------------------------------------------------
    # An eager ubatch is 2 launches per layer through ctypes (~80 us of host time per layer, as much as the kernels of a 2048-token
    # ubatch take).  The launches of a ubatch depend on its SHAPE only (token count, sequence count, longest query run, decode or
    # prefill, the block-rounded longest sequence): its tables live in persistent device buffers that plan() refreshes, and the
    # per-layer launches are captured once per shape and caller-tensor set, then replayed (what the reference's CUDA backend does
    # for decode graphs, /root/reference/ggml/src/ggml-cuda/ggml-cuda.cu "cuda graphs").
    def _persistent(self):
        if getattr(self, "_pt", None) is None:
            torch = self.torch
            nt, ns = self.max_batch_tokens, self.cache.n_seq_max
            i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device="cuda")  # noqa: E731
            self._pt = dict(slot_block=i32(nt), slot_off=i32(nt), pos=i32(nt), block_table=i32(ns, -(-self.max_seq_len // self.BS)),
                            seq_lens=i32(ns), q_start=i32(ns + 1), rope_cache=torch.zeros((nt, self.D // 2, 2), dtype=torch.float32, device="cuda"))
            self._graphs = {}
            self._stream = torch.cuda.Stream()
        return self._pt

    def decode_captured(self, seq_id, pos, q: List, k_cur: List, v_cur: List, out: List):
        """decode() with out given, through a cached hipGraph.  Results and cache contents are those of decode() (same kernels, same
        arguments; max_seq_len is passed rounded up to whole blocks so that a growing sequence keeps its graph for a block)."""
        torch = self.torch
        plan = self.plan(seq_id, pos, with_rope_cache=False)
        pt = self._persistent()
        n_tok, n_seq = plan["n_tok"], plan["n_seq"]
        for k in ("slot_block", "slot_off", "pos"):
            pt[k][:n_tok].copy_(plan[k], non_blocking=True)
        bt = plan["block_table"]
        pt["block_table"][:n_seq, :bt.shape[1]].copy_(bt, non_blocking=True)
        pt["seq_lens"][:n_seq].copy_(plan["seq_lens"], non_blocking=True)
        pt["q_start"][:n_seq + 1].copy_(plan["q_start"], non_blocking=True)
        msl = -(-plan["max_seq_len"] // self.BS) * self.BS
        gplan = dict(plan, slot_block=pt["slot_block"][:n_tok], slot_off=pt["slot_off"][:n_tok], pos=pt["pos"][:n_tok],
                     block_table=pt["block_table"][:n_seq], seq_lens=pt["seq_lens"][:n_seq], q_start=pt["q_start"][:n_seq + 1],
                     max_seq_len=msl, rope_cache=pt["rope_cache"][:n_tok] if (plan["all_decode"] or self.fused_prefill) else None)
        key = (n_tok, n_seq, plan["max_q"], plan["all_decode"], msl) + tuple(x.data_ptr() for lst in (q, k_cur, v_cur, out) for x in lst)

        def launches():
            if gplan["rope_cache"] is not None:
                rope_cache(gplan["pos"], self.rope, self.D, out=gplan["rope_cache"])
            for il in range(self.L):
                self.layer(il, gplan, q[il], k_cur[il], v_cur[il], out[il])

        g = self._graphs.get(key)
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            if g is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self._stream):
                    launches()
                self._graphs[key] = g
            g.replay()
        cur.wait_stream(self._stream)
        return out

    def close(self):
        self.cache.close()
