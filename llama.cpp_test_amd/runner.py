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
               attn_prefill_workspace_bytes, attn_workspace, qkv_prepare, rope_cache, rope_params)


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

    def plan(self, seq_id: Sequence[int], pos: Sequence[int]):
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
        starts = [0] + [i for i in range(1, n_tok) if sid[i] != sid[i - 1]] + [n_tok]
        seqs = [int(sid[s]) for s in starts[:-1]]
        assert len(set(seqs)) == len(seqs), "tokens of one sequence must be adjacent in the ubatch"
        n_q = np.diff(starts)
        for s in seqs:
            if not self.cache.seq_is_regular(s):
                raise PkvError(-5, "irregular sequence (holes / shifted cells): call defrag() + update() first")
        bt, sl = self.cache.block_table(seqs)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        all_decode = bool((n_q == 1).all())
        dpos = dev(p)
        return dict(n_tok=n_tok, n_seq=len(seqs), max_q=int(n_q.max()), all_decode=all_decode,
                    max_seq_len=int(sl.max()), slot_block=dev(blk), slot_off=dev(off), pos=dpos, block_table=dev(bt),
                    seq_lens=dev(sl), q_start=dev(np.asarray(starts, np.int32)),
                    # ggml_rope_cache_init once per ubatch, shared by every layer's fused decode step
                    rope_cache=rope_cache(dpos, self.rope, self.D) if all_decode else None)

    def layer(self, il: int, plan, q, k_cur, v_cur, out=None):
        """device half, one layer: q f32 [n_tok][n_head][D], k_cur/v_cur f32 [n_tok][n_head_kv][D], all un-rotated.
        Decode ubatches (one token per sequence) leave q untouched; prefill ubatches rotate it in place."""
        c = self.cache
        kw = dict(n_head_kv=self.HK, type_k=self.tk, type_v=self.tv, block_size=self.BS, scale=self.scale,
                  max_seq_len=plan["max_seq_len"], out=out, workspace=self.ws)
        if plan["all_decode"]:
            return attn_decode_step(q, k_cur, v_cur, plan["pos"], self.rope, c.k_pool(il), c.v_pool(il), plan["block_table"],
                                    plan["seq_lens"], rope_cache=plan["rope_cache"], **kw)
        qkv_prepare(q, k_cur, v_cur, c.k_pool(il), c.v_pool(il), type_k=self.tk, type_v=self.tv, slot_block=plan["slot_block"],
                    slot_off=plan["slot_off"], block_size=self.BS, pos=plan["pos"], rope=self.rope)
        return attn_prefill(q, c.k_pool(il), c.v_pool(il), plan["block_table"], plan["seq_lens"], plan["q_start"],
                            max_q_per_seq=plan["max_q"], **kw)

    def decode(self, seq_id, pos, q: List, k_cur: List, v_cur: List, out: List | None = None):
        """one ubatch through all layers; q/k_cur/v_cur: per-layer device tensors"""
        plan = self.plan(seq_id, pos)
        return [self.layer(il, plan, q[il], k_cur[il], v_cur[il], None if out is None else out[il]) for il in range(self.L)]

    def close(self):
        self.cache.close()
