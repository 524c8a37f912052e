"""Own host loop of a llama decode step — `llm_build_llama` (/root/reference/src/llama-model.cpp:4037-4198) for ONE token of ONE
sequence, driven through the C ABI only (no reference code anywhere on this path):

    inp_embd = get_rows(tok_embd, token)                                   :4047   pkv_get_rows
    per layer:
      attn_norm + Qcur, Kcur, Vcur   rms_norm(x) * w, wq / wk / wv          :4059   pkv_gemv_q   (ONE launch: norm + q8_K column in its prologue)
      rope, KV store, attention over the paged cache                        :4091   pkv_attn_decode_step    (+ the split-KV merge)
      attn_out + residual  wo product, ffn_inp = cur + inpSA                :4120   pkv_gemv_q   (residual epilogue)
      ffn_norm + SwiGLU    silu(gate x') * (up x'), x' = rms_norm(x) * w    :4140   pkv_gemv_q   (norm prologue, silu_mul epilogue)
      ffn_down + residual                                                   :4160   pkv_gemv_q   (residual epilogue)
    output_norm + lm_head                                                   :4180   pkv_gemv_q
  = 6 launches per layer.

The host half of `llama_context::decode` (:1155-1487) is the paged block manager: find_slot -> block table / sequence length, no dense
mask.  Every per-step input (token id, position, sequence length, block-table row) lives in device buffers that the host refreshes
before the launches, so the whole step is ONE captured hipGraph that is replayed per token (`capture()`), as bench.py's `e2e` record
times it.  Weights that are not K-quants (f32 / f16 / q8_0 files) take the generic pkv_mul_mat products with the same structure.

The loader is gguf_synth.read_gguf (llama-architecture keys, /root/reference/src/llama-model.cpp:406-521); tensors are uploaded as the
reference's own bytes."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import (F16, Q4_K, Q5_K, Q6_K, PACKED, act_image, weights_pack, PagedKVCache, PkvError, attn_decode_step, attn_workspace, gemv_q, get_rows, mul_mat, add, rms_norm_mul, rope_cache,
               rope_params, silu_mul, kv_store)
from .gguf_synth import read_gguf, row_bytes

_KQ = (Q4_K, Q5_K, Q6_K)    # row types of the decode-step products (pkv_gemv_q)


class LlamaDecoder:
    def __init__(self, path: str, *, n_ctx: int, type_k: int = F16, type_v: int = F16, block_size: int = 128, device: int = 0, pack_weights: bool = True):
        import torch
        self.torch = torch
        meta, tensors, _ = read_gguf(path)
        get = lambda k, d=None: meta[k][1] if k in meta else d   # noqa: E731
        if get("general.architecture") != "llama":
            raise PkvError(-5, "LlamaDecoder: llama architecture only")
        self.n_layer, self.n_embd, self.n_ff = get("llama.block_count"), get("llama.embedding_length"), get("llama.feed_forward_length")
        self.n_head = get("llama.attention.head_count")
        self.n_head_kv = get("llama.attention.head_count_kv", self.n_head)
        self.D = get("llama.attention.key_length", self.n_embd // self.n_head)
        self.eps = float(get("llama.attention.layer_norm_rms_epsilon", 1e-5))
        self.n_ctx, self.BS, self.tk, self.tv = n_ctx, block_size, type_k, type_v
        dev = torch.device("cuda", device)
        self.dev = dev

        def up(name):
            ne, t, _, data = tensors[name]
            w = torch.from_numpy(np.array(data)).to(dev)                    # a copy: the file map is read-only
            ne, t = tuple(int(x) for x in ne), int(t)
            if pack_weights and t in _KQ and len(ne) == 2 and ne[0] % 256 == 0:
                # K-quant rows are re-laid inside their own bytes at upload (plane layout, csrc/pkv_wlayout.h; the reference's CPU backend repacks on
                # set_tensor too, /root/reference/ggml/src/ggml-cpu/ggml-cpu-aarch64.cpp:4145-4172): same integers, same results, coalesced streaming
                w, t = weights_pack(w, t, ne[1], ne[0]), t | PACKED
            return w, t, ne
        self.w: Dict[str, tuple] = {name: up(name) for name in tensors}
        self.n_vocab = self.w["token_embd.weight"][2][1]
        if "output.weight" not in self.w:                             # tied embeddings (:1601-1606)
            self.w["output.weight"] = self.w["token_embd.weight"]
        ff = self.w.get("rope_freqs.weight")
        self.rope = rope_params(n_dims=get("llama.rope.dimension_count", self.D), mode=0, n_ctx_orig=get("llama.context_length", 8192),
                                freq_base=float(get("llama.rope.freq_base", 10000.0)),
                                freq_factors=ff[0].view(torch.float32) if ff is not None else None)
        self.scale = self.D ** -0.5
        n_blocks = -(-n_ctx // block_size)
        self.cache = PagedKVCache(n_blocks=n_blocks, block_size=block_size, n_seq_max=1, n_layer=self.n_layer, n_head_kv=self.n_head_kv,
                                  head_dim=self.D, type_k=type_k, type_v=type_v, device=device)
        E, F = self.n_embd, self.n_ff
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)   # noqa: E731
        self.h, self.xn = f32(E), f32(E)
        self.q, self.k_cur, self.v_cur = f32(1, self.n_head, self.D), f32(1, self.n_head_kv, self.D), f32(1, self.n_head_kv, self.D)
        self.attn, self.act, self.logits = f32(1, self.n_head, self.D), f32(F), f32(self.n_vocab)
        self.gate, self.upv, self.tmp_e = f32(F), f32(F), f32(E)
        self.ws = attn_workspace(1, self.n_head, self.D, n_ctx, block_size, device=dev)
        self.tok = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seq_lens = torch.zeros(1, dtype=torch.int32, device=dev)
        self.block_table = torch.zeros((1, n_blocks), dtype=torch.int32, device=dev)
        self.rope_cs = torch.zeros((1, self.D // 2, 2), dtype=torch.float32, device=dev)
        self.graph = None
        self.graph_len = 0
        self.trace = None
        self.attn_flags = 0                 # e.g. ATTN_FUSED_COMBINE: the split-KV merge inside the decode launch (measured)
        # opt-in experiment (use_attn_image()): the attention merge launch also leaves its row as the q8_K image the wo product streams against
        # (pkv_attn_args::out_image).  Measured end to end in one process (probes/e2e_profile.py [--no-image], 8B Q4_K_M mix, ctx 4096): 604 tok/s
        # with it, 608.5 without — the wo product's quantization of its column already hides under the latency of its first weight unit
        # (5.04 vs 5.11 us per launch), and the merge launch is not the cheaper for writing 5 KB more.  Off.
        self.attn_img = None

    def use_attn_image(self, on: bool = True):
        ok = (256 % self.D == 0 and (self.n_head * self.D) % 256 == 0 and self._kq(["blk.0.attn_output.weight"]))
        self.attn_img = act_image(self.n_head * self.D, self.dev) if (on and ok) else None
        self.graph = None

    # ---- model bytes ---------------------------------------------------------------------------------------------------------
    def weight_bytes(self) -> int:
        """bytes of weights one decode step reads: every layer tensor + output norm + lm_head (+ ONE row of the embedding)"""
        n = 0
        for name, (t, ty, ne) in self.w.items():
            if name == "token_embd.weight":
                n += row_bytes(ty & ~PACKED, ne[0])
                if self.w["output.weight"] is not self.w["token_embd.weight"]:
                    continue
            if name == "rope_freqs.weight":
                continue
            n += t.numel()
        return n

    # ---- products ------------------------------------------------------------------------------------------------------------
    def _kq(self, names) -> bool:
        ts = [self.w[n][1] for n in names]
        return (all((t & ~PACKED) in _KQ for t in ts) and all(self.w[n][2][0] % 256 == 0 and self.w[n][2][1] % 2 == 0 for n in names)
                and len({t & PACKED for t in ts}) == 1)

    def _products(self, names, outs, x_f32, *, norm=None, residual=None, silu=False, image=None):
        """outs[i] = W_i x' (+ residual | SwiGLU) with x' = x, or rms_norm(x) * norm weights: ONE pkv_gemv_q launch for K-quant rows (the column is
        normalised and quantized inside it), pkv_rms_norm_mul_f32 / pkv_mul_mat + element-wise launches otherwise"""
        ws = [self.w[n] for n in names]
        K = ws[0][2][0]
        nw = self.w[norm][0].view(self.torch.float32) if norm is not None else None
        if self._kq(names) and (nw is None or K <= 8192):
            # image: the column as the q8_K activation image its producer left (the attention merge launch: pkv_attn_args::out_image)
            gemv_q([w[0] for w in ws], [w[1] for w in ws], [w[2][1] for w in ws], K, image if nw is None else None, outs, residual=residual, silu_mul=silu,
                   x=x_f32, norm_w=nw, eps=self.eps)
            return
        if nw is not None:
            self.xn.copy_(rms_norm_mul(x_f32.reshape(1, -1), nw, self.eps).reshape(-1))
            x_f32 = self.xn
        x2 = x_f32.reshape(1, K)
        if silu:
            mul_mat(ws[0][0], ws[0][1], ws[0][2][1], K, x2, out=self.gate.reshape(1, -1))
            mul_mat(ws[1][0], ws[1][1], ws[1][2][1], K, x2, out=self.upv.reshape(1, -1))
            outs[0].copy_(silu_mul(self.gate, self.upv))
            return
        for w, o in zip(ws, outs):
            if residual is not None:
                mul_mat(w[0], w[1], w[2][1], K, x2, out=self.tmp_e.reshape(1, -1))
                add(residual.reshape(1, -1), self.tmp_e.reshape(1, -1), out=o.reshape(1, -1))
            else:
                mul_mat(w[0], w[1], w[2][1], K, x2, out=o.reshape(1, -1))

    # ---- one step: every launch of one token, reading tok / pos / seq_lens / block_table from device buffers --------------------
    def launches(self, max_seq_len: Optional[int] = None):
        E = self.n_embd
        te = self.w["token_embd.weight"]
        get_rows(te[0], te[1], self.tok, E, self.n_vocab, out=self.h.reshape(1, E))
        rope_cache(self.pos, self.rope, self.D, out=self.rope_cs)
        msl = max_seq_len or self.n_ctx
        for l in range(self.n_layer):
            p = f"blk.{l}."
            self._products([p + "attn_q.weight", p + "attn_k.weight", p + "attn_v.weight"],
                           [self.q.reshape(-1), self.k_cur.reshape(-1), self.v_cur.reshape(-1)], self.h, norm=p + "attn_norm.weight")
            attn_decode_step(self.q, self.k_cur, self.v_cur, self.pos, self.rope, self.cache.k_pool(l), self.cache.v_pool(l), self.block_table,
                             self.seq_lens, n_head_kv=self.n_head_kv, type_k=self.tk, type_v=self.tv, block_size=self.BS, scale=self.scale,
                             max_seq_len=msl, out=self.attn, workspace=self.ws, rope_cache=self.rope_cs, flags=self.attn_flags,
                             out_image=self.attn_img if not self.attn_flags else None)
            if self.trace is not None:   # debugging aid (eager launches only): per-layer copies of the attention path's inputs and output
                self.trace.append({k: getattr(self, k).detach().clone() for k in ("q", "k_cur", "v_cur", "attn")})
            self._products([p + "attn_output.weight"], [self.h], self.attn.reshape(-1), residual=self.h, image=self.attn_img if not self.attn_flags else None)
            self._products([p + "ffn_gate.weight", p + "ffn_up.weight"], [self.act], self.h, norm=p + "ffn_norm.weight", silu=True)
            self._products([p + "ffn_down.weight"], [self.h], self.act, residual=self.h)
        self._products(["output.weight"], [self.logits], self.h, norm="output_norm.weight")

    # ---- host half ---------------------------------------------------------------------------------------------------------------
    def clear(self):
        self.cache.clear()

    def _place(self, token: int, pos: int):
        torch = self.torch
        self.cache.update(self.rope)
        self.cache.find_slot([0], [pos])
        self.cache.update(self.rope)
        bt, sl = self.cache.block_table([0], self.block_table.shape[1])
        self.block_table.copy_(torch.from_numpy(bt))
        self.seq_lens.copy_(torch.from_numpy(sl))
        self.pos.fill_(pos)
        self.tok.fill_(token)
        return int(sl[0])

    def eval(self, token: int, pos: int):
        """decode one token at position `pos` (appended to the sequence); returns the device logits tensor [n_vocab]"""
        n = self._place(token, pos)
        if self.graph is not None and n <= self.graph_len:
            self.graph.replay()
        else:
            self.launches(max_seq_len=-(-n // self.BS) * self.BS)
        return self.logits

    def capture(self, stream=None):
        """capture the step as ONE hipGraph sized for the full context (splits past the sequence's end exit at once); eval() then replays it"""
        torch = self.torch
        st = stream or torch.cuda.Stream()
        with torch.cuda.stream(st):
            self.launches(max_seq_len=self.n_ctx)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                self.launches(max_seq_len=self.n_ctx)
        self.graph, self.graph_len = g, self.n_ctx
        return g

    def fill_kv_synthetic(self, n_past: int, seed: int = 0):
        """benchmarks: occupy cells [0, n_past) of the sequence with seeded random K / V rows through the real store path (no prompt run)"""
        torch = self.torch
        self.cache.clear()
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        CH = 1024
        for p0 in range(0, n_past, CH):
            n = min(CH, n_past - p0)
            pos = np.arange(p0, p0 + n, dtype=np.int32)
            blk, off = self.cache.find_slot([0] * n, pos)
            self.cache.update(self.rope)
            sb, so, dp = (torch.from_numpy(x).to(self.dev) for x in (blk, off, pos))
            for l in range(self.n_layer):
                k = torch.rand((n, self.n_head_kv, self.D), generator=gen, device=self.dev) * 2 - 1
                v = torch.rand((n, self.n_head_kv, self.D), generator=gen, device=self.dev) * 2 - 1
                kv_store(k, v, self.cache.k_pool(l), self.cache.v_pool(l), type_k=self.tk, type_v=self.tv, slot_block=sb, slot_off=so,
                         block_size=self.BS, pos=dp, rope=self.rope)
        torch.cuda.synchronize()

    def close(self):
        self.cache.close()
