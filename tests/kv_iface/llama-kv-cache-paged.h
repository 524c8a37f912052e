// tests/kv_iface/llama-kv-cache-paged.h — the reference-side file of INTEGRATION.md §1, COMPILED: `llama_kv_cache_paged`, the paged block
// manager of libmi355_paged_kv.so behind the reference's own memory interface.
//
//   struct llama_kv_cache : llama_memory_i      /root/reference/src/llama-kv-cache.h:17-26, /root/reference/src/llama-memory.h:7-21
//   llama_kv_cache_unified::find_slot(ubatch)   /root/reference/src/llama-kv-cache.h:103-108 (same signature here)
//
// It is built by oracle/Makefile (`make ref`) against the reference's headers where they lie (nothing of the reference is copied) into
// oracle/_ref/libkv_iface.so together with tests/kv_iface/kv_iface_driver.cpp, and driven by tests/test_kv_iface.py through the reference's
// llama_kv_cache_* wrappers (/root/reference/src/llama-kv-cache.cpp:1227-1332), i.e. through the virtual table, next to
// llama_kv_cache_unified.  A maintainer drops this header into src/ unchanged.
#pragma once

#include "llama-batch.h"
#include "llama-kv-cache.h"

#include "mi355_paged_kv.h"

#include <stdexcept>
#include <vector>

struct llama_kv_cache_paged : public llama_kv_cache {
    pkv_cache * c = nullptr;

    // the slot mapping of the last ubatch (what the KV-store launch consumes: pkv_kv_store / pkv_qkv_prepare slot_block, slot_off)
    std::vector<int32_t> slot_block, slot_off;

    // n_layer == 0: host bookkeeping only (no device memory; the unit tests of the interface run without a GPU)
    llama_kv_cache_paged(int32_t n_layer, int32_t n_head_kv, int32_t head_dim, ggml_type type_k, ggml_type type_v, uint32_t kv_size,
                         uint32_t n_seq_max, int32_t block_size = 128, int32_t device = 0) {
        pkv_cache_params p = {};
        p.n_layer = n_layer;          p.n_head_kv = n_head_kv;   p.head_dim = head_dim;
        p.type_k  = (int32_t) type_k; p.type_v    = (int32_t) type_v;          // PKV_TYPE_* are ggml_type's values
        p.block_size = block_size;    p.n_blocks  = (int32_t) ((kv_size + block_size - 1) / block_size);
        p.n_seq_max  = (int32_t) n_seq_max;   p.device = device;
        c = pkv_cache_create(&p);
        if (!c) throw std::runtime_error(pkv_last_error());                     // llama_kv_cache_unified::init returning false (:19-124)
    }
    ~llama_kv_cache_paged() { pkv_cache_free(c); }
    llama_kv_cache_paged(const llama_kv_cache_paged &) = delete;
    llama_kv_cache_paged & operator=(const llama_kv_cache_paged &) = delete;

    // ---- llama_memory_i ----
    void clear()  override { pkv_cache_clear(c); }
    void defrag() override { pkv_cache_defrag(c); }
    bool seq_rm  (llama_seq_id seq_id, llama_pos p0, llama_pos p1) override { return pkv_cache_seq_rm(c, seq_id, p0, p1) != 0; }
    void seq_cp  (llama_seq_id seq_id_src, llama_seq_id seq_id_dst, llama_pos p0, llama_pos p1) override { pkv_cache_seq_cp(c, seq_id_src, seq_id_dst, p0, p1); }
    void seq_keep(llama_seq_id seq_id) override { pkv_cache_seq_keep(c, seq_id); }
    void seq_add (llama_seq_id seq_id, llama_pos p0, llama_pos p1, llama_pos delta) override { pkv_cache_seq_add(c, seq_id, p0, p1, delta); }
    void seq_div (llama_seq_id seq_id, llama_pos p0, llama_pos p1, int d) override { pkv_cache_seq_div(c, seq_id, p0, p1, d); }
    llama_pos seq_pos_max(llama_seq_id seq_id) override { return pkv_cache_seq_pos_max(c, seq_id); }

    // ---- llama_kv_cache ----
    int32_t  get_n_tokens()   const override { return pkv_cache_n_tokens(c); }
    uint32_t get_used_cells() const override { return (uint32_t) pkv_cache_used_cells(c); }
    bool     get_can_shift()  const override { return pkv_cache_can_shift(c) != 0; }

    // ---- what llama_context::decode calls on the concrete type (src/llama-context.cpp:1297-1314) ----
    // token i of the ubatch belongs to sequence seq_id[i / n_seq_tokens][0] (the non-recurrent branch of the reference's find_slot reads
    // seq_id[s][j] for every j: a token listed in several sequences is placed once per sequence here, the first id owning the slot mapping)
    llama_kv_cache_slot_info find_slot(const llama_ubatch & batch) {
        const uint32_t n = batch.n_tokens;
        std::vector<int32_t> sid(n), pos(n);
        for (uint32_t s = 0; s < batch.n_seqs; ++s)
            for (uint32_t j = 0; j < batch.n_seq_tokens; ++j) {
                const uint32_t i = s * batch.n_seq_tokens + j;
                sid[i] = batch.seq_id[s][0];
                pos[i] = batch.pos[i];
            }
        slot_block.assign(n, -1);
        slot_off.assign(n, -1);
        if (pkv_cache_find_slot(c, (int) n, sid.data(), pos.data(), slot_block.data(), slot_off.data()) != PKV_OK)
            return llama_kv_cache_slot_info(false);                              // "failed to find KV cache slot" (:1303-1308)
        return llama_kv_cache_slot_info(0, n);
    }
};
