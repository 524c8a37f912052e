// tests/kv_iface/kv_iface_driver.cpp — TEST INFRASTRUCTURE ONLY.  One C handle over a `llama_kv_cache *` that is EITHER the reference's
// llama_kv_cache_unified OR llama_kv_cache_paged (llama-kv-cache-paged.h, our block manager behind the reference's interface): every
// edit goes through the reference's own llama_kv_cache_* wrappers (/root/reference/src/llama-kv-cache.cpp:1227-1332), i.e. through the
// base-class pointer and the virtual table; find_slot through the concrete type, as llama_context::decode does
// (/root/reference/src/llama-context.cpp:181, :1301).  Built by `make -C oracle ref` into oracle/_ref/libkv_iface.so.
#include "llama-kv-cache-paged.h"

#include "llama-cparams.h"
#include "llama-model.h"

#include <memory>
#include <set>

namespace {
struct Handle {
    int kind;                                     // 0 = llama_kv_cache_unified, 1 = llama_kv_cache_paged
    llama_model   model;
    llama_cparams cparams{};
    std::unique_ptr<llama_kv_cache> kv;           // the ONLY pointer the edit calls see
    Handle() : model(llama_model_default_params()) {}
};
}  // namespace

extern "C" {

void * kvi_create(int kind, int n_head_kv, int head_dim, int kv_size, int n_seq_max, int block_size) {
    auto * h = new Handle();
    h->kind = kind;
    try {
        if (kind == 0) {
            h->model.arch = LLM_ARCH_LLAMA;
            llama_hparams & hp = h->model.hparams;
            hp.n_layer = 1;
            hp.n_embd_head_k = hp.n_embd_head_v = head_dim;
            hp.n_head_arr[0] = n_head_kv; hp.n_head_kv_arr[0] = n_head_kv;
            h->cparams.flash_attn = true;
            auto * u = new llama_kv_cache_unified(hp, {nullptr});
            h->kv.reset(u);
            if (!u->init(h->model, h->cparams, GGML_TYPE_F16, GGML_TYPE_F16, (uint32_t) kv_size, false)) { delete h; return nullptr; }
        } else {
            h->kv.reset(new llama_kv_cache_paged(/*n_layer*/ 0, n_head_kv, head_dim, GGML_TYPE_F16, GGML_TYPE_F16, (uint32_t) kv_size,
                                                 (uint32_t) n_seq_max, block_size));
        }
    } catch (const std::exception &) { delete h; return nullptr; }
    return h;
}
void kvi_free(void * h) { delete (Handle *) h; }

static llama_kv_cache * kv_of(void * h) { return ((Handle *) h)->kv.get(); }

// ---- the virtual interface, through the reference's wrappers -------------------------------------------------------------------------
int  kvi_n_tokens(void * h)   { return llama_kv_cache_n_tokens(kv_of(h)); }
int  kvi_used_cells(void * h) { return llama_kv_cache_used_cells(kv_of(h)); }
void kvi_clear(void * h)      { llama_kv_cache_clear(kv_of(h)); }
int  kvi_seq_rm(void * h, int s, int p0, int p1)          { return llama_kv_cache_seq_rm(kv_of(h), s, p0, p1); }
void kvi_seq_cp(void * h, int a, int b, int p0, int p1)   { llama_kv_cache_seq_cp(kv_of(h), a, b, p0, p1); }
void kvi_seq_keep(void * h, int s)                        { llama_kv_cache_seq_keep(kv_of(h), s); }
void kvi_seq_add(void * h, int s, int p0, int p1, int d)  { llama_kv_cache_seq_add(kv_of(h), s, p0, p1, d); }
void kvi_seq_div(void * h, int s, int p0, int p1, int d)  { llama_kv_cache_seq_div(kv_of(h), s, p0, p1, d); }
int  kvi_seq_pos_max(void * h, int s)                     { return llama_kv_cache_seq_pos_max(kv_of(h), s); }
void kvi_defrag(void * h)                                 { llama_kv_cache_defrag(kv_of(h)); }
int  kvi_can_shift(void * h)                              { return llama_kv_cache_can_shift(kv_of(h)); }
int  kvi_can_edit(void * h)                               { return kv_of(h)->get_can_edit(); }      // llama_memory_i::get_can_edit

// ---- one ubatch: llama_sbatch simple split (llama-batch.cpp:129-151), find_slot on the concrete type, then what decode does on success ---
int kvi_find_slot(void * hh, int n_tok, const int32_t * seq_id, const int32_t * pos, int32_t * out_block, int32_t * out_off) {
    auto * h = (Handle *) hh;
    std::vector<llama_pos>      p(pos, pos + n_tok);
    std::vector<int32_t>        nsi(n_tok, 1);
    std::vector<llama_seq_id>   ids(seq_id, seq_id + n_tok);
    std::vector<llama_seq_id *> pid(n_tok);
    for (int i = 0; i < n_tok; ++i) pid[i] = &ids[i];
    llama_ubatch ub{};
    ub.equal_seqs = false; ub.n_tokens = n_tok; ub.n_seq_tokens = 1; ub.n_seqs = n_tok;
    ub.pos = p.data(); ub.n_seq_id = nsi.data(); ub.seq_id = pid.data();
    if (h->kind == 0) {
        auto * u = static_cast<llama_kv_cache_unified *>(h->kv.get());
        const auto slot = u->find_slot(ub);
        if (!slot) return 0;
        for (int i = 0; i < n_tok; ++i) { if (out_block) out_block[i] = 0; if (out_off) out_off[i] = (int32_t) u->head + i; }
        u->head += n_tok;                                       // llama-context.cpp:1347-1352
        if (u->head >= u->size) u->head = 0;
        return 1;
    }
    auto * g = static_cast<llama_kv_cache_paged *>(h->kv.get());
    const auto slot = g->find_slot(ub);
    if (!slot) return 0;
    for (int i = 0; i < n_tok; ++i) { if (out_block) out_block[i] = g->slot_block[i]; if (out_off) out_off[i] = g->slot_off[i]; }
    return 1;
}

// ---- the per-sequence view: the positions sequence `s` sees, ascending (what attention over that sequence reads) ------------------------
int kvi_seq_positions(void * hh, int s, int32_t * out, int cap) {
    auto * h = (Handle *) hh;
    std::multiset<int32_t> ps;
    if (h->kind == 0) {
        auto * u = static_cast<llama_kv_cache_unified *>(h->kv.get());
        for (uint32_t i = 0; i < u->size; ++i)
            if (u->cells[i].has_seq_id(s)) ps.insert(u->cells[i].pos);
    } else {
        auto * g = static_cast<llama_kv_cache_paged *>(h->kv.get());
        const int n = pkv_cache_seq_len(g->c, s);
        std::vector<int32_t> sp((size_t) (n > 0 ? n : 1));
        const int m = pkv_cache_seq_slot_pos(g->c, s, sp.data(), n);
        for (int i = 0; i < m && i < n; ++i) if (sp[i] >= 0) ps.insert(sp[i]);
    }
    int k = 0;
    for (int32_t p : ps) { if (k < cap) out[k] = p; ++k; }
    return k;
}
// is any cell of `s` shared with another sequence (the documented deviation: seq_add / seq_div on shared cells copy-on-write here)
int kvi_seq_shared(void * hh, int s) {
    auto * h = (Handle *) hh;
    if (h->kind != 0) return -1;
    auto * u = static_cast<llama_kv_cache_unified *>(h->kv.get());
    for (uint32_t i = 0; i < u->size; ++i)
        if (u->cells[i].has_seq_id(s) && u->cells[i].seq_id.size() > 1) return 1;
    return 0;
}
int kvi_n_free_blocks(void * hh) {
    auto * h = (Handle *) hh;
    return h->kind == 1 ? pkv_cache_n_free_blocks(static_cast<llama_kv_cache_paged *>(h->kv.get())->c) : -1;
}

}  // extern "C"
