// oracle/ref_kv_driver.cpp — TEST INFRASTRUCTURE ONLY (never linked into, or called by, the product).
//
// Our own shim over the REFERENCE's llama_kv_cache_unified, compiled against the reference's llama sources where they
// lie (oracle/Makefile `make ref`: plain g++ on /root/reference/src/*.cpp, outputs only under oracle/_ref/).  It lets
// the tests drive the real cell ring — find_slot, seq_*, state_write / state_read — so that
//   * oracle/kv_cells_ref.py (the Python restatement the GPU-box tests use) is PINNED against the reference, and
//   * session-state golden files come from the reference's own writer (tests/golden/make_state_golden.py).
// Interfaces used: /root/reference/src/llama-kv-cache.h:63-178, llama-io.h:9-35, llama-batch.h:10-24.
#include "llama-batch.h"
#include "llama-cparams.h"
#include "llama-io.h"
#include "llama-kv-cache.h"
#include "llama-model.h"

#include "ggml-backend.h"

#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

namespace {
struct RefKv {
    llama_model   model;
    llama_cparams cparams{};
    std::unique_ptr<llama_kv_cache_unified> kv;
    RefKv() : model(llama_model_default_params()) {}
};
struct Writer : llama_io_write_i {
    std::vector<uint8_t> buf;
    void write(const void * src, size_t size) override { const uint8_t * p = (const uint8_t *) src; buf.insert(buf.end(), p, p + size); }
    void write_tensor(const ggml_tensor * t, size_t offset, size_t size) override {
        const size_t at = buf.size();
        buf.resize(at + size);
        ggml_backend_tensor_get(t, buf.data() + at, offset, size);
    }
    size_t n_bytes() override { return buf.size(); }
};
struct Reader : llama_io_read_i {
    const uint8_t * p; size_t n, at = 0;
    Reader(const void * src, size_t n_) : p((const uint8_t *) src), n(n_) {}
    const uint8_t * read(size_t size) override {
        if (at + size > n) throw std::runtime_error("unexpectedly reached end of buffer");
        const uint8_t * r = p + at; at += size; return r;
    }
    void read_to(void * dst, size_t size) override { memcpy(dst, read(size), size); }
    size_t n_bytes() override { return at; }
};
}  // namespace

extern "C" {

void * refkv_create(int n_layer, int n_head_kv, int head_dim, int type_k, int type_v, int kv_size, int flash_attn) {
    auto * r = new RefKv();
    r->model.arch = LLM_ARCH_LLAMA;
    llama_hparams & hp = r->model.hparams;
    hp.n_layer = n_layer;
    hp.n_embd_head_k = hp.n_embd_head_v = head_dim;
    for (int i = 0; i < n_layer; ++i) { hp.n_head_arr[i] = n_head_kv; hp.n_head_kv_arr[i] = n_head_kv; }
    r->cparams.flash_attn = flash_attn != 0;
    r->kv.reset(new llama_kv_cache_unified(hp, {/*get_rope_factors*/ nullptr}));
    if (!r->kv->init(r->model, r->cparams, (ggml_type) type_k, (ggml_type) type_v, (uint32_t) kv_size, /*offload*/ false)) { delete r; return nullptr; }
    return r;
}
void refkv_free(void * h) { delete (RefKv *) h; }

// llama_sbatch simple split (llama-batch.cpp:129-151): every token is its own "sequence" of one token
int refkv_find_slot(void * h, int n_tok, const int32_t * seq_id, const int32_t * pos) {
    auto & kv = *((RefKv *) h)->kv;
    std::vector<llama_pos>      p(pos, pos + n_tok);
    std::vector<int32_t>        nsi(n_tok, 1);
    std::vector<llama_seq_id>   ids(seq_id, seq_id + n_tok);
    std::vector<llama_seq_id *> pid(n_tok);
    for (int i = 0; i < n_tok; ++i) pid[i] = &ids[i];
    llama_ubatch ub{};
    ub.equal_seqs = false; ub.n_tokens = n_tok; ub.n_seq_tokens = 1; ub.n_seqs = n_tok;
    ub.pos = p.data(); ub.n_seq_id = nsi.data(); ub.seq_id = pid.data();
    const auto slot = kv.find_slot(ub);
    return slot ? (int) kv.head : -1;
}
// llama_context::decode after a ubatch (llama-context.cpp:1347-1352)
void refkv_advance_head(void * h, int n) {
    auto & kv = *((RefKv *) h)->kv;
    kv.head += n;
    if (kv.head >= kv.size) kv.head = 0;
}
void refkv_clear(void * h) { ((RefKv *) h)->kv->clear(); }
int  refkv_seq_rm(void * h, int s, int p0, int p1) { return ((RefKv *) h)->kv->seq_rm(s, p0, p1); }
void refkv_seq_cp(void * h, int a, int b, int p0, int p1) { ((RefKv *) h)->kv->seq_cp(a, b, p0, p1); }
void refkv_seq_keep(void * h, int s) { ((RefKv *) h)->kv->seq_keep(s); }
void refkv_seq_add(void * h, int s, int p0, int p1, int d) { ((RefKv *) h)->kv->seq_add(s, p0, p1, d); }
void refkv_seq_div(void * h, int s, int p0, int p1, int d) { ((RefKv *) h)->kv->seq_div(s, p0, p1, d); }
int  refkv_seq_pos_max(void * h, int s) { return ((RefKv *) h)->kv->seq_pos_max(s); }
int  refkv_n_tokens(void * h) { return ((RefKv *) h)->kv->get_n_tokens(); }
int  refkv_used_cells(void * h) { return (int) ((RefKv *) h)->kv->get_used_cells(); }
int  refkv_head(void * h) { return (int) ((RefKv *) h)->kv->head; }
int  refkv_has_shift(void * h) { return ((RefKv *) h)->kv->has_shift; }
int  refkv_v_trans(void * h) { return ((RefKv *) h)->kv->v_trans; }

// cell dump: pos[size], delta[size], n_seq[size], seqs[size * max_seq] (ascending ids, as std::set iterates)
void refkv_cells(void * h, int32_t * pos, int32_t * delta, int32_t * n_seq, int32_t * seqs, int max_seq) {
    auto & kv = *((RefKv *) h)->kv;
    for (uint32_t i = 0; i < kv.size; ++i) {
        pos[i] = kv.cells[i].pos; delta[i] = kv.cells[i].delta;
        int k = 0;
        for (auto id : kv.cells[i].seq_id) if (k < max_seq) seqs[(size_t) i * max_seq + k++] = id;
        n_seq[i] = (int32_t) kv.cells[i].seq_id.size();
    }
}
// llama_kv_cache_view_init + _update (llama-kv-cache.cpp:1338-1433): stats[5] = {n_cells, token_count, used_cells,
// max_contiguous, max_contiguous_idx}; pos[size]; seqs[size * n_seq_max]
void refkv_view(void * h, int n_seq_max, int32_t * stats, int32_t * pos, int32_t * seqs) {
    auto & kv = *((RefKv *) h)->kv;
    llama_kv_cache_view v = llama_kv_cache_view_init(kv, n_seq_max);
    llama_kv_cache_view_update(&v, &kv);
    stats[0] = v.n_cells; stats[1] = v.token_count; stats[2] = v.used_cells; stats[3] = v.max_contiguous; stats[4] = v.max_contiguous_idx;
    for (int i = 0; i < v.n_cells; ++i) {
        pos[i] = v.cells[i].pos;
        for (int j = 0; j < n_seq_max; ++j) seqs[(size_t) i * n_seq_max + j] = v.cells_sequences[(size_t) i * n_seq_max + j];
    }
    llama_kv_cache_view_free(&v);
}

// raw access to the flat per-layer tensors (byte offsets into k_l[il] / v_l[il])
void refkv_tensor_set(void * h, int il, int is_v, const void * src, size_t offset, size_t size) {
    auto & kv = *((RefKv *) h)->kv;
    ggml_backend_tensor_set(is_v ? kv.v_l[il] : kv.k_l[il], src, offset, size);
}
void refkv_tensor_get(void * h, int il, int is_v, void * dst, size_t offset, size_t size) {
    auto & kv = *((RefKv *) h)->kv;
    ggml_backend_tensor_get(is_v ? kv.v_l[il] : kv.k_l[il], dst, offset, size);
}
int64_t refkv_state_write(void * h, int seq_id, void * dst, size_t cap) {
    Writer w;
    ((RefKv *) h)->kv->state_write(w, seq_id);
    if (dst && w.buf.size() <= cap) memcpy(dst, w.buf.data(), w.buf.size());
    return (int64_t) w.buf.size();
}
int refkv_state_read(void * h, int seq_id, const void * src, size_t n) {
    Reader r(src, n);
    try { ((RefKv *) h)->kv->state_read(r, seq_id); } catch (const std::exception &) { return -1; }
    return 0;
}

}  // extern "C"
