// ggml_backend_mi355.cpp — ggml-backend plugin (GGML_BACKEND_API_VERSION 1) over the C ABI of libmi355_paged_kv.so.
//
// Boundary (b1) of SURVEY.md §8: exports `ggml_backend_init` / `ggml_backend_score`
// (/ref/ggml/src/ggml-backend-impl.h:215-251) so that the reference's registry loads it through GGML_BACKEND_PATH
// (/ref/ggml/src/ggml-backend-reg.cpp:581-585) and its own `test-backend-ops` harness can drive our kernels against
// the reference's CPU backend (/ref/tests/test-backend-ops.cpp:4574-4592, ggml_backend_compare_graph_backend).
// The object graph (reg -> device -> buffer type -> buffer / backend) is static and owned by the plugin.
//
// Compiled ONLY where the reference's headers exist (dev container): -I/root/reference/ggml/{include,src}; nothing
// of the reference is copied.  Tensors in our buffers hold the reference's own bytes (plain f32 / f16 / block_q8_0 /
// block_q4_0), exactly what ggml_backend_tensor_set/get move; ops convert to the private paged pool layout on the fly:
//   GGML_OP_FLASH_ATTN_EXT  K/V views -> pkv_pool_scatter_rows_strided (identity block table) -> pkv_attn_decode
//                           (1 query, no mask) or pkv_attn_prefill (general mask / ALiBi / softcap, non-causal); head sizes other than
//                           64 / 128, bf16 rows, odd gqa ratios -> pkv_flash_attn_ext_flat on the tensors as they lie
//   GGML_OP_ROPE(_BACK)     pkv_rope_f32_strided / pkv_rope_f16_strided (NORM / NEOX)
//   GGML_OP_CPY             pkv_convert_f32           (f32 -> f16 / q8_0 / q4_0, contiguous)
//   view ops                no-ops
// Everything else is reported unsupported, so the scheduler keeps it on another backend.
#include "ggml-backend-impl.h"
#include "ggml-backend.h"
#include "ggml.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/mi355_paged_kv.h"

#define MI355_LOG(...) fprintf(stderr, "ggml-mi355: " __VA_ARGS__)

namespace {

constexpr int BLOCK = 128;

// r02: a KV cache tensor of a flash-attention context IS its paged pool.  The tile layout of "Pool layout" (include/mi355_paged_kv.h) has the
// reference's bytes per token, so the first FLASH_ATTN_EXT over a cache tensor re-lays its memory out in place (one transient copy, once) and from
// then on the KV-store CPY writes tiles only, attention reads them with the identity block table (cell c = block c / 128, offset c % 128 — the
// placement is the reference runtime's: its own find_slot decides cells), the in-place K-shift RoPE runs on tiles (pkv_k_shift), and
// ggml_backend_tensor_set / _get on the tensor (session files) translate whole cell rows through pkv_pool_scatter_rows / _gather_rows.  KV memory 1x.
// Cache tensors whose cell count is not a multiple of 128, and GGML_MI355_KV_MIRROR=1, keep r01's separate mirror:
// Persistent paged mirror of a KV cache tensor that lives in this buffer (the reference's flat `cache_k_l%d` / `cache_v_l%d`,
// /ref/src/llama-kv-cache.cpp:100-103): the attention kernels read 128-token tiles, so instead of re-scattering all n_kv cells at
// every FLASH_ATTN_EXT the plugin keeps a pool beside the flat tensor and re-scatters only the byte ranges written since the
// last attention call (the KV-store CPY of the step; anything else that writes — set_tensor, clear, the K-shift RoPE — marks
// the whole tensor).  Identity block table: block b = cells [128 b, 128 b + 128).
struct kv_mirror {
    void * pool = nullptr;
    bool   in_place = false;   // the cache tensor's OWN memory holds the paged rows (r02): no flat copy exists, every access goes through the tile layout
    int    type = 0, is_v = 0, n_head_kv = 0, D = 0;
    int64_t n_cells = 0;
    std::vector<std::pair<size_t, size_t>> dirty;    // byte ranges [begin, end) of the flat tensor not yet in the pool
};
struct buffer_ctx {
    void * base;
    std::unordered_map<const void *, kv_mirror> mirrors;   // keyed by the flat tensor's data pointer
    std::unordered_map<const void *, size_t>    tracked;   // flat tensors we have seen as K/V roots -> their byte size
    // K-quant weight tensors of a WEIGHTS buffer whose rows are in the plane layout (csrc/pkv_wlayout.h): re-laid inside their own bytes by set_tensor, as the
    // reference's CPU backend repacks on set_tensor (/ref/ggml/src/ggml-cpu/ggml-cpu-aarch64.cpp:4145-4172); get_tensor applies the inverse, so the reference's bytes
    // round-trip.  Keyed by the tensor's data pointer; the products name such a tensor `type | PKV_TYPE_PACKED`.
    std::unordered_set<const void *>            packed;
    ~buffer_ctx() { for (auto & kv : mirrors) if (kv.second.pool && !kv.second.in_place) (void) hipFree(kv.second.pool); }
};
struct backend_ctx {
    hipStream_t stream = nullptr;
    void *      scratch = nullptr;
    size_t      scratch_size = 0;
    // results of products that were pulled forward although the graph's own output address was still occupied (Kcur / Vcur share
    // Qcur's address, see be_graph_compute): tensor -> where its values really are, for the ROPE / CPY that consume them
    void *      side = nullptr;
    size_t      side_size = 0;
    std::unordered_map<const ggml_tensor *, const char *> moved;
    const void * src_data(const ggml_tensor * t) const {
        if (moved.empty()) return t->data;
        size_t off = 0;
        const ggml_tensor * root = t;
        while (root->view_src) { off += root->view_offs; root = root->view_src; }
        auto it = moved.find(root);
        return it == moved.end() ? t->data : (const void *) (it->second + off);
    }
    void * side_buf(size_t bytes) {
        if (bytes > side_size) {
            (void) hipStreamSynchronize(stream);
            if (side) (void) hipFree(side);
            side = nullptr; side_size = 0;
            if (hipMalloc(&side, bytes * 2) != hipSuccess) return nullptr;
            side_size = bytes * 2;
        }
        return side;
    }
    void * arena(size_t bytes) {  // grow-only device scratch for the per-op pools / tables / workspace
        if (bytes > scratch_size) {
            (void) hipStreamSynchronize(stream);
            if (scratch) (void) hipFree(scratch);
            scratch = nullptr;
            scratch_size = 0;
            if (hipMalloc(&scratch, bytes + (bytes >> 2)) != hipSuccess) return nullptr;
            scratch_size = bytes + (bytes >> 2);
        }
        return scratch;
    }
};

bool kv_type(ggml_type t) { return t == GGML_TYPE_F16 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_0; }

// ---- buffer ------------------------------------------------------------------------------------------------------------
void buf_free(ggml_backend_buffer_t b) {
    auto * c = (buffer_ctx *) b->context;
    (void) hipFree(c->base);
    delete c;
}
// a write of [off, off + n) bytes into tensor t (or into the tensor it is a view of): remember it for the mirrors
void note_write(const ggml_tensor * t, size_t off, size_t n) {
    const ggml_tensor * root = t;
    while (root->view_src) { off += root->view_offs; root = root->view_src; }
    if (!root->buffer || root->buffer->iface.free_buffer != buf_free) return;
    auto * c = (buffer_ctx *) root->buffer->context;
    auto it = c->mirrors.find(root->data);
    if (it != c->mirrors.end()) it->second.dirty.emplace_back(off, off + n);
}
void note_write_all(const ggml_tensor * t) { note_write(t, 0, ggml_nbytes(t)); }
void * buf_base(ggml_backend_buffer_t b) { return ((buffer_ctx *) b->context)->base; }
// the in-place pool `t` (or the tensor it views) lives in, and t's byte offset inside that tensor
kv_mirror * in_place_of(const ggml_tensor * t, size_t * off_out = nullptr) {
    size_t off = 0;
    const ggml_tensor * root = t;
    while (root->view_src) { off += root->view_offs; root = root->view_src; }
    if (!root->buffer || root->buffer->iface.free_buffer != buf_free) return nullptr;
    auto * c = (buffer_ctx *) root->buffer->context;
    auto it = c->mirrors.find(root->data);
    if (it == c->mirrors.end() || !it->second.in_place) return nullptr;
    if (off_out) *off_out = off;
    return &it->second;
}
// bytes [off, off + n) of an in-place cache tensor, in the reference's flat row format <-> host memory.  The tile layout is addressed by whole cells: a range that
// starts or ends inside a cell is served by gathering the touched cells, patching / cutting the byte range on the host side of the staging rows, and (for a write)
// scattering them back.  The buffer interface has no error return: a failed transfer ABORTS (a dropped write or an uninitialised read would carry on as silent
// KV corruption — session restore, test-backend-ops' sentinels).
void in_place_rows(kv_mirror * m, size_t off, size_t n, void * host, bool to_pool) {
    if (n == 0) return;
    const size_t row_cell = ggml_row_size((ggml_type) m->type, m->D) * m->n_head_kv;
    const int64_t c0 = (int64_t) (off / row_cell), c1 = (int64_t) ((off + n + row_cell - 1) / row_cell), nc = c1 - c0;
    GGML_ASSERT(c1 <= m->n_cells && "access past the end of a paged cache tensor");
    const size_t span = (size_t) nc * row_cell, head = off - (size_t) c0 * row_cell;
    const bool whole = head == 0 && n == span;
    std::vector<int32_t> slot(2 * (size_t) nc);
    for (int64_t i = 0; i < nc; ++i) { slot[i] = (int32_t) ((c0 + i) / BLOCK); slot[nc + i] = (int32_t) ((c0 + i) % BLOCK); }
    void * rows = nullptr; int32_t * dslot = nullptr;
    bool ok = hipMalloc(&rows, span) == hipSuccess && hipMalloc((void **) &dslot, slot.size() * sizeof(int32_t)) == hipSuccess &&
              hipMemcpy(dslot, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && (!to_pool || !whole))      // a read, or the cells a partial write does not cover completely
        ok = pkv_pool_gather_rows(m->pool, m->type, m->is_v, rows, dslot, dslot + nc, (int) nc, m->n_head_kv, m->D, BLOCK, nullptr) == PKV_OK && hipDeviceSynchronize() == hipSuccess;
    if (ok && to_pool) ok = hipMemcpy((char *) rows + head, host, n, hipMemcpyHostToDevice) == hipSuccess &&
                            pkv_pool_scatter_rows(m->pool, m->type, m->is_v, rows, dslot, dslot + nc, (int) nc, m->n_head_kv, m->D, BLOCK, nullptr) == PKV_OK;
    if (ok && !to_pool) ok = hipMemcpy(host, (const char *) rows + head, n, hipMemcpyDeviceToHost) == hipSuccess;
    ok = hipDeviceSynchronize() == hipSuccess && ok;
    if (rows) (void) hipFree(rows);
    if (dslot) (void) hipFree(dslot);
    if (!ok) GGML_ABORT("mi355: transfer of %zu bytes at offset %zu %s a paged cache tensor failed: %s", n, off, to_pool ? "into" : "out of", pkv_last_error());
}
// ---- K-quant weights in the plane layout ------------------------------------------------------------------------------------
bool packable(const ggml_tensor * t) {
    static const bool off = getenv("GGML_MI355_NO_PACK") != nullptr;
    return !off && !t->view_src && t->buffer && t->buffer->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS && (t->type == GGML_TYPE_Q4_K || t->type == GGML_TYPE_Q5_K || t->type == GGML_TYPE_Q6_K) &&
           ggml_is_contiguous(t) && t->ne[0] % 256 == 0;
}
// the packed weight tensor `t` is (a row-aligned view of), or NULL
const ggml_tensor * packed_root(const ggml_tensor * t) {
    const ggml_tensor * root = t;
    while (root->view_src) root = root->view_src;
    if (!root->buffer || root->buffer->iface.free_buffer != buf_free) return nullptr;
    auto * c = (buffer_ctx *) root->buffer->context;
    return c->packed.count(root->data) ? root : nullptr;
}
int wtype(const ggml_tensor * w) { return (int) w->type | (packed_root(w) ? PKV_TYPE_PACKED : 0); }
// rows [r0, r1) of a packed tensor <-> GGUF-order bytes in a device staging buffer
void pack_rows(const ggml_tensor * t, int64_t r0, int64_t r1, void * staging, bool inverse) {
    const size_t rb = ggml_row_size(t->type, t->ne[0]);
    char * rows = (char *) t->data + (size_t) r0 * rb;
    const int rc = inverse ? pkv_weights_pack(rows, staging, (int) t->type, r1 - r0, t->ne[0], 1, nullptr) : pkv_weights_pack(staging, rows, (int) t->type, r1 - r0, t->ne[0], 0, nullptr);
    if (rc != PKV_OK || hipDeviceSynchronize() != hipSuccess) GGML_ABORT("mi355: re-laying the rows of weight tensor %s failed: %s", t->name, pkv_last_error());
}
// bytes [off, off + n) of a packable weight tensor, GGUF order on the host side
void packed_io(buffer_ctx * c, ggml_tensor * t, size_t off, size_t n, void * host, bool write) {
    if (n == 0) return;
    const size_t rb = ggml_row_size(t->type, t->ne[0]), total = ggml_nbytes(t);
    GGML_ASSERT(off + n <= total);
    (void) hipDeviceSynchronize();
    const bool known = c->packed.count(t->data) != 0;
    if (!known) {   // first contact: whatever the tensor holds (zeros of the fresh buffer, or a memset) becomes packed rows — unless this write replaces all of it anyway
        if (!(write && off == 0 && n == total)) {
            void * all = nullptr;
            if (hipMalloc(&all, total) != hipSuccess || hipMemcpy(all, t->data, total, hipMemcpyDeviceToDevice) != hipSuccess) GGML_ABORT("mi355: staging %zu bytes for weight tensor %s failed", total, t->name);
            pack_rows(t, 0, (int64_t) (total / rb), all, false);
            (void) hipFree(all);
        }
        c->packed.insert(t->data);
    }
    const int64_t r0 = (int64_t) (off / rb), r1 = (int64_t) ((off + n + rb - 1) / rb);
    const size_t span = (size_t) (r1 - r0) * rb, head = off - (size_t) r0 * rb;
    void * st = nullptr;
    if (hipMalloc(&st, span) != hipSuccess) GGML_ABORT("mi355: staging %zu bytes for weight tensor %s failed", span, t->name);
    const bool whole = head == 0 && n == span;
    if (!write || !whole) pack_rows(t, r0, r1, st, true);                    // the rows as the GGUF holds them
    bool ok = true;
    if (write) { ok = hipMemcpy((char *) st + head, host, n, hipMemcpyHostToDevice) == hipSuccess; if (ok) pack_rows(t, r0, r1, st, false); }
    else       ok = hipMemcpy(host, (const char *) st + head, n, hipMemcpyDeviceToHost) == hipSuccess;
    (void) hipFree(st);
    if (!ok) GGML_ABORT("mi355: transfer of %zu bytes %s weight tensor %s failed", n, write ? "into" : "out of", t->name);
}
void buf_memset(ggml_backend_buffer_t b, ggml_tensor * t, uint8_t v, size_t off, size_t n) {
    size_t toff = 0;
    if (kv_mirror * m = in_place_of(t, &toff)) {
        (void) hipDeviceSynchronize();
        if (toff + off == 0 && n == ggml_nbytes(t) && !t->view_src) {   // a uniform fill of the whole tensor does not depend on the layout
            if (hipMemset(t->data, v, n) != hipSuccess) GGML_ABORT("mi355: memset of a paged cache tensor failed");
            return;
        }
        std::vector<uint8_t> h(n, v);
        in_place_rows(m, toff + off, n, h.data(), true);
        return;
    }
    auto * c = (buffer_ctx *) b->context;
    if (c->packed.count(t->data) && !(off == 0 && n == ggml_nbytes(t))) {   // a partial fill of packed rows: through the GGUF-order view (a whole-tensor fill is layout-free)
        std::vector<uint8_t> h(n, v);
        packed_io(c, t, off, n, h.data(), true);
        return;
    }
    (void) hipMemset((char *) t->data + off, v, n);
    note_write(t, off, n);
}
void buf_set(ggml_backend_buffer_t b, ggml_tensor * t, const void * d, size_t off, size_t n) {
    size_t toff = 0;
    if (kv_mirror * m = in_place_of(t, &toff)) {
        (void) hipDeviceSynchronize();
        in_place_rows(m, toff + off, n, const_cast<void *>(d), true);
        return;
    }
    auto * c = (buffer_ctx *) b->context;
    if (packable(t) || c->packed.count(t->data)) { packed_io(c, t, off, n, const_cast<void *>(d), true); return; }
    if (const ggml_tensor * root = packed_root(t)) { (void) root; GGML_ABORT("mi355: set_tensor through a view of packed weight tensor %s is not supported", t->name); }
    if (hipMemcpy((char *) t->data + off, d, n, hipMemcpyHostToDevice) != hipSuccess) GGML_ABORT("mi355: set_tensor of %zu bytes into %s failed", n, t->name);
    note_write(t, off, n);
}
void buf_get(ggml_backend_buffer_t b, const ggml_tensor * t, void * d, size_t off, size_t n) {
    (void) hipDeviceSynchronize();
    size_t toff = 0;
    if (kv_mirror * m = in_place_of(t, &toff)) { in_place_rows(m, toff + off, n, d, false); return; }
    auto * c = (buffer_ctx *) b->context;
    if (c->packed.count(t->data) && !t->view_src) { packed_io(c, const_cast<ggml_tensor *>(t), off, n, d, false); return; }
    if (const ggml_tensor * root = packed_root(t)) {                          // a view of packed rows: read through the root at the view's byte offset
        size_t voff = 0;
        for (const ggml_tensor * x = t; x->view_src; x = x->view_src) voff += x->view_offs;
        packed_io(c, const_cast<ggml_tensor *>(root), voff + off, n, d, false);
        return;
    }
    if (hipMemcpy(d, (const char *) t->data + off, n, hipMemcpyDeviceToHost) != hipSuccess) GGML_ABORT("mi355: get_tensor of %zu bytes out of %s failed", n, t->name);
}
void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    auto * c = (buffer_ctx *) b->context;
    (void) hipMemset(c->base, v, b->size);                                      // (uniform bytes: the same in either row layout, packed tensors stay packed)
    for (auto & kv : c->mirrors) { kv.second.dirty.clear(); kv.second.dirty.emplace_back(0, (size_t) -1); }
}
// a tensor (re)allocated at an address: whatever the previous tenant of that address left in the maps is void.  (ggml-alloc hands the same compute-buffer
// address to different tensors graph after graph; cache and weight tensors are initialised once.)  Views are initialised too and share their root's address at
// offset 0: they are not tenants.
enum ggml_status buf_init_tensor(ggml_backend_buffer_t b, ggml_tensor * t) {
    if (t->view_src) return GGML_STATUS_SUCCESS;
    auto * c = (buffer_ctx *) b->context;
    auto it = c->mirrors.find(t->data);
    if (it != c->mirrors.end()) {
        if (it->second.pool && !it->second.in_place) (void) hipFree(it->second.pool);
        c->mirrors.erase(it);
    }
    c->packed.erase(t->data);
    return GGML_STATUS_SUCCESS;
}

const ggml_backend_buffer_i BUFFER_IFACE = {
    /* free_buffer   */ buf_free,
    /* get_base      */ buf_base,
    /* init_tensor   */ buf_init_tensor,
    /* memset_tensor */ buf_memset,
    /* set_tensor    */ buf_set,
    /* get_tensor    */ buf_get,
    /* cpy_tensor    */ nullptr,
    /* clear         */ buf_clear,
    /* reset         */ nullptr,
};

// ---- buffer type ---------------------------------------------------------------------------------------------------------
const char * buft_name(ggml_backend_buffer_type_t) { return "MI355"; }
ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    void * p = nullptr;
    if (hipMalloc(&p, size ? size : 1) != hipSuccess) { MI355_LOG("hipMalloc(%zu) failed\n", size); return nullptr; }
    (void) hipMemset(p, 0, size ? size : 1);
    auto * c = new buffer_ctx();
    c->base = p;
    return ggml_backend_buffer_init(buft, BUFFER_IFACE, c, size);
}
size_t buft_align(ggml_backend_buffer_type_t) { return 256; }

// ---- ops -------------------------------------------------------------------------------------------------------------------
// what every FLASH_ATTN_EXT of this device needs
bool fa_common_ok(const ggml_tensor * op) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2], * m = op->src[3];
    if (!q || !k || !v || q->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
    const int64_t D = q->ne[0];
    if (k->ne[0] != D || v->ne[0] != D) return false;
    if (q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return false;
    if (k->ne[2] <= 0 || q->ne[2] % k->ne[2] != 0 || v->ne[2] != k->ne[2] || v->ne[1] != k->ne[1]) return false;
    if (q->nb[0] != sizeof(float)) return false;
    if (k->nb[0] != ggml_type_size(k->type) || v->nb[0] != ggml_type_size(v->type)) return false;
    if (!ggml_is_contiguous(op)) return false;
    if (m && (m->type != GGML_TYPE_F16 || !ggml_is_contiguous(m) || m->ne[0] != k->ne[1] || m->ne[1] < q->ne[1])) return false;
    float max_bias;
    memcpy(&max_bias, (const float *) op->op_params + 1, sizeof(float));
    if (max_bias > 0.0f && !m) return false;
    return true;
}
// the paged kernels (tiles of a block pool): the north-star geometries
bool fa_paged_ok(const ggml_tensor * op) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2];
    if (!kv_type(k->type) || !kv_type(v->type)) return false;
    const int64_t D = q->ne[0];
    if (D != 64 && D != 128) return false;
    const int64_t g = q->ne[2] / k->ne[2];
    if (g != 1 && g != 2 && g != 4 && g != 8 && g != 16 && g != 32) return false;
    if (q->nb[1] % 16 || q->nb[2] % 16) return false;
    return true;
}
// pkv_flash_attn_ext_flat: the rest of the operator's domain, on the tensors as they lie (head sizes up to 256, bf16 rows, any gqa ratio)
bool flat_only_kv(ggml_type t) { return t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_IQ4_NL; }   // cache types that stay flat rows
bool fa_flat_type(ggml_type t) { return t == GGML_TYPE_F16 || t == GGML_TYPE_BF16 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_0 || flat_only_kv(t); }
bool fa_flat_ok(const ggml_tensor * op) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2];
    if (!fa_flat_type(k->type) || !fa_flat_type(v->type)) return false;
    const int64_t D = q->ne[0];
    if (D > 256 || D % 2) return false;
    if ((ggml_is_quantized(k->type) || ggml_is_quantized(v->type)) && D % 32) return false;
    if (q->nb[1] % 4 || q->nb[2] % 4 || k->nb[1] % 2 || k->nb[2] % 2 || v->nb[1] % 2 || v->nb[2] % 2) return false;
    return true;
}
bool supports_flash_attn(const ggml_tensor * op) { return fa_common_ok(op) && (fa_paged_ok(op) || fa_flat_ok(op)); }

// Returns the mirror pool of the flat tensor `t` views, brought up to date for cells [0, n_kv), or NULL when `t` is not such a
// view (test-backend-ops hands over free-standing K/V tensors: those take the scratch path).
char * mirror_sync(backend_ctx * bc, const ggml_tensor * t, int is_v, int n_kv, int n_head_kv, int D) {
    const ggml_tensor * root = t->view_src;
    if (!root || root->view_src || t->view_offs != 0 || !root->buffer || root->buffer->iface.free_buffer != buf_free) return nullptr;
    const size_t row_head = ggml_row_size(t->type, D), row_cell = row_head * n_head_kv;
    if (t->nb[1] != row_cell || t->nb[2] != row_head || !ggml_is_contiguous(root) || ggml_nbytes(root) % row_cell != 0) return nullptr;
    auto * c = (buffer_ctx *) root->buffer->context;
    kv_mirror & m = c->mirrors[root->data];
    const int64_t n_cells = (int64_t) (ggml_nbytes(root) / row_cell);
    if (m.in_place) {
        if (m.type != (int) t->type || m.n_head_kv != n_head_kv || m.D != D || m.is_v != is_v) { MI355_LOG("a paged cache tensor is read with another geometry than it was laid out for\n"); return nullptr; }
        return (char *) m.pool;
    }
    if (!m.pool || m.type != (int) t->type || m.n_head_kv != n_head_kv || m.D != D || m.n_cells != n_cells) {
        if (m.pool) (void) hipFree(m.pool);
        m = kv_mirror();
        const int64_t nb = (n_cells + BLOCK - 1) / BLOCK;
        const size_t bytes = pkv_pool_bytes(t->type, (int) nb, n_head_kv, BLOCK, D);
        static const bool keep_mirror = getenv("GGML_MI355_KV_MIRROR") != nullptr;
        // ... never a tensor of a graph-allocated compute buffer: ggml-alloc gives that address to another tensor in the next graph, which would then be read as
        // tiles; test-backend-ops' free-standing K / V in such buffers keep the mirror path with its dirty ranges
        if (!keep_mirror && n_cells % BLOCK == 0 && bytes == ggml_nbytes(root) && ggml_backend_buffer_get_usage(root->buffer) != GGML_BACKEND_BUFFER_USAGE_COMPUTE) {
            // the tensor becomes its own pool: tiles built beside it once, copied over its flat rows, the transient freed
            void * tmp = nullptr;
            if (hipMalloc(&tmp, bytes) != hipSuccess) return nullptr;
            int rc = hipMemsetAsync(tmp, 0, bytes, bc->stream) == hipSuccess ? 0 : -1;
            if (!rc) rc = pkv_pool_scatter_rows_strided(tmp, t->type, is_v, root->data, (int64_t) row_cell, (int64_t) row_head, nullptr, nullptr, (int) n_cells, n_head_kv, D, BLOCK, bc->stream);
            if (!rc) rc = hipMemcpyAsync(root->data, tmp, bytes, hipMemcpyDeviceToDevice, bc->stream) == hipSuccess ? 0 : -1;
            (void) hipStreamSynchronize(bc->stream);
            (void) hipFree(tmp);
            if (rc) return nullptr;
            m.pool = root->data; m.in_place = true;
            m.type = t->type; m.is_v = is_v; m.n_head_kv = n_head_kv; m.D = D; m.n_cells = n_cells;
            return (char *) m.pool;
        }
        if (hipMalloc(&m.pool, bytes) != hipSuccess) { m.pool = nullptr; return nullptr; }
        (void) hipMemsetAsync(m.pool, 0, bytes, bc->stream);
        m.type = t->type; m.is_v = is_v; m.n_head_kv = n_head_kv; m.D = D; m.n_cells = n_cells;
        m.dirty.emplace_back(0, (size_t) -1);
    }
    // bring the dirty byte ranges into the pool: whole blocks (identity placement starts at a block boundary)
    const size_t tile_block = pkv_pool_bytes(t->type, 1, n_head_kv, BLOCK, D);
    for (const auto & r : m.dirty) {
        int64_t c0 = (int64_t) (r.first / row_cell), c1 = r.second == (size_t) -1 ? n_cells : (int64_t) ((r.second + row_cell - 1) / row_cell);
        c1 = std::min<int64_t>(c1, std::min<int64_t>(n_cells, ((int64_t) n_kv + BLOCK - 1) / BLOCK * BLOCK));   // cells beyond n_kv stay dirty-free: they are re-marked below
        c0 = c0 / BLOCK * BLOCK;
        if (c1 <= c0) continue;
        const int rc = pkv_pool_scatter_rows_strided((char *) m.pool + (size_t) (c0 / BLOCK) * tile_block, t->type, is_v,
                                                     (const char *) root->data + (size_t) c0 * row_cell, (int64_t) row_cell, (int64_t) row_head,
                                                     nullptr, nullptr, (int) (c1 - c0), n_head_kv, D, BLOCK, bc->stream);
        if (rc) return nullptr;
    }
    // ranges that reached past the cells this call sees must be kept for a later, longer call
    std::vector<std::pair<size_t, size_t>> keep;
    const size_t seen = (size_t) std::min<int64_t>(n_cells, ((int64_t) n_kv + BLOCK - 1) / BLOCK * BLOCK) * row_cell;
    for (const auto & r : m.dirty) if (r.second > seen) keep.emplace_back(std::max(r.first, seen), r.second);
    m.dirty.swap(keep);
    return (char *) m.pool;
}

ggml_status run_flash_attn(backend_ctx * bc, ggml_tensor * op) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2], * m = op->src[3];
    float scale, max_bias, softcap;
    memcpy(&scale, (const float *) op->op_params + 0, sizeof(float));
    memcpy(&max_bias, (const float *) op->op_params + 1, sizeof(float));
    memcpy(&softcap, (const float *) op->op_params + 2, sizeof(float));
    const int D = (int) q->ne[0], n_q = (int) q->ne[1], n_head = (int) q->ne[2], n_kv = (int) k->ne[1], n_head_kv = (int) k->ne[2];
    if (!fa_paged_ok(op)) {   // outside the paged kernels' geometries: the flat kernel reads the tensors where they lie
        // (a cache tensor that an earlier paged call re-laid as tiles has no flat rows any more: one context never mixes the two kinds of node, stop if one does)
        if (in_place_of(k) || in_place_of(v)) GGML_ABORT("mi355: flash_attn_ext outside the paged geometries over a cache tensor that is laid out as a block pool (%s)", k->name);
        pkv_flat_attn_args f = {};
        f.q = (const float *) q->data; f.q_stride_tok = (int64_t) (q->nb[1] / sizeof(float)); f.q_stride_head = (int64_t) (q->nb[2] / sizeof(float));
        f.k = k->data; f.k_stride_tok = (int64_t) k->nb[1]; f.k_stride_head = (int64_t) k->nb[2];
        f.v = v->data; f.v_stride_tok = (int64_t) v->nb[1]; f.v_stride_head = (int64_t) v->nb[2];
        f.mask = m ? m->data : nullptr; f.mask_stride = m ? (int64_t) m->ne[0] : 0;
        f.out = (float *) op->data;
        f.n_q = n_q; f.n_kv = n_kv; f.n_head = n_head; f.n_head_kv = n_head_kv; f.head_dim = D; f.type_k = k->type; f.type_v = v->type;
        f.scale = scale; f.max_bias = max_bias; f.logit_softcap = softcap;
        f.workspace_bytes = pkv_flash_attn_ext_flat_workspace_bytes(n_q, n_head, D, n_kv);
        f.workspace = f.workspace_bytes ? bc->arena(f.workspace_bytes) : nullptr;   // (no arena: the launch runs unsplit)
        if (!f.workspace) f.workspace_bytes = 0;
        if (pkv_flash_attn_ext_flat(&f, bc->stream)) { MI355_LOG("flash_attn_ext (flat) failed: %s\n", pkv_last_error()); return GGML_STATUS_FAILED; }
        return GGML_STATUS_SUCCESS;
    }
    const int n_blocks = (n_kv + BLOCK - 1) / BLOCK;
    // K/V of a llama graph are views of the flat cache tensors (cell-major rows of n_head_kv * D elements, first cell 0):
    // serve them from the persistent mirrors
    char * mk = mirror_sync(bc, k, 0, n_kv, n_head_kv, D), * mv = mirror_sync(bc, v, 1, n_kv, n_head_kv, D);
    const bool mirrored = mk && mv;
    const size_t kb = mirrored ? 0 : pkv_pool_bytes(k->type, n_blocks, n_head_kv, BLOCK, D), vb = mirrored ? 0 : pkv_pool_bytes(v->type, n_blocks, n_head_kv, BLOCK, D);
    const size_t wsb = pkv_attn_workspace_bytes(n_q, n_head, D, n_kv, BLOCK);
    const size_t meta = ((size_t) n_blocks + 4) * sizeof(int32_t);
    auto up = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    char * base = (char *) bc->arena(up(kb) + up(vb) + up(wsb) + up(meta));
    if (!base) return GGML_STATUS_ALLOC_FAILED;
    char * kp = base, * vp = kp + up(kb), * ws = vp + up(vb);
    int32_t * tab = (int32_t *) (ws + up(wsb));
    std::vector<int32_t> h(n_blocks + 3);
    for (int i = 0; i < n_blocks; ++i) h[i] = i;            // identity block table
    h[n_blocks] = n_kv; h[n_blocks + 1] = 0; h[n_blocks + 2] = n_q;  // seq_lens[1], q_start[2]
    if (hipMemcpyAsync(tab, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, bc->stream) != hipSuccess) return GGML_STATUS_FAILED;
    (void) hipStreamSynchronize(bc->stream);               // h is a stack-frame object
    int rc = 0;
    if (mirrored) {
        kp = mk; vp = mv;
    } else {
        // the pools are private scratch: clear them so that the tail of the last block is finite
        (void) hipMemsetAsync(kp, 0, up(kb) + up(vb), bc->stream);
        rc = pkv_pool_scatter_rows_strided(kp, k->type, 0, k->data, (int64_t) k->nb[1], (int64_t) k->nb[2], nullptr, nullptr, n_kv,
                                           n_head_kv, D, BLOCK, bc->stream);
        if (!rc) rc = pkv_pool_scatter_rows_strided(vp, v->type, 1, v->data, (int64_t) v->nb[1], (int64_t) v->nb[2], nullptr, nullptr,
                                                    n_kv, n_head_kv, D, BLOCK, bc->stream);
    }
    pkv_attn_args a = {};
    a.q = (const float *) q->data; a.q_stride_tok = (int64_t) (q->nb[1] / sizeof(float)); a.q_stride_head = (int64_t) (q->nb[2] / sizeof(float));
    a.out = (float *) op->data; a.k_pool = kp; a.v_pool = vp; a.block_table = tab; a.seq_lens = tab + n_blocks;
    a.workspace = ws; a.workspace_bytes = wsb;
    a.n_seq = 1; a.n_tok = n_q; a.max_q_per_seq = n_q; a.n_head = n_head; a.n_head_kv = n_head_kv; a.head_dim = D;
    a.block_size = BLOCK; a.max_blocks = n_blocks; a.type_k = k->type; a.type_v = v->type;
    a.scale = scale; a.logit_softcap = softcap; a.max_seq_len = n_kv;
    if (!rc) {
        if (n_q == 1 && (!m || max_bias == 0.0f)) {
            // the split-KV decode kernel; the reference's decode step always carries a KQ_mask row (0 / -inf per cell)
            a.mask = m ? m->data : nullptr; a.mask_stride = m ? (int64_t) m->ne[0] : 0;
            rc = pkv_attn_decode(&a, bc->stream);
        } else {
            a.q_start = tab + n_blocks + 1;
            a.flags = PKV_ATTN_NONCAUSAL;
            a.mask = m ? m->data : nullptr; a.mask_stride = m ? (int64_t) m->ne[0] : 0; a.max_bias = max_bias;
            rc = pkv_attn_prefill(&a, bc->stream);
        }
    }
    if (rc) { MI355_LOG("flash_attn_ext failed: %s\n", pkv_last_error()); return GGML_STATUS_FAILED; }
    return GGML_STATUS_SUCCESS;
}

bool rope_params(const ggml_tensor * op, pkv_rope_params * rp) {
    const int32_t * p = (const int32_t *) op->op_params;
    rp->n_dims = p[1]; rp->mode = p[2]; rp->n_ctx_orig = p[4];
    memcpy(&rp->freq_base, p + 5, 4);  memcpy(&rp->freq_scale, p + 6, 4);  memcpy(&rp->ext_factor, p + 7, 4);
    memcpy(&rp->attn_factor, p + 8, 4); memcpy(&rp->beta_fast, p + 9, 4); memcpy(&rp->beta_slow, p + 10, 4);
    rp->freq_factors = op->src[2] ? (const float *) op->src[2]->data : nullptr;
    return rp->mode == 0 || rp->mode == GGML_ROPE_TYPE_NEOX;
}
bool supports_rope(const ggml_tensor * op) {
    const ggml_tensor * x = op->src[0];
    pkv_rope_params rp;
    if (!x || (x->type != GGML_TYPE_F32 && x->type != GGML_TYPE_F16) || op->type != x->type || !rope_params(op, &rp)) return false;
    if (x->ne[3] != 1 || x->nb[0] != ggml_type_size(x->type) || !ggml_is_contiguous(op) || x->ne[0] % 2) return false;
    if (op->src[2] && (op->src[2]->type != GGML_TYPE_F32 || op->src[2]->ne[0] < rp.n_dims / 2)) return false;
    return rp.n_dims > 0 && rp.n_dims % 2 == 0 && rp.n_dims <= x->ne[0];
}
ggml_status run_rope(backend_ctx * bc, ggml_tensor * op, bool back) {
    const ggml_tensor * x = op->src[0];
    pkv_rope_params rp;
    rope_params(op, &rp);
    size_t toff = 0;
    if (kv_mirror * m = in_place_of(x, &toff)) {
        // the K-shift of kv_self_update (/ref/src/llama-context.cpp:447-496): ggml_rope_ext_inplace over a [D, n_head_kv, n_cells] view of the K cache,
        // src[1] = the per-cell position deltas.  The cache is its pool: rotate the tile rows in place (pkv_k_shift), cells = the view's
        const size_t row_cell = ggml_row_size((ggml_type) m->type, m->D) * m->n_head_kv;
        const int64_t nc = x->ne[2];
        if (back || m->is_v || op->data != x->data || x->ne[0] != m->D || x->ne[1] != m->n_head_kv || toff % row_cell != 0 || x->nb[2] != row_cell) {
            MI355_LOG("rope over a paged cache tensor: only the in-place K-shift of whole cells is supported\n");
            return GGML_STATUS_FAILED;
        }
        const int64_t c0 = (int64_t) (toff / row_cell);
        std::vector<int32_t> h(2 * (size_t) nc);
        for (int64_t i = 0; i < nc; ++i) { h[i] = (int32_t) ((c0 + i) / BLOCK); h[nc + i] = (int32_t) ((c0 + i) % BLOCK); }
        int32_t * d = (int32_t *) bc->arena(h.size() * sizeof(int32_t));
        if (!d || hipMemcpyAsync(d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, bc->stream) != hipSuccess) return GGML_STATUS_ALLOC_FAILED;
        (void) hipStreamSynchronize(bc->stream);
        if (pkv_k_shift(m->pool, m->type, d, d + nc, (const int32_t *) op->src[1]->data, (int) nc, m->n_head_kv, m->D, BLOCK, &rp, bc->stream) != PKV_OK) {
            MI355_LOG("k-shift on the paged cache failed: %s\n", pkv_last_error());
            return GGML_STATUS_FAILED;
        }
        return GGML_STATUS_SUCCESS;
    }
    const int64_t es = (int64_t) ggml_type_size(x->type);
    const int rc = x->type == GGML_TYPE_F32
        ? pkv_rope_f32_strided((const float *) bc->src_data(x), (int64_t) x->nb[2] / es, (int64_t) x->nb[1] / es, (const int32_t *) op->src[1]->data,
                               (float *) op->data, (int) x->ne[2], (int) x->ne[1], (int) x->ne[0], &rp, back ? 1 : 0, bc->stream)
        : pkv_rope_f16_strided(bc->src_data(x), (int64_t) x->nb[2] / es, (int64_t) x->nb[1] / es, (const int32_t *) op->src[1]->data, op->data,
                               (int) x->ne[2], (int) x->ne[1], (int) x->ne[0], &rp, back ? 1 : 0, bc->stream);
    if (rc) { MI355_LOG("rope failed: %s\n", pkv_last_error()); return GGML_STATUS_FAILED; }
    note_write_all(op);                                     // in-place K-shift on the cache (/ref/src/llama-context.cpp:447-496)
    return GGML_STATUS_SUCCESS;
}

bool plain_type(ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16; }
bool cpy_type(ggml_type t) { return plain_type(t) || t == GGML_TYPE_BF16; }   // (bf16: the rows of a `-ctk / -ctv bf16` cache, read by pkv_flash_attn_ext_flat)
// ggml_cpy / ggml_cont between strided f32 / f16 tensors (the V store into the transposed v cache of a -fa 0 graph, /ref/src/llama-graph.cpp:1387-1394;
// kqv_merged -> cont, :1232-1234): the generic element-by-element copy
bool supports_cpy_strided(const ggml_tensor * s, const ggml_tensor * d) {
    return s && d && cpy_type(s->type) && cpy_type(d->type) && ggml_nelements(s) == ggml_nelements(d);
}
bool supports_cpy(const ggml_tensor * op) {
    const ggml_tensor * s = op->src[0], * d = op->src[1];
    // the KV store of build_attn copies k_cur [head_dim, n_head_kv, n_tokens] into a 1-D view of the cache
    // (/ref/src/llama-graph.cpp:1375-1397): shapes differ, the element order does not; rows of 64/128 keep quant blocks whole
    if (s && d && s->type == GGML_TYPE_F32 && kv_type(d->type) && ggml_is_contiguous(s) && ggml_is_contiguous(d) &&
        ggml_nelements(s) == ggml_nelements(d) && s->ne[0] % 64 == 0) return true;
    // the KV store of a q4_1 / q5_0 / q5_1 / iq4_nl cache (flat rows, read by pkv_flash_attn_ext_flat): the reference's *_ref quantizers, block by block
    if (s && d && s->type == GGML_TYPE_F32 && flat_only_kv(d->type) && ggml_is_contiguous(s) && ggml_is_contiguous(d) && ggml_nelements(s) == ggml_nelements(d) &&
        s->ne[0] % 32 == 0) return true;
    // defrag moves of a quantized cache (build_kv_self_defrag, /ref/src/llama-context.cpp:668-711): same type, same shape, contiguous row ranges — cells of a
    // tensor that is its own pool (run_cpy_cells) or flat bytes (a staged byte copy); f16 moves also pass the plain-type rule below
    if (s && d && s->type == d->type && (kv_type(s->type) || flat_only_kv(s->type)) && ggml_are_same_shape(s, d) && ggml_is_contiguous(s) && ggml_is_contiguous(d)) return true;
    return supports_cpy_strided(s, d);
}
bool supports_cont(const ggml_tensor * op) { return supports_cpy_strided(op->src[0], op) && op->src[0]->type == op->type; }
// the paged mirror a KV-store CPY can write directly: dst is a view into a cache tensor of this buffer type whose mirror exists, the
// copy covers whole cells [*cell0, *cell0 + *n_cells) of it
const kv_mirror * store_mirror(const ggml_tensor * cpy, int64_t * cell0, int64_t * n_cells) {
    const ggml_tensor * s = cpy->src[0], * d = cpy->src[1], * root = d;
    size_t off = 0;
    while (root->view_src) { off += root->view_offs; root = root->view_src; }
    if (root == d || !root->buffer || root->buffer->iface.free_buffer != buf_free) return nullptr;
    auto * c = (buffer_ctx *) root->buffer->context;
    auto it = c->mirrors.find(root->data);
    if (it == c->mirrors.end() || !it->second.pool || it->second.type != (int) d->type) return nullptr;
    const kv_mirror & m = it->second;
    const size_t row_cell = ggml_row_size(d->type, m.D) * m.n_head_kv;
    const int64_t per_cell = (int64_t) m.n_head_kv * m.D, n_el = ggml_nelements(s);
    if (off % row_cell != 0 || n_el % per_cell != 0 || s->ne[0] % m.D != 0 || (int64_t) (off / row_cell) + n_el / per_cell > m.n_cells) return nullptr;
    *cell0 = (int64_t) (off / row_cell);
    *n_cells = n_el / per_cell;
    return &m;
}
// whole cells [*c0, *c0 + *nc) of the in-place pool that `t` views as contiguous rows, or NULL
kv_mirror * in_place_cells(const ggml_tensor * t, int64_t * c0, int64_t * nc) {
    size_t off = 0;
    kv_mirror * m = in_place_of(t, &off);
    if (!m || (int) t->type != m->type || !ggml_is_contiguous(t)) return nullptr;
    const size_t row_cell = ggml_row_size((ggml_type) m->type, m->D) * m->n_head_kv, nb = ggml_nbytes(t);
    if (off % row_cell != 0 || nb % row_cell != 0 || (int64_t) ((off + nb) / row_cell) > m->n_cells) return nullptr;
    *c0 = (int64_t) (off / row_cell); *nc = (int64_t) (nb / row_cell);
    return m;
}
// ggml_cpy between cell ranges of ONE paged cache tensor: the moves of build_kv_self_defrag (/ref/src/llama-context.cpp:668-711: view_2d of nm cell rows at cell i ->
// the same at cell id).  The tile layout is addressed by cells, so the copy is a gather of the source cells into flat staging rows and a scatter into the
// destination cells; staging because the two ranges may overlap.  r02 returned FAILED here; kv_self_update ignores graph_compute's status (:773) after defrag_prepare
// has already moved the cell metadata — the data stayed put under metadata that pointed elsewhere.
ggml_status run_cpy_cells(backend_ctx * bc, kv_mirror * m, int64_t cs, int64_t cd, int64_t nc) {
    if (nc == 0 || cs == cd) return GGML_STATUS_SUCCESS;
    const size_t row_cell = ggml_row_size((ggml_type) m->type, m->D) * m->n_head_kv;
    std::vector<int32_t> h(4 * (size_t) nc);
    for (int64_t i = 0; i < nc; ++i) {
        h[i] = (int32_t) ((cs + i) / BLOCK); h[nc + i] = (int32_t) ((cs + i) % BLOCK);
        h[2 * nc + i] = (int32_t) ((cd + i) / BLOCK); h[3 * nc + i] = (int32_t) ((cd + i) % BLOCK);
    }
    auto up = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    char * base = (char *) bc->arena(up((size_t) nc * row_cell) + up(h.size() * sizeof(int32_t)));
    if (!base) return GGML_STATUS_ALLOC_FAILED;
    int32_t * dt = (int32_t *) (base + up((size_t) nc * row_cell));
    bool ok = hipMemcpyAsync(dt, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, bc->stream) == hipSuccess;
    (void) hipStreamSynchronize(bc->stream);               // h is a local
    ok = ok && pkv_pool_gather_rows(m->pool, m->type, m->is_v, base, dt, dt + nc, (int) nc, m->n_head_kv, m->D, BLOCK, bc->stream) == PKV_OK &&
               pkv_pool_scatter_rows(m->pool, m->type, m->is_v, base, dt + 2 * nc, dt + 3 * nc, (int) nc, m->n_head_kv, m->D, BLOCK, bc->stream) == PKV_OK;
    // no error can reach the caller that matters (kv_self_update drops the status): a failed move of KV cells must not pass silently
    if (!ok) GGML_ABORT("mi355: moving %lld cells of a paged cache tensor failed: %s", (long long) nc, pkv_last_error());
    return GGML_STATUS_SUCCESS;
}
ggml_status run_cpy_strided(backend_ctx * bc, const ggml_tensor * s, ggml_tensor * d) {
    int64_t nbs[4], nbd[4];
    for (int i = 0; i < 4; ++i) { nbs[i] = (int64_t) s->nb[i]; nbd[i] = (int64_t) d->nb[i]; }
    if (in_place_of(d) || in_place_of(s)) {
        int64_t cs, ns, cd, nd;
        kv_mirror * ms = in_place_cells(s, &cs, &ns), * md = in_place_cells(d, &cd, &nd);
        if (ms && ms == md && ns == nd) return run_cpy_cells(bc, ms, cs, cd, ns);
        // anything else over a paged cache tensor (a -fa 0 op on a flash-attention cache, a copy between two tensors) has no meaning in the tile layout: stop,
        // do not carry on with metadata and data out of step
        GGML_ABORT("mi355: copy %s -> %s over a paged cache tensor that is not a move of whole cells inside one tensor", s->name, d->name);
    }
    if (s->type == d->type && !plain_type(s->type) && ggml_is_contiguous(s) && ggml_is_contiguous(d)) {   // quantized rows of a flat tensor: bytes, staged (ranges may overlap)
        const size_t nb = ggml_nbytes(s);
        void * st = bc->arena(nb);
        if (!st) return GGML_STATUS_ALLOC_FAILED;
        if (hipMemcpyAsync(st, bc->src_data(s), nb, hipMemcpyDeviceToDevice, bc->stream) != hipSuccess || hipMemcpyAsync(d->data, st, nb, hipMemcpyDeviceToDevice, bc->stream) != hipSuccess)
            GGML_ABORT("mi355: byte copy %s -> %s failed", s->name, d->name);
        note_write_all(d);
        return GGML_STATUS_SUCCESS;
    }
    const int rc = pkv_cpy_strided(bc->src_data(s), s->type, s->ne, nbs, d->data, d->type, d->ne, nbd, bc->stream);
    if (rc) { MI355_LOG("strided cpy failed: %s\n", pkv_last_error()); return GGML_STATUS_FAILED; }
    note_write_all(d);
    return GGML_STATUS_SUCCESS;
}
ggml_status run_cpy(backend_ctx * bc, ggml_tensor * op) {
    const ggml_tensor * s = op->src[0];
    ggml_tensor * d = op->src[1];
    if (s->type == GGML_TYPE_F32 && flat_only_kv(d->type)) {
        const int rc = pkv_convert_f32((const float *) bc->src_data(s), d->data, d->type, ggml_nrows(s), s->ne[0], bc->stream);
        if (rc) { MI355_LOG("cpy failed: %s\n", pkv_last_error()); return GGML_STATUS_FAILED; }
        note_write_all(d);
        return GGML_STATUS_SUCCESS;
    }
    if (!(s->type == GGML_TYPE_F32 && kv_type(d->type) && ggml_is_contiguous(s) && ggml_is_contiguous(d) && s->ne[0] % 64 == 0)) return run_cpy_strided(bc, s, d);
    // the KV store of a step into a cache tensor that already has its paged mirror: flat bytes and mirror rows in ONE launch (same
    // quantized bits), nothing left to re-scatter at the next attention call
    int64_t cell0, n_cells;
    if (const kv_mirror * m = store_mirror(op, &cell0, &n_cells)) {
        if (pkv_convert_f32_paged((const float *) bc->src_data(s), m->in_place ? nullptr : d->data, d->type, m->pool, m->is_v, cell0, n_cells, m->n_head_kv, m->D, BLOCK, bc->stream) == PKV_OK)
            return GGML_STATUS_SUCCESS;
    }
    if (in_place_of(d)) { MI355_LOG("KV store into a paged cache tensor that does not cover whole cells\n"); return GGML_STATUS_FAILED; }
    const int rc = pkv_convert_f32((const float *) bc->src_data(s), d->data, d->type, ggml_nrows(s), s->ne[0], bc->stream);
    if (rc) { MI355_LOG("cpy failed: %s\n", pkv_last_error()); return GGML_STATUS_FAILED; }
    note_write_all(d);                                      // the KV store of the step: the next attention re-scatters just these cells
    return GGML_STATUS_SUCCESS;
}

// ---- first slice of the rest of the decoder layer (include/mi355_paged_kv.h: pkv_rms_norm_f32 ... pkv_mul_mat) --------------------------
bool f32c(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && ggml_is_contiguous(t); }
bool supports_rms_norm(const ggml_tensor * op) { return f32c(op->src[0]) && f32c(op) && op->src[0]->ne[0] > 0; }
// same shape, or src1 = a block of rows that repeats over src0's rows (norm weight [ne0], residual [ne0, n]); no other broadcast
bool supports_binary(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    if (!f32c(a) || !f32c(b) || !f32c(op) || a->ne[0] != b->ne[0] || !ggml_are_same_shape(a, op)) return false;
    const int64_t ra = ggml_nrows(a), rb = ggml_nrows(b);
    if (rb <= 0 || ra % rb != 0) return false;
    // rows of b must tile rows of a in order: b's higher dims are either equal to a's or 1 from some dim upward
    bool tail = false;
    for (int d = 1; d < 4; ++d) {
        if (b->ne[d] == a->ne[d] && !tail) continue;
        if (b->ne[d] == 1) { tail = true; continue; }
        return false;
    }
    return true;
}
// ggml_soft_max_ext: f32 contiguous logits, optional f32 / f16 contiguous mask of >= ne01 rows of ne00 (the KQ_mask of a -fa 0 graph)
bool supports_soft_max(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * m = op->src[1];
    if (!f32c(a) || !f32c(op) || !ggml_are_same_shape(a, op)) return false;
    if (m && ((m->type != GGML_TYPE_F32 && m->type != GGML_TYPE_F16) || !ggml_is_contiguous(m) || m->ne[0] != a->ne[0] || m->ne[1] < a->ne[1])) return false;
    return true;
}
bool supports_silu(const ggml_tensor * op) { return ggml_get_unary_op(op) == GGML_UNARY_OP_SILU && f32c(op->src[0]) && f32c(op); }
bool mul_mat_2d(const ggml_tensor * op) {
    const ggml_tensor * w = op->src[0], * x = op->src[1];
    if (!w || !f32c(x) || !f32c(op) || !ggml_is_contiguous(w)) return false;
    if (w->type != GGML_TYPE_F32 && w->type != GGML_TYPE_F16 && w->type != GGML_TYPE_Q8_0 && w->type != GGML_TYPE_Q4_K && w->type != GGML_TYPE_Q5_K && w->type != GGML_TYPE_Q6_K) return false;
    if (w->ne[2] != 1 || w->ne[3] != 1 || x->ne[2] != 1 || x->ne[3] != 1) return false;   // plain 2-D products only
    if (w->type == GGML_TYPE_Q6_K && (w->ne[0] % 256 != 0 || (x->ne[1] == 1 && w->ne[1] % 2 != 0))) return false;   // q6_K: the decode product (row pairs) or the MFMA product
    return w->ne[0] == x->ne[0] && w->ne[0] % (w->type == GGML_TYPE_F32 ? 4 : (w->type == GGML_TYPE_F16 ? 8 : ((w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K) ? 256 : 32))) == 0;
}
// kq = mul_mat(k, q) and kqv = mul_mat(v, kq) of a -fa 0 graph: f16 views of the cache (permuted / transposed, broadcast over the query heads of a kv
// head) against permuted f32 activations (/ref/src/llama-graph.cpp:1201-1230)
bool mul_mat_strided(const ggml_tensor * op) {
    const ggml_tensor * w = op->src[0], * x = op->src[1];
    if (!w || !x || !plain_type(w->type) || x->type != GGML_TYPE_F32 || !f32c(op)) return false;
    if (w->nb[0] != ggml_type_size(w->type) || x->nb[0] != sizeof(float) || w->ne[0] != x->ne[0]) return false;
    if (w->ne[2] <= 0 || w->ne[3] <= 0 || x->ne[2] % w->ne[2] != 0 || x->ne[3] % w->ne[3] != 0 || x->ne[2] * x->ne[3] > 65535) return false;
    return op->ne[0] == w->ne[1] && op->ne[1] == x->ne[1] && op->ne[2] == x->ne[2] && op->ne[3] == x->ne[3];
}
bool supports_mul_mat(const ggml_tensor * op) { return mul_mat_2d(op) || mul_mat_strided(op); }
ggml_status run_layer_op(backend_ctx * bc, ggml_tensor * op) {
    int rc = PKV_ERR_UNSUPPORTED;
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    switch (op->op) {
        case GGML_OP_RMS_NORM: {
            float eps; memcpy(&eps, op->op_params, sizeof(float));
            rc = pkv_rms_norm_f32((const float *) a->data, (float *) op->data, ggml_nrows(a), a->ne[0], eps, bc->stream);
        } break;
        case GGML_OP_ADD: rc = pkv_add_f32((const float *) a->data, (const float *) b->data, (float *) op->data, ggml_nrows(a), a->ne[0], ggml_nrows(b), bc->stream); break;
        case GGML_OP_MUL: rc = pkv_mul_f32((const float *) a->data, (const float *) b->data, (float *) op->data, ggml_nrows(a), a->ne[0], ggml_nrows(b), bc->stream); break;
        case GGML_OP_SOFT_MAX: {
            float scale, max_bias; memcpy(&scale, (const float *) op->op_params + 0, sizeof(float)); memcpy(&max_bias, (const float *) op->op_params + 1, sizeof(float));
            rc = pkv_soft_max_ext_f32((const float *) a->data, b ? b->data : nullptr, b && b->type == GGML_TYPE_F16, (float *) op->data, a->ne[0], a->ne[1], a->ne[2], a->ne[3],
                                      scale, max_bias, bc->stream);
        } break;
        case GGML_OP_UNARY: rc = pkv_silu_f32((const float *) a->data, (float *) op->data, ggml_nelements(a), bc->stream); break;
        case GGML_OP_MUL_MAT:
            if (mul_mat_2d(op)) rc = pkv_mul_mat(a->data, wtype(a), (const float *) b->data, (float *) op->data, a->ne[1], a->ne[0], b->ne[1], bc->stream);
            else {
                int64_t nba[4], nbb[4];
                for (int i = 0; i < 4; ++i) { nba[i] = (int64_t) a->nb[i]; nbb[i] = (int64_t) b->nb[i]; }
                if (in_place_of(a)) { MI355_LOG("mul_mat over a paged cache tensor is not supported\n"); return GGML_STATUS_FAILED; }
                rc = pkv_mul_mat_strided(a->data, a->type, a->ne, nba, (const float *) bc->src_data(b), b->ne, nbb, (float *) op->data, bc->stream);
            }
            break;
        default: break;
    }
    if (rc) { MI355_LOG("%s failed: %s\n", ggml_op_name(op->op), pkv_last_error()); return GGML_STATUS_FAILED; }
    return GGML_STATUS_SUCCESS;
}

// ---- backend -----------------------------------------------------------------------------------------------------------------
const char * be_name(ggml_backend_t) { return "MI355"; }
void be_free(ggml_backend_t b) {
    auto * c = (backend_ctx *) b->context;
    if (c->scratch) (void) hipFree(c->scratch);
    if (c->side) (void) hipFree(c->side);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
    delete b;
}
void be_sync(ggml_backend_t b) { (void) hipStreamSynchronize(((backend_ctx *) b->context)->stream); }
// every reader of t in the graph is (through RESHAPE / VIEW / PERMUTE / TRANSPOSE) the src[0] of an f32 ROPE or of a CPY: the ops that
// resolve their source through backend_ctx::src_data
// who reads a tensor in this graph, built once per graph_compute (r02: the per-tensor scan of every node cost the host ~0.3 ms per decode step)
struct graph_users {
    std::unordered_map<const ggml_tensor *, std::vector<int>> by_src, by_view;
    explicit graph_users(ggml_cgraph * g) {
        for (int j = 0; j < ggml_graph_n_nodes(g); ++j) {
            const ggml_tensor * nj = ggml_graph_node(g, j);
            for (int s_ = 0; s_ < GGML_MAX_SRC; ++s_) if (nj->src[s_]) by_src[nj->src[s_]].push_back(j);
            if (nj->view_src) by_view[nj->view_src].push_back(j);
        }
    }
};
bool consumers_take_moved(ggml_cgraph * g, const graph_users & u, const ggml_tensor * t, int depth = 0) {
    if (depth > 3 || (t->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    auto iv = u.by_view.find(t);
    if (iv != u.by_view.end()) for (int j : iv->second) if (ggml_graph_node(g, j)->src[0] != t) return false;
    auto is = u.by_src.find(t);
    if (is == u.by_src.end()) return false;
    for (int j : is->second) {
        const ggml_tensor * nj = ggml_graph_node(g, j);
        for (int s_ = 0; s_ < GGML_MAX_SRC; ++s_) {
            if (nj->src[s_] != t) continue;
            if (s_ != 0) return false;
            if ((nj->op == GGML_OP_ROPE && nj->type == GGML_TYPE_F32) || nj->op == GGML_OP_CPY) continue;
            if (nj->op == GGML_OP_RESHAPE || nj->op == GGML_OP_VIEW || nj->op == GGML_OP_PERMUTE || nj->op == GGML_OP_TRANSPOSE) {
                if (!consumers_take_moved(g, u, nj, depth + 1)) return false;
                continue;
            }
            return false;
        }
    }
    return true;
}
ggml_status be_graph_compute(ggml_backend_t b, ggml_cgraph * g) {
    auto * bc = (backend_ctx *) b->context;
    (void) hipDeviceSynchronize();  // set_tensor uses the null stream
    const int n_nodes = ggml_graph_n_nodes(g);
    std::vector<int> silu_of(n_nodes, -1);   // MUL node -> the SILU node whose launch was deferred into it
    auto overlaps = [](const ggml_tensor * a, const ggml_tensor * t) {
        const char * a0 = (const char *) a->data, * t0 = (const char *) t->data;
        return a0 && t0 && a0 < t0 + ggml_nbytes(t) && t0 < a0 + ggml_nbytes(a);
    };
    // a fused launch writes `out` rows while other workgroups may still be reading rows of `in`: fine when the two do not overlap, or
    // alias exactly (row r is written by the workgroup that has already read row r); a shifted overlap — ggml-alloc placing a result
    // over a freed, merged range (ggml-alloc.c:470-560) — is a race at more than one row
    auto alias_ok = [&](const ggml_tensor * out, const ggml_tensor * in) {
        return !overlaps(out, in) || (out->data == in->data && ggml_nbytes(out) == ggml_nbytes(in));
    };
    static const bool trace = getenv("GGML_MI355_TRACE") != nullptr;   // node list of every graph, for working on the fusion rules
    if (trace) for (int i = 0; i < n_nodes; ++i) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        MI355_LOG("node %3d %-14s %-24s data %p nbytes %zu src0 %s src1 %s\n", i, ggml_op_name(t->op), t->name, t->data, ggml_nbytes(t),
                  t->src[0] ? t->src[0]->name : "-", t->src[1] ? t->src[1]->name : "-");
    }
    const bool fuse = getenv("GGML_MI355_NO_FUSE") == nullptr;   // GGML_MI355_NO_FUSE=1: one launch per graph node, in graph order (read per graph: tests flip it)
    bc->moved.clear();
    // The side buffer of redirected K / V products (below) is sized ONCE per graph, for every product that could be redirected, and handed out by a
    // bump pointer: no reallocation while `moved` entries point into it, and no slot is reused before its ROPE / CPY consumers have run
    size_t side_need = 0, side_used = 0;
    const graph_users users(g);
    if (fuse) for (int i = 0; i < n_nodes; ++i) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        if (t->op == GGML_OP_MUL_MAT && t->src[0] && (t->src[0]->type == GGML_TYPE_Q4_K || t->src[0]->type == GGML_TYPE_Q8_0) && !(t->flags & GGML_TENSOR_FLAG_OUTPUT) &&
            consumers_take_moved(g, users, t)) side_need += (ggml_nbytes(t) + 255) / 256 * 256;
    }
    char * side_base = side_need && side_need <= ((size_t) 1 << 30) ? (char *) bc->side_buf(side_need) : nullptr;
    std::vector<char> done(n_nodes, 0);      // MUL_MAT nodes already run together with an earlier product on the same activations
    // RMS_NORM n1 -> MUL n2 of ONE row whose only readers are K-quant products on that single column (the decode-chain rule takes all of it)
    auto norm_feeds_decode_chain = [&](const ggml_tensor * n1, const ggml_tensor * n2) {
        if (getenv("GGML_MI355_CHAIN") && !(atoi(getenv("GGML_MI355_CHAIN")) & 1)) return false;
        if (n1->op != GGML_OP_RMS_NORM || n2->op != GGML_OP_MUL || n2->src[0] != n1 || ggml_nrows(n1) != 1 || n1->ne[0] > 8192 || n1->ne[0] % 256 != 0) return false;
        auto i1 = users.by_src.find(n1), i2 = users.by_src.find(n2);
        if (i1 == users.by_src.end() || i1->second.size() != 1 || i2 == users.by_src.end() || i2->second.empty() || i2->second.size() > 3) return false;
        for (int k : i2->second) {
            const ggml_tensor * t = ggml_graph_node(g, k);
            if (t->op != GGML_OP_MUL_MAT || t->src[1] != n2 || !t->src[0] || (t->src[0]->type != GGML_TYPE_Q4_K && t->src[0]->type != GGML_TYPE_Q5_K && t->src[0]->type != GGML_TYPE_Q6_K) || t->src[0]->ne[1] % 2 != 0) return false;
        }
        return true;
    };
    for (int i = 0; i < n_nodes; ++i) {
        ggml_tensor * n = ggml_graph_node(g, i);
        ggml_status st = GGML_STATUS_SUCCESS;
        if (done[i]) continue;
        // ---- decode-step chains (ONE activation column): [RMS_NORM -> MUL(norm weight) ->] {K-quant products on that column} [-> ADD(residual) |
        //      SILU(gate) * up]  =  ONE pkv_gemv_q launch (csrc/pkv_gemv.hip: the norm and the q8_K column in the product's prologue, the residual add or
        //      the SwiGLU in its epilogue) — a llama layer's decode step is then 7 launches (norm+qkv, rope/store, attention, merge, wo+add,
        //      norm+gate+up+swiglu, down+add) instead of r01's 10.  Same arithmetic as the per-node launches (GGML_MI355_NO_FUSE=1), bit for bit.
        static const int chain_mask = getenv("GGML_MI355_CHAIN") ? atoi(getenv("GGML_MI355_CHAIN")) : 7;   // debugging aid: 1 = norm chains, 2 = residual epilogue, 4 = SwiGLU epilogue
        if (fuse && (n->op == GGML_OP_RMS_NORM || n->op == GGML_OP_MUL_MAT)) {
            auto kq_col = [&](const ggml_tensor * t, const ggml_tensor * col) {   // t = MUL_MAT of K-quant rows with the single column `col`
                const ggml_tensor * w = t->src[0], * x = t->src[1];
                return t->op == GGML_OP_MUL_MAT && x == col && w && (w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K) && ggml_is_contiguous(w) && f32c(x) && f32c(t) &&
                       ggml_nrows(x) == 1 && w->ne[2] == 1 && w->ne[3] == 1 && w->ne[0] == x->ne[0] && w->ne[0] % 256 == 0 && w->ne[0] <= 49152 && w->ne[1] % 2 == 0 &&
                       !(t->flags & GGML_TENSOR_FLAG_OUTPUT);
            };
            auto only_user = [&](const ggml_tensor * t) -> int {                  // index of t's single consumer in this graph, or -1
                auto it = users.by_src.find(t);
                if (it == users.by_src.end() || it->second.size() != 1 || users.by_view.count(t) || (t->flags & GGML_TENSOR_FLAG_OUTPUT)) return -1;
                return it->second[0];
            };
            const ggml_tensor * col = nullptr, * xsrc = nullptr, * nw = nullptr;
            float eps = 0.0f;
            int i_mul = -1;
            if (n->op == GGML_OP_RMS_NORM && (chain_mask & 1) && supports_rms_norm(n) && ggml_nrows(n) == 1 && n->ne[0] <= 8192 && n->ne[0] % 256 == 0) {
                i_mul = only_user(n);
                ggml_tensor * n1 = i_mul >= 0 ? ggml_graph_node(g, i_mul) : nullptr;
                if (n1 && n1->op == GGML_OP_MUL && n1->src[0] == n && supports_binary(n1) && ggml_nrows(n1->src[1]) == 1 && !(n1->flags & GGML_TENSOR_FLAG_OUTPUT) &&
                    ((uintptr_t) n->src[0]->data % 16) == 0 && ((uintptr_t) n1->src[1]->data % 16) == 0 && ((uintptr_t) n->data % 16) == 0 && ((uintptr_t) n1->data % 16) == 0) {
                    col = n1; xsrc = n->src[0]; nw = n1->src[1];
                    memcpy(&eps, n->op_params, sizeof(float));
                }
            } else if (n->op == GGML_OP_MUL_MAT && ((uintptr_t) n->src[1]->data % 16) == 0) {
                col = n->src[1]; xsrc = col;
            }
            int prod[3], np = 0;
            bool ok = col != nullptr;
            if (ok && nw) {        // the products of the normalised column that this graph holds (the column itself is still written: readers elsewhere are fine)
                auto it = users.by_src.find(col);
                if (it != users.by_src.end()) for (int k : it->second) if (np < 3 && !done[k] && kq_col(ggml_graph_node(g, k), col)) prod[np++] = k;
                std::sort(prod, prod + np);
                ok = np > 0;
            } else if (ok) {       // a column that exists: this product and the ones that share it close by (as the grouping rule below finds them)
                for (int k = i; k < n_nodes && k <= i + 16 && np < 3; ++k) if (!done[k] && kq_col(ggml_graph_node(g, k), col)) prod[np++] = k;
                ok = np > 0 && prod[0] == i;
            }
            // epilogue of the chain.  Its output is written EARLY — in the products' launch — so it must not alias anything that launch still reads or also writes
            // (the column source, the norm weights, and norm_out / col_out, which workgroup 0 writes in the same launch), nor anything a node between the products
            // and the epilogue node reads or writes (those nodes run AFTER the fused launch): ggml-alloc's in-place placement of today's llama graphs is safe, a
            // graph cut by cb_eval / OUTPUT flags / adapters need not be
            bool wtype_uniform = true;                       // one launch = one row layout (plane-layout weights of a WEIGHTS buffer and GGUF-order ones do not mix)
            for (int q = 1; q < np; ++q) wtype_uniform = wtype_uniform && (wtype(ggml_graph_node(g, prod[q])->src[0]) & PKV_TYPE_PACKED) == (wtype(ggml_graph_node(g, prod[0])->src[0]) & PKV_TYPE_PACKED);
            if (!wtype_uniform) ok = false;
            auto early_ok = [&](const ggml_tensor * out, int last_idx, std::initializer_list<int> chain) {
                if (overlaps(out, xsrc) || (nw && (overlaps(out, nw) || overlaps(out, n) || overlaps(out, col)))) return false;
                for (int k = i; k <= last_idx && k < n_nodes; ++k) {
                    if (done[k] || std::find(chain.begin(), chain.end(), k) != chain.end() || std::find(prod, prod + np, k) != prod + np) continue;
                    const ggml_tensor * nk = ggml_graph_node(g, k);
                    if (overlaps(out, nk)) return false;
                    for (int s_ = 0; s_ < GGML_MAX_SRC; ++s_) if (nk->src[s_] && overlaps(out, nk->src[s_])) return false;
                }
                return true;
            };
            int i_silu = -1, i_mulgu = -1, i_add = -1;
            const ggml_tensor * resid = nullptr;
            float * yv[3] = {nullptr, nullptr, nullptr};
            if (ok && np == 2 && (chain_mask & 4)) {   // gate, SILU(gate), up, MUL(silu, up): build_ffn's LLM_FFN_SILU + LLM_FFN_PAR
                ggml_tensor * p0 = ggml_graph_node(g, prod[0]), * p1 = ggml_graph_node(g, prod[1]);
                const int us0 = only_user(p0), us1 = only_user(p1);
                ggml_tensor * sn = us0 >= 0 ? ggml_graph_node(g, us0) : nullptr;
                const int um = sn ? only_user(sn) : -1;
                ggml_tensor * mn = um >= 0 ? ggml_graph_node(g, um) : nullptr;
                if (sn && mn && sn->op == GGML_OP_UNARY && supports_silu(sn) && sn->src[0] == p0 && mn->op == GGML_OP_MUL && us1 == um && !done[us0] && !done[um] &&
                    ((mn->src[0] == sn && mn->src[1] == p1) || (mn->src[1] == sn && mn->src[0] == p1)) && f32c(mn) && ggml_nelements(mn) == p0->ne[0] && p0->ne[0] == p1->ne[0] &&
                    early_ok(mn, um, {i_mul, us0, um})) { i_silu = us0; i_mulgu = um; yv[0] = (float *) mn->data; }
            }
            if (ok && np == 1 && (chain_mask & 2)) {   // attn_out / ffn_down + the residual stream
                ggml_tensor * p0 = ggml_graph_node(g, prod[0]);
                const int ua = only_user(p0);
                ggml_tensor * an = ua >= 0 ? ggml_graph_node(g, ua) : nullptr;
                if (an && an->op == GGML_OP_ADD && !done[ua] && supports_binary(an) && ggml_are_same_shape(an->src[0], an->src[1]) && (an->src[0] == p0) != (an->src[1] == p0)) {
                    const ggml_tensor * r = an->src[0] == p0 ? an->src[1] : an->src[0];
                    bool r_ready = true;                     // the other operand exists already: a leaf, or a node that has run
                    for (int q = i; q < n_nodes && r_ready; ++q) if (ggml_graph_node(g, q) == r || (r->view_src && ggml_graph_node(g, q) == r->view_src)) r_ready = done[q] != 0;
                    if (r_ready && early_ok(an, ua, {i_mul, ua}) && alias_ok(an, r)) { i_add = ua; resid = r; yv[0] = (float *) an->data; }
                }
            }
            size_t side_bytes = 0;
            bool redirect[3] = {false, false, false};
            if (ok && i_silu < 0 && i_add < 0) {
                if (!nw && np == 1) ok = false;              // a lone product without norm or epilogue: nothing to merge, the rules below handle it
                for (int q = 0; q < np && ok; ++q) {         // plain outputs: must not be written over the column other workgroups still read, nor over each other
                    ggml_tensor * pq = ggml_graph_node(g, prod[q]);
                    bool safe = !overlaps(pq, xsrc) && !(nw && overlaps(pq, nw));
                    for (int r_ = 0; r_ < q && safe; ++r_) if (overlaps(pq, ggml_graph_node(g, prod[r_]))) safe = false;
                    for (int k = i; k < prod[q] && safe; ++k) {          // pulled forward over the nodes in between (none of them may touch its output range)
                        const ggml_tensor * nk = ggml_graph_node(g, k);
                        if (k == i_mul || std::find(prod, prod + np, k) != prod + np) continue;
                        if (overlaps(pq, nk)) safe = false;
                        for (int s_ = 0; s_ < GGML_MAX_SRC && safe; ++s_) if (nk->src[s_] && overlaps(pq, nk->src[s_])) safe = false;
                    }
                    if (!safe && !overlaps(pq, xsrc) && side_base && consumers_take_moved(g, users, pq)) { redirect[q] = true; safe = true; side_bytes += (ggml_nbytes(pq) + 255) / 256 * 256; }
                    ok = safe;
                    yv[q] = (float *) pq->data;
                }
                if (ok && side_used + side_bytes > side_need) ok = false;
            }
            if (ok) {
                pkv_gemv gv = {};
                gv.n = np; gv.K = col->ne[0]; gv.x = (const float *) bc->src_data(xsrc); gv.norm_w = nw ? (const float *) nw->data : nullptr; gv.eps = eps;
                gv.silu_mul = i_silu >= 0; gv.residual = resid ? (const float *) bc->src_data(resid) : nullptr;
                // the normalised column and the norm are still written (one workgroup): products of the same column that sit in ANOTHER graph of the
                // scheduler (a split, or a cb_eval observer cutting the graph) read them
                if (nw) { gv.norm_out = (float *) n->data; gv.col_out = (float *) col->data; }
                char * side = side_base + side_used;
                for (int q = 0; q < np; ++q) {
                    ggml_tensor * pq = ggml_graph_node(g, prod[q]);
                    gv.type_w[q] = (int32_t) wtype(pq->src[0]); gv.w[q] = pq->src[0]->data; gv.M[q] = pq->src[0]->ne[1];
                    gv.y[q] = yv[q] ? yv[q] : yv[0];
                    if (redirect[q]) { gv.y[q] = (float *) side; bc->moved[pq] = side; side += (ggml_nbytes(pq) + 255) / 256 * 256; }
                }
                if (trace) MI355_LOG("chain at node %d (%s): np %d types %d %d %d norm %d silu %d add %d redirect %d %d %d side %zu/%zu x %p\n", i, n->name, np, gv.type_w[0], gv.type_w[1], gv.type_w[2],
                                     nw != nullptr, i_silu, i_add, redirect[0], redirect[1], redirect[2], side_used, side_need, (const void *) gv.x);
                if (pkv_gemv_q(&gv, bc->stream) == PKV_OK) {
                    side_used += side_bytes;
                    if (i_mul >= 0) done[i_mul] = 1;
                    for (int q = 0; q < np; ++q) done[prod[q]] = 1;
                    if (i_silu >= 0) { done[i_silu] = 1; done[i_mulgu] = 1; }
                    if (i_add >= 0) done[i_add] = 1;
                    continue;
                }
                for (int q = 0; q < np; ++q) if (redirect[q]) bc->moved.erase(ggml_graph_node(g, prod[q]));
            }
        }
        // q4_K products that read the SAME activation tensor (wq / wk / wv of a layer, gate / up of its FFN) run as one launch: the later
        // ones are pulled forward to the first, when their output does not alias anything the nodes in between read or write
        // (ggml-alloc hands freed ranges to later nodes, ggml-alloc.c:470-560)
        if (fuse && n->op == GGML_OP_MUL_MAT && (n->src[0]->type == GGML_TYPE_Q4_K || n->src[0]->type == GGML_TYPE_Q8_0) && supports_mul_mat(n)) {
            int grp[PKV_MUL_MAT_MAX], ngp = 1;
            bool redirect[PKV_MUL_MAT_MAX] = {false};
            grp[0] = i;
            for (int k = i + 1; k < n_nodes && k <= i + 16 && ngp < PKV_MUL_MAT_MAX; ++k) {
                ggml_tensor * nk = ggml_graph_node(g, k);
                if (nk->op != GGML_OP_MUL_MAT || nk->src[1] != n->src[1] || !nk->src[0] || nk->src[0]->type != n->src[0]->type || wtype(nk->src[0]) != wtype(n->src[0]) || !supports_mul_mat(nk) || done[k]) continue;
                bool safe = true;
                for (int q = i; q < k && safe; ++q) {
                    const ggml_tensor * nq = ggml_graph_node(g, q);
                    if (overlaps(nk, nq)) safe = false;
                    for (int s_ = 0; s_ < GGML_MAX_SRC && safe; ++s_) if (nq->src[s_] && overlaps(nk, nq->src[s_])) safe = false;
                }
                // Kcur / Vcur: ggml-alloc gives them Qcur's address (each is dead after its out-of-place ROPE / CPY).  They can still be
                // computed now, into a side buffer, when everything that reads them is a ROPE or CPY of this backend (through view
                // ops): those take their source through backend_ctx::src_data
                // (a product that is a graph output, or whose consumers are not all in THIS scheduler split, keeps its own address: another
                // split copies from nk->data)
                if (!safe && side_base && !(nk->flags & GGML_TENSOR_FLAG_OUTPUT) && consumers_take_moved(g, users, nk)) { redirect[ngp] = true; safe = true; }
                if (safe) grp[ngp++] = k;
            }
            if (ngp > 1) {
                const void * wv[PKV_MUL_MAT_MAX]; float * yv[PKV_MUL_MAT_MAX]; int64_t mv[PKV_MUL_MAT_MAX];
                size_t side_bytes = 0;
                for (int q = 0; q < ngp; ++q) if (redirect[q]) side_bytes += (ggml_nbytes(ggml_graph_node(g, grp[q])) + 255) / 256 * 256;
                char * side = side_bytes ? side_base + side_used : nullptr;
                if (side_bytes && side_used + side_bytes > side_need) { MI355_LOG("side buffer accounting broke (%zu + %zu > %zu)\n", side_used, side_bytes, side_need); return GGML_STATUS_FAILED; }
                side_used += side_bytes;
                for (int q = 0; q < ngp; ++q) {
                    const ggml_tensor * nq = ggml_graph_node(g, grp[q]);
                    wv[q] = nq->src[0]->data; yv[q] = (float *) nq->data; mv[q] = nq->src[0]->ne[1];
                    if (redirect[q]) { yv[q] = (float *) side; bc->moved[nq] = side; side += (ggml_nbytes(nq) + 255) / 256 * 256; }
                    done[grp[q]] = 1;
                }
                if (pkv_mul_mat_multi(ngp, wv, wtype(n->src[0]), (const float *) n->src[1]->data, yv, mv, n->src[0]->ne[0], n->src[1]->ne[1], bc->stream) != PKV_OK) {
                    MI355_LOG("mul_mat_multi failed: %s\n", pkv_last_error());
                    return GGML_STATUS_FAILED;
                }
                continue;
            }
        }
        // two-node patterns of a llama layer run as one launch (both outputs are still written: the graph names them):
        //   RMS_NORM -> MUL(norm, weight row)      (/ref/src/llama-model.cpp: build_norm)
        //   UNARY(SILU)(gate) ... MUL(silu, up)    (build_ffn, LLM_FFN_SILU + LLM_FFN_PAR).  The graph's topological order puts the up
        //   product BETWEEN the two (gate product, silu, up product, mul): the SILU launch is deferred into the MUL when no node in
        //   between reads the silu result or writes over it or over the gate.
        if (silu_of[i] >= 0) {
            ggml_tensor * sn = ggml_graph_node(g, silu_of[i]);
            const ggml_tensor * up = n->src[0] == sn ? n->src[1] : n->src[0];
            if (pkv_silu_mul_f32((const float *) sn->src[0]->data, (const float *) up->data, (float *) sn->data, (float *) n->data,
                                 ggml_nelements(sn), bc->stream) != PKV_OK) return GGML_STATUS_FAILED;
            continue;
        }
        if (fuse && n->op == GGML_OP_UNARY && supports_silu(n)) {
            int j = -1;
            for (int k = i + 1; k < n_nodes && k <= i + 3; ++k) {
                ggml_tensor * nk = ggml_graph_node(g, k);
                bool reads = false;
                for (int s_ = 0; s_ < GGML_MAX_SRC; ++s_) reads = reads || (nk->src[s_] && (nk->src[s_] == n || nk->src[s_]->view_src == n));
                if (nk->op == GGML_OP_MUL && reads && supports_binary(nk) && ggml_are_same_shape(nk->src[0], nk->src[1]) &&
                    (nk->src[0] == n) != (nk->src[1] == n) &&
                    (!overlaps(nk, n->src[0]) || nk->data == n->src[0]->data) && (!overlaps(nk, n) || nk->data == n->data)) { j = k; break; }   // ggml-alloc runs SILU and MUL in place (ggml-alloc.c:41-67): exact aliases are element-wise safe
                if (reads || nk->view_src == n || overlaps(nk, n) || overlaps(nk, n->src[0])) break;
            }
            if (j >= 0) { silu_of[j] = i; continue; }
        }
        //   ADD(a, b) -> RMS_NORM -> MUL            (the residual add in front of a norm: ffn_inp / l_out; its sum is still written)
        if (fuse && i + 2 < n_nodes && n->op == GGML_OP_ADD && supports_binary(n) && ggml_are_same_shape(n->src[0], n->src[1])) {
            ggml_tensor * n1 = ggml_graph_node(g, i + 1), * n2 = ggml_graph_node(g, i + 2);
            if (norm_feeds_decode_chain(n1, n2)) {}   // one row whose norm goes into the products' own launch (above): the ADD runs by itself
            else if (n1->op == GGML_OP_RMS_NORM && n1->src[0] == n && supports_rms_norm(n1) && n2->op == GGML_OP_MUL && n2->src[0] == n1 && supports_binary(n2) &&
                ggml_nrows(n2->src[1]) == 1 && n1->data != n->data && n2->data != n->data &&   // the sum must survive: it is the next residual
                !overlaps(n1, n) && !overlaps(n2, n) &&
                alias_ok(n, n->src[0]) && alias_ok(n, n->src[1]) && alias_ok(n1, n->src[0]) && alias_ok(n1, n->src[1]) &&
                alias_ok(n2, n->src[0]) && alias_ok(n2, n->src[1])) {
                float eps; memcpy(&eps, n1->op_params, sizeof(float));
                if (pkv_add_rms_norm_mul_f32((const float *) n->src[0]->data, (const float *) n->src[1]->data, (float *) n->data, (const float *) n2->src[1]->data,
                                             (float *) n1->data, (float *) n2->data, ggml_nrows(n), n->ne[0], eps, bc->stream) != PKV_OK) return GGML_STATUS_FAILED;
                i += 2;
                continue;
            }
        }
        if (fuse && i + 1 < n_nodes) {
            ggml_tensor * nx = ggml_graph_node(g, i + 1);
            if (n->op == GGML_OP_RMS_NORM && nx->op == GGML_OP_MUL && nx->src[0] == n && supports_rms_norm(n) && supports_binary(nx) &&
                ggml_nrows(nx->src[1]) == 1 && alias_ok(n, n->src[0]) && alias_ok(nx, n->src[0])) {
                float eps; memcpy(&eps, n->op_params, sizeof(float));
                if (pkv_rms_norm_mul_f32((const float *) n->src[0]->data, (const float *) nx->src[1]->data, (float *) n->data, (float *) nx->data,
                                         ggml_nrows(n), n->ne[0], eps, bc->stream) != PKV_OK) return GGML_STATUS_FAILED;
                ++i;
                continue;
            }
        }
        //   ROPE(q) + [ROPE(k) -> CPY(k -> cache)] + CPY(v -> cache): everything between the q / k / v products and the attention of
        //   a layer in ONE launch, when k and v are already computed (the products were pulled together above), ROPE(k) feeds
        //   nothing but its CPY, and both cache tensors have their paged mirrors
        if (fuse && n->op == GGML_OP_ROPE && n->type == GGML_TYPE_F32 && n->src[0]->type == GGML_TYPE_F32 && n->src[0]->ne[3] == 1 && ggml_is_contiguous(n)) {
            auto ready = [&](const ggml_tensor * t) {      // t's producer has run: an earlier node, or one already pulled forward
                while (t->view_src) t = t->view_src;
                for (int q = std::max(0, i - 12); q < n_nodes && q <= i + 12; ++q) if (ggml_graph_node(g, q) == t) return q < i || done[q] != 0;
                return false;
            };
            int ir = -1, ick = -1, icv = -1;
            for (int k = i + 1; k < n_nodes && k <= i + 12; ++k) {
                ggml_tensor * nk = ggml_graph_node(g, k);
                if (done[k]) continue;
                if (ir < 0 && nk->op == GGML_OP_ROPE && nk->type == GGML_TYPE_F32 && nk->src[0]->type == GGML_TYPE_F32 && nk->src[1] == n->src[1] &&
                    nk->src[2] == n->src[2] && memcmp(nk->op_params, n->op_params, sizeof(n->op_params)) == 0) ir = k;
                else if (ir >= 0 && ick < 0 && nk->op == GGML_OP_CPY && nk->src[0] == ggml_graph_node(g, ir) && supports_cpy(nk)) ick = k;
                else if (icv < 0 && nk->op == GGML_OP_CPY && supports_cpy(nk) && (ir < 0 || nk->src[0] != ggml_graph_node(g, ir))) icv = k;
            }
            if (ir >= 0 && ick >= 0 && icv >= 0) {
                ggml_tensor * r2 = ggml_graph_node(g, ir), * ck = ggml_graph_node(g, ick), * cv = ggml_graph_node(g, icv);
                int users = 0;
                for (int q = 0; q < n_nodes; ++q) for (int s_ = 0; s_ < GGML_MAX_SRC; ++s_) users += ggml_graph_node(g, q)->src[s_] == r2;
                int64_t ck0, ckn, cv0, cvn;
                const kv_mirror * mk = store_mirror(ck, &ck0, &ckn), * mv = store_mirror(cv, &cv0, &cvn);
                const ggml_tensor * xq = n->src[0], * xk = r2->src[0], * xv = cv->src[0];
                pkv_rope_params rp;
                rope_params(n, &rp);
                if (users == 1 && !(r2->flags & GGML_TENSOR_FLAG_OUTPUT) && mk && mv && !mk->is_v && mv->is_v && mk->D == mv->D && mk->n_head_kv == mv->n_head_kv &&
                    xq->ne[0] == mk->D && xk->ne[0] == mk->D && xk->ne[1] == mk->n_head_kv && xk->ne[3] == 1 && xk->ne[2] == xq->ne[2] && ckn == xq->ne[2] &&
                    cvn == xq->ne[2] && ((rp.mode & 2) == 0 || rp.n_dims == mk->D) && ready(xk) && ready(xv)) {
                    pkv_qkv_store_args a = {};
                    a.q = (const float *) bc->src_data(xq); a.q_stride_tok = (int64_t) (xq->nb[2] / sizeof(float)); a.q_stride_head = (int64_t) (xq->nb[1] / sizeof(float));
                    a.q_out = (float *) n->data;
                    a.k = (const float *) bc->src_data(xk); a.k_stride_tok = (int64_t) (xk->nb[2] / sizeof(float)); a.k_stride_head = (int64_t) (xk->nb[1] / sizeof(float));
                    a.k_dst = mk->in_place ? nullptr : ck->src[1]->data; a.k_pool = mk->pool; a.k_cell0 = ck0;
                    a.v = (const float *) bc->src_data(xv); a.v_dst = mv->in_place ? nullptr : cv->src[1]->data; a.v_pool = mv->pool; a.v_cell0 = cv0;
                    a.pos = (const int32_t *) n->src[1]->data; a.rope = &rp;
                    a.type_k = ck->src[1]->type; a.type_v = cv->src[1]->type; a.n_tok = (int32_t) xq->ne[2]; a.n_head = (int32_t) xq->ne[1];
                    a.n_head_kv = mk->n_head_kv; a.head_dim = mk->D; a.block_size = BLOCK;
                    if (pkv_qkv_store_f32_paged(&a, bc->stream) == PKV_OK) {
                        done[ir] = done[ick] = done[icv] = 1;
                        goto next_node;
                    }
                }
            }
        }
        //   ROPE(k_cur) ... CPY(roped -> cache view)   (build_attn's KV store): the CPY is pulled forward into the ROPE launch — its
        //   destination, cells of the cache nobody reads before the attention of this layer, cannot alias anything in between
        if (fuse && n->op == GGML_OP_ROPE && n->type == GGML_TYPE_F32 && n->src[0]->type == GGML_TYPE_F32) {
            for (int k = i + 1; k < n_nodes && k <= i + 8; ++k) {
                ggml_tensor * nk = ggml_graph_node(g, k);
                if (nk->op != GGML_OP_CPY || nk->src[0] != n || done[k] || !supports_cpy(nk)) continue;
                int64_t cell0, n_cells;
                const kv_mirror * m = store_mirror(nk, &cell0, &n_cells);
                pkv_rope_params rp;
                rope_params(n, &rp);
                const ggml_tensor * x = n->src[0];
                if (m && !m->is_v && x->ne[0] == m->D && x->ne[1] == m->n_head_kv && x->ne[2] == n_cells && x->ne[3] == 1 && ggml_is_contiguous(n) &&
                    ((rp.mode & 2) == 0 || rp.n_dims == m->D) &&
                    pkv_rope_convert_f32_paged((const float *) bc->src_data(x), (int64_t) (x->nb[2] / sizeof(float)), (int64_t) (x->nb[1] / sizeof(float)),
                                               (const int32_t *) n->src[1]->data, &rp, (float *) n->data, m->in_place ? nullptr : nk->src[1]->data, nk->src[1]->type, m->pool, cell0, n_cells,
                                               m->n_head_kv, m->D, BLOCK, bc->stream) == PKV_OK) {
                    done[k] = 1;
                    st = GGML_STATUS_SUCCESS;
                    goto next_node;
                }
                break;
            }
        }
        switch (n->op) {
            case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: break;
            case GGML_OP_FLASH_ATTN_EXT: st = run_flash_attn(bc, n); break;
            case GGML_OP_ROPE:           st = run_rope(bc, n, false); break;
            case GGML_OP_ROPE_BACK:      st = run_rope(bc, n, true); break;
            case GGML_OP_CPY:            st = run_cpy(bc, n); break;
            case GGML_OP_CONT:           st = run_cpy_strided(bc, n->src[0], n); break;
            case GGML_OP_RMS_NORM: case GGML_OP_ADD: case GGML_OP_MUL: case GGML_OP_UNARY: case GGML_OP_MUL_MAT: case GGML_OP_SOFT_MAX: st = run_layer_op(bc, n); break;
            default: MI355_LOG("op %s reached graph_compute but is not supported\n", ggml_op_name(n->op)); return GGML_STATUS_FAILED;
        }
        if (st != GGML_STATUS_SUCCESS) return st;
    next_node:;
    }
    (void) hipStreamSynchronize(bc->stream);
    return GGML_STATUS_SUCCESS;
}
const ggml_backend_i BACKEND_IFACE = {
    /* get_name           */ be_name,
    /* free               */ be_free,
    /* set_tensor_async   */ nullptr,
    /* get_tensor_async   */ nullptr,
    /* cpy_tensor_async   */ nullptr,
    /* synchronize        */ be_sync,
    /* graph_plan_create  */ nullptr,
    /* graph_plan_free    */ nullptr,
    /* graph_plan_update  */ nullptr,
    /* graph_plan_compute */ nullptr,
    /* graph_compute      */ be_graph_compute,
    /* event_record       */ nullptr,
    /* event_wait         */ nullptr,
};
ggml_guid_t be_guid() {
    static ggml_guid guid = {0x6d, 0x69, 0x33, 0x35, 0x35, 0x78, 0x2d, 0x70, 0x61, 0x67, 0x65, 0x64, 0x2d, 0x6b, 0x76, 0x01};
    return &guid;
}

// ---- device / registry -----------------------------------------------------------------------------------------------------
ggml_backend_buffer_type BUFT;   // filled in by reg()
ggml_backend_device      DEVICE;
ggml_backend_reg         REG;

const char * dev_name(ggml_backend_dev_t) { return "MI355"; }
const char * dev_desc(ggml_backend_dev_t) { return "AMD Instinct MI355X (gfx950) paged-KV attention path"; }
void dev_memory(ggml_backend_dev_t, size_t * free, size_t * total) {
    if (hipMemGetInfo(free, total) != hipSuccess) { *free = 0; *total = 0; }
}
enum ggml_backend_dev_type dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
void dev_props(ggml_backend_dev_t dev, ggml_backend_dev_props * p) {
    p->name = dev_name(dev); p->description = dev_desc(dev); p->type = dev_type(dev);
    dev_memory(dev, &p->memory_free, &p->memory_total);
    p->caps = {/* async */ false, /* host_buffer */ false, /* buffer_from_host_ptr */ false, /* events */ false};
}
ggml_backend_t dev_init(ggml_backend_dev_t dev, const char *) {
    auto * bc = new backend_ctx;
    if (hipStreamCreate(&bc->stream) != hipSuccess) { delete bc; return nullptr; }
    return new ggml_backend{be_guid(), BACKEND_IFACE, dev, bc};
}
ggml_backend_buffer_type_t dev_buft(ggml_backend_dev_t) { return &BUFT; }
bool dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op) {
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return true;
        case GGML_OP_FLASH_ATTN_EXT: return supports_flash_attn(op);
        case GGML_OP_ROPE: case GGML_OP_ROPE_BACK: return supports_rope(op);
        case GGML_OP_CPY: return supports_cpy(op);
        case GGML_OP_CONT: return supports_cont(op);
        case GGML_OP_RMS_NORM: return supports_rms_norm(op);
        case GGML_OP_ADD: case GGML_OP_MUL: return supports_binary(op);
        case GGML_OP_UNARY: return supports_silu(op);
        case GGML_OP_SOFT_MAX: return supports_soft_max(op);
        case GGML_OP_MUL_MAT: return supports_mul_mat(op);
        default: return false;
    }
}
bool dev_supports_buft(ggml_backend_dev_t, ggml_backend_buffer_type_t buft) { return buft == &BUFT; }

const char * reg_name(ggml_backend_reg_t) { return "MI355"; }
size_t reg_count(ggml_backend_reg_t) { return pkv_device_count() > 0 ? 1 : 0; }
ggml_backend_dev_t reg_dev(ggml_backend_reg_t, size_t i) { return i == 0 ? &DEVICE : nullptr; }
void * reg_proc(ggml_backend_reg_t, const char *) { return nullptr; }

ggml_backend_reg_t reg() {
    static bool once = false;
    if (!once) {
        once = true;
        BUFT   = {{buft_name, buft_alloc, buft_align, /* get_max_size */ nullptr, /* get_alloc_size */ nullptr, /* is_host */ nullptr}, &DEVICE, nullptr};
        DEVICE = {{dev_name, dev_desc, dev_memory, dev_type, dev_props, dev_init, dev_buft, /* host buft */ nullptr,
                   /* buffer_from_host_ptr */ nullptr, dev_supports_op, dev_supports_buft, /* offload_op */ nullptr,
                   /* event_new */ nullptr, /* event_free */ nullptr, /* event_synchronize */ nullptr},
                  &REG, nullptr};
        REG    = {GGML_BACKEND_API_VERSION, {reg_name, reg_count, reg_dev, reg_proc}, nullptr};
    }
    return &REG;
}

}  // namespace

extern "C" {
__attribute__((visibility("default"))) ggml_backend_reg_t ggml_backend_init(void) { return reg(); }
__attribute__((visibility("default"))) int ggml_backend_score(void) { return pkv_device_count() > 0 ? 100 : 0; }
}
