// pkv_capi.cpp — the extern "C" boundary declared in include/mi355_paged_kv.h.
// Argument validation happens here, on the host, BEFORE any launch (a faulting kernel can reset the whole GPU node).
#include "pkv_cache.h"
#include "pkv_common.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

namespace pkv {

static thread_local char g_err[512] = "";

void set_error(const char * fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char * what) {
    set_error("HIP error %d (%s) in %s", (int) e, hipGetErrorString(e), what);
    return PKV_ERR_HIP;
}
int ensure_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void) hipGetLastError();
        set_error("no HIP device visible: this library has no CPU path (kernels are gfx950 code objects)");
        return PKV_ERR_NO_DEVICE;
    }
    return PKV_OK;
}

// device half of a pkv_cache
struct CacheDev {
    std::vector<void *> k, v;
    int32_t * scratch = nullptr;  // device i32 staging for copy / shift lists
    size_t    scratch_n = 0;
};

}  // namespace pkv

using namespace pkv;

struct pkv_cache {
    KvCachePaged host;
    CacheDev     dev;
    explicit pkv_cache(const pkv_cache_params & p) : host(p) {}
};
struct pkv_toy_block_manager { ToyBlockManager bm; explicit pkv_toy_block_manager(int n) : bm(n) {} };
struct pkv_toy_kv_manager    { ToyKvManager m;     explicit pkv_toy_kv_manager(int c) : m(c) {} };
struct pkv_toy_block_kv_cache { ToyBlockKVCache c; pkv_toy_block_kv_cache(int nb, int bs, int h, int d) : c(nb, bs, h, d) {} };

namespace {
struct StateLayout { size_t n, meta, k_row, v_row, total; };
StateLayout state_layout(const pkv_cache_params & p, const std::vector<KvCachePaged::StateCell> & cells) {
    StateLayout L{};
    L.n = cells.size();
    L.meta = 4;
    for (const auto & c : cells) L.meta += 8 + 4 * c.seq_ids.size();
    L.k_row = p.n_layer > 0 ? row_size(p.type_k, (int64_t) p.n_head_kv * p.head_dim) : 0;
    L.v_row = p.n_layer > 0 ? row_size(p.type_v, (int64_t) p.n_head_kv * p.head_dim) : 0;
    L.total = L.meta + 8 + (size_t) p.n_layer * (12 + L.n * L.k_row) + (size_t) p.n_layer * (12 + L.n * L.v_row);
    return L;
}
template <typename T> void put(char *& w, T v) { memcpy(w, &v, sizeof(T)); w += sizeof(T); }
template <typename T> bool get(const char *& r, const char * end, T * v) {
    if ((size_t) (end - r) < sizeof(T)) return false;
    memcpy(v, r, sizeof(T)); r += sizeof(T);
    return true;
}
}  // namespace

extern "C" {

int          pkv_abi_version(void) { return PKV_ABI_VERSION; }
const char * pkv_last_error(void) { return g_err; }
int          pkv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return n;
}
const char * pkv_device_arch(int device) {
    static thread_local char arch[256];
    hipDeviceProp_t prop;
    if (device < 0 || device >= pkv_device_count() || hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
    snprintf(arch, sizeof(arch), "%s", prop.gcnArchName);
    if (char * c = strchr(arch, ':')) *c = 0;
    return arch;
}

size_t pkv_row_size(int type, int64_t ne) { return row_size(type, ne); }
size_t pkv_tile_bytes(int type, int block_size, int head_dim) { return tile_bytes(type, block_size, head_dim); }
size_t pkv_pool_bytes(int type, int n_blocks, int n_head_kv, int block_size, int head_dim) {
    return (size_t) n_blocks * n_head_kv * tile_bytes(type, block_size, head_dim);
}

// ---- kernels -----------------------------------------------------------------------------------------------
int pkv_rope_f32(const float * x, const int32_t * pos, float * out, int n_tok, int n_head, int head_dim,
                 const pkv_rope_params * rope, int inverse, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && pos && out && n_tok >= 0 && n_head > 0 && head_dim > 0 && head_dim % 2 == 0, "pkv_rope_f32: bad arguments");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_rope_f32(x, pos, out, n_tok, n_head, head_dim, r, inverse, (hipStream_t) stream);
}

int pkv_rope_f32_strided(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, float * out, int n_tok,
                         int n_head, int head_dim, const pkv_rope_params * rope, int inverse, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && pos && out && n_tok >= 0 && n_head > 0 && head_dim > 0 && head_dim % 2 == 0 && xs_tok > 0 && xs_head > 0,
                "pkv_rope_f32_strided: bad arguments");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_rope_f32(x, pos, out, n_tok, n_head, head_dim, r, inverse, (hipStream_t) stream, xs_tok, xs_head);
}

int pkv_rope_f16_strided(const void * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, void * out, int n_tok,
                         int n_head, int head_dim, const pkv_rope_params * rope, int inverse, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && pos && out && n_tok >= 0 && n_head > 0 && head_dim > 0 && head_dim % 2 == 0 && xs_tok > 0 && xs_head > 0,
                "pkv_rope_f16_strided: bad arguments");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_rope_f16(x, pos, out, n_tok, n_head, head_dim, r, inverse, (hipStream_t) stream, xs_tok, xs_head);
}

int pkv_convert_f32(const float * x, void * dst, int type, int64_t n_rows, int64_t ne0, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && dst && n_rows >= 0 && ne0 > 0 && (kv_type_ok(type) || flat_only_type(type)), "pkv_convert_f32: bad arguments");
    if (flat_only_type(type)) return launch_quantize_blocks(x, dst, type, n_rows, ne0, (hipStream_t) stream);
    return launch_convert_rows(x, dst, type, n_rows, ne0, (hipStream_t) stream);
}
int pkv_convert_f32_paged(const float * x, void * dst, int type, void * pool, int is_v, int64_t cell0, int64_t n_cells, int n_head_kv, int head_dim,
                          int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && pool && kv_type_ok(type) && cell0 >= 0 && n_cells >= 0 && n_head_kv > 0 && block_size > 0 && block_size % 32 == 0,
                "pkv_convert_f32_paged: bad arguments");
    return launch_convert_rows_mirror(x, dst, type, pool, is_v, cell0, n_cells * n_head_kv, n_head_kv, head_dim, block_size, (hipStream_t) stream);
}

int pkv_rope_convert_f32_paged(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, const pkv_rope_params * rope, float * roped, void * dst,
                               int type, void * pool, int64_t cell0, int64_t n_tok, int n_head_kv, int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && pos && rope && pool && kv_type_ok(type) && cell0 >= 0 && n_tok >= 0 && n_head_kv > 0 && block_size > 0 && block_size % 32 == 0 &&
                xs_tok >= 0 && xs_head >= 0, "pkv_rope_convert_f32_paged: bad arguments");
    PKV_REQUIRE((rope->mode & 2) == 0 || rope->n_dims == head_dim, "pkv_rope_convert_f32_paged: NEOX pairs need n_dims == head_dim");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_rope_convert_mirror(x, xs_tok, xs_head, pos, roped, dst, type, pool, cell0, n_tok, n_head_kv, head_dim, block_size, r, (hipStream_t) stream);
}

int pkv_qkv_store_f32_paged(const pkv_qkv_store_args * a, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(a && a->q && a->q_out && a->k && a->k_pool && a->v && a->v_pool && a->pos && a->rope, "pkv_qkv_store_f32_paged: null pointer");
    PKV_REQUIRE(kv_type_ok(a->type_k) && kv_type_ok(a->type_v) && a->n_tok >= 0 && a->n_head > 0 && a->n_head_kv > 0 && a->k_cell0 >= 0 && a->v_cell0 >= 0 &&
                a->block_size > 0 && a->block_size % 32 == 0, "pkv_qkv_store_f32_paged: bad arguments");
    PKV_REQUIRE((a->rope->mode & 2) == 0 || a->rope->n_dims == a->head_dim, "pkv_qkv_store_f32_paged: NEOX pairs need n_dims == head_dim");
    RopeDev r;
    if (int rc = make_rope(a->rope, a->head_dim, &r)) return rc;
    return launch_qkv_store_mirror(a, r, (hipStream_t) stream);
}

int pkv_kv_store(const float * k_cur, const float * v_cur, void * k_pool, void * v_pool, int type_k, int type_v,
                 const int32_t * slot_block, const int32_t * slot_off, const int32_t * pos,
                 const pkv_rope_params * rope, int n_tok, int n_head_kv, int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE((k_cur || v_cur) && slot_block && slot_off, "pkv_kv_store: null pointer");
    PKV_REQUIRE(!k_cur || k_pool, "pkv_kv_store: k_cur without k_pool");
    PKV_REQUIRE(!v_cur || v_pool, "pkv_kv_store: v_cur without v_pool");
    PKV_REQUIRE(kv_type_ok(type_k) && kv_type_ok(type_v), "pkv_kv_store: KV types must be f16/q8_0/q4_0");
    PKV_REQUIRE(n_tok >= 0 && n_head_kv > 0 && block_size > 0 && block_size % 8 == 0, "pkv_kv_store: bad sizes");
    PKV_REQUIRE(!rope || pos, "pkv_kv_store: fused rope needs pos");
    RopeDev r;
    if (rope) { if (int rc = make_rope(rope, head_dim, &r)) return rc; }
    return launch_kv_store(nullptr, 0, k_cur, v_cur, k_pool, v_pool, type_k, type_v, slot_block, slot_off, pos,
                           rope ? &r : nullptr, n_tok, n_head_kv, head_dim, block_size, (hipStream_t) stream);
}

int pkv_qkv_prepare(float * q, int n_head, const float * k_cur, const float * v_cur, void * k_pool, void * v_pool,
                    int type_k, int type_v, const int32_t * slot_block, const int32_t * slot_off, const int32_t * pos,
                    const pkv_rope_params * rope, int n_tok, int n_head_kv, int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(q && k_cur && v_cur && k_pool && v_pool && slot_block && slot_off && pos && rope, "pkv_qkv_prepare: null pointer");
    PKV_REQUIRE(kv_type_ok(type_k) && kv_type_ok(type_v), "pkv_qkv_prepare: KV types must be f16/q8_0/q4_0");
    PKV_REQUIRE(n_tok >= 0 && n_head > 0 && n_head_kv > 0 && block_size > 0 && block_size % 8 == 0, "pkv_qkv_prepare: bad sizes");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_kv_store(q, n_head, k_cur, v_cur, k_pool, v_pool, type_k, type_v, slot_block, slot_off, pos, &r, n_tok,
                           n_head_kv, head_dim, block_size, (hipStream_t) stream);
}

size_t pkv_attn_workspace_bytes(int n_tok, int n_head, int head_dim, int max_seq_len, int block_size) {
    (void) block_size;
    const size_t a = decode_workspace_bytes(n_tok, n_head, head_dim, max_seq_len);
    const size_t b = prefill_workspace_bytes(n_tok, n_head, head_dim);
    return a > b ? a : b;
}

size_t pkv_attn_prefill_workspace_bytes(int n_tok, int n_head, int head_dim) { return prefill_workspace_bytes(n_tok, n_head, head_dim); }
int pkv_attn_decode_tiles(int max_seq_len, int n_head, int n_head_kv, int n_seq) { return decode_tiles_for(max_seq_len, n_head, n_head_kv, n_seq); }

int pkv_attn_workspace_init(void * workspace, size_t bytes, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(workspace != nullptr, "pkv_attn_workspace_init: null workspace");
    PKV_HIP(hipMemsetAsync(workspace, 0, bytes, (hipStream_t) stream));
    return PKV_OK;
}

int pkv_attn_decode(const pkv_attn_args * args, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args != nullptr, "pkv_attn_decode: null args");
    return launch_attn_decode(*args, (hipStream_t) stream);
}

int pkv_rope_cache(const int32_t * pos, int n_tok, int head_dim, const pkv_rope_params * rope, float * cache, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(pos && cache && n_tok >= 0 && head_dim > 0 && head_dim % 2 == 0, "pkv_rope_cache: bad arguments");
    PKV_REQUIRE(((uintptr_t) cache % 16) == 0, "pkv_rope_cache: cache must be 16-byte aligned");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_rope_cache(pos, cache, n_tok, head_dim, r, (hipStream_t) stream);
}

int pkv_attn_decode_step(const pkv_attn_args * args, const pkv_step_args * step, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args != nullptr && step != nullptr, "pkv_attn_decode_step: null args");
    return launch_attn_decode(*args, (hipStream_t) stream, step);
}

int pkv_attn_step_layers(const pkv_attn_args * args, const pkv_step_args * step, int n_layer, const void * const * q, const void * const * k_cur,
                         const void * const * v_cur, void * const * out, const void * const * k_pool, const void * const * v_pool,
                         const int32_t * slot_block, const int32_t * slot_off, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args && step && n_layer >= 0 && q && k_cur && v_cur && out && k_pool && v_pool, "pkv_attn_step_layers: null argument");
    PKV_REQUIRE((slot_block == nullptr) == (slot_off == nullptr), "pkv_attn_step_layers: slot_block and slot_off come together");
    for (int l = 0; l < n_layer; ++l) {
        pkv_attn_args a = *args;
        pkv_step_args s = *step;
        a.q = (const float *) q[l]; a.out = (float *) out[l]; a.k_pool = k_pool[l]; a.v_pool = v_pool[l];
        s.k_cur = (const float *) k_cur[l]; s.v_cur = (const float *) v_cur[l];
        const int rc = slot_block ? pkv_attn_prefill_step(&a, &s, slot_block, slot_off, stream) : pkv_attn_decode_step(&a, &s, stream);
        if (rc != PKV_OK) return rc;
    }
    return PKV_OK;
}

int pkv_attn_decode_stamps(const pkv_attn_args * args, const pkv_step_args * step, uint64_t * stamps, int mode, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args != nullptr && step != nullptr && stamps != nullptr, "pkv_attn_decode_stamps: null args");
    return launch_attn_decode_stamps(*args, step, (unsigned long long *) stamps, mode, (hipStream_t) stream);
}

// raw hipEvent_t helpers for callers without a HIP binding (bench.py measures the decode kernel with them)
void * pkv_event_create(void) {
    hipEvent_t e = nullptr;
    if (ensure_device() != PKV_OK || hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void pkv_event_destroy(void * e) { if (e) (void) hipEventDestroy((hipEvent_t) e); }
int  pkv_event_synchronize(void * e) { PKV_HIP(hipEventSynchronize((hipEvent_t) e)); return PKV_OK; }
int  pkv_event_elapsed_ms(void * start, void * stop, float * ms) {
    PKV_REQUIRE(start && stop && ms, "pkv_event_elapsed_ms: null argument");
    PKV_HIP(hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop));
    return PKV_OK;
}

void pkv_debug_set_prefill_stamps(void * dev_u64) { pkv::set_prefill_stamps(dev_u64); }

int pkv_attn_prefill(const pkv_attn_args * args, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args != nullptr, "pkv_attn_prefill: null args");
    return launch_attn_prefill(*args, (hipStream_t) stream);
}

size_t pkv_flash_attn_ext_flat_workspace_bytes(int n_q, int n_head, int head_dim, int n_kv) { return flat_workspace_bytes(n_q, n_head, head_dim, n_kv); }
int pkv_flash_attn_ext_flat(const pkv_flat_attn_args * args, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args != nullptr, "pkv_flash_attn_ext_flat: null args");
    return launch_attn_flat(*args, (hipStream_t) stream);
}

int pkv_attn_prefill_step(const pkv_attn_args * args, const pkv_step_args * step, const int32_t * slot_block, const int32_t * slot_off, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(args != nullptr && step != nullptr && slot_block && slot_off, "pkv_attn_prefill_step: null args");
    PKV_REQUIRE(step->k_cur && step->v_cur, "pkv_attn_prefill_step: k_cur / v_cur missing");
    hipStream_t st = (hipStream_t) stream;
    RopeDev r;
    const float * cache = nullptr;
    if (step->rope) {
        if (int rc = make_rope(step->rope, args->head_dim, &r)) return rc;
        if (r.mode & 2) { set_error("pkv_attn_prefill_step: NEOX pairs are not fused (use pkv_qkv_prepare + pkv_attn_prefill)"); return PKV_ERR_UNSUPPORTED; }
        PKV_REQUIRE(step->rope_cache != nullptr, "pkv_attn_prefill_step: rope_cache (pkv_rope_cache rows of the ubatch's tokens) is required with rope");
        PKV_REQUIRE(step->pos != nullptr, "pkv_attn_prefill_step: pos missing");
        cache = step->rope_cache;
    }
    // K rotated + K / V stored (the Q plane of pkv_qkv_prepare is what the attention kernel's Q fetch does now), then the attention
    if (int rc = launch_kv_store(nullptr, 0, step->k_cur, step->v_cur, const_cast<void *>(args->k_pool), const_cast<void *>(args->v_pool), args->type_k, args->type_v,
                                 slot_block, slot_off, step->pos, step->rope ? &r : nullptr, args->n_tok, args->n_head_kv, args->head_dim, args->block_size, st, cache)) return rc;
    return launch_attn_prefill(*args, st, cache);
}

int pkv_k_shift(void * k_pool, int type_k, const int32_t * cell_block, const int32_t * cell_off,
                const int32_t * cell_delta, int n_cells, int n_head_kv, int head_dim, int block_size,
                const pkv_rope_params * rope, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(k_pool && cell_block && cell_off && cell_delta && n_cells >= 0, "pkv_k_shift: bad arguments");
    PKV_REQUIRE(kv_type_ok(type_k), "pkv_k_shift: K type must be f16/q8_0/q4_0");
    RopeDev r;
    if (int rc = make_rope(rope, head_dim, &r)) return rc;
    return launch_k_shift(k_pool, type_k, cell_block, cell_off, cell_delta, n_cells, n_head_kv, head_dim, block_size, r,
                          (hipStream_t) stream);
}

int pkv_copy_blocks(void * pool, int type, const int32_t * src, const int32_t * dst, int n, int n_head_kv,
                    int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(pool && src && dst && n >= 0 && kv_type_ok(type), "pkv_copy_blocks: bad arguments");
    return launch_copy_blocks(pool, (size_t) n_head_kv * tile_bytes(type, block_size, head_dim), src, dst, n, (hipStream_t) stream);
}

int pkv_pool_scatter_rows(void * pool, int type, int is_v, const void * rows, const int32_t * slot_block,
                          const int32_t * slot_off, int n_tok, int n_head_kv, int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(pool && rows && slot_block && slot_off && n_tok >= 0 && kv_type_ok(type), "pkv_pool_scatter_rows: bad arguments");
    return launch_rows_io(pool, type, is_v, const_cast<void *>(rows), slot_block, slot_off, n_tok, n_head_kv, head_dim, block_size,
                          true, (hipStream_t) stream);
}
int pkv_pool_scatter_rows_strided(void * pool, int type, int is_v, const void * rows, int64_t rs_tok, int64_t rs_head,
                                  const int32_t * slot_block, const int32_t * slot_off, int n_tok, int n_head_kv,
                                  int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(pool && rows && n_tok >= 0 && kv_type_ok(type) && rs_tok > 0 && rs_head > 0 && (!slot_block || slot_off),
                "pkv_pool_scatter_rows_strided: bad arguments");
    return launch_rows_io(pool, type, is_v, const_cast<void *>(rows), slot_block, slot_off, n_tok, n_head_kv, head_dim, block_size,
                          true, (hipStream_t) stream, rs_tok, rs_head);
}
int pkv_pool_gather_rows(const void * pool, int type, int is_v, void * rows, const int32_t * slot_block,
                         const int32_t * slot_off, int n_tok, int n_head_kv, int head_dim, int block_size, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(pool && rows && slot_block && slot_off && n_tok >= 0 && kv_type_ok(type), "pkv_pool_gather_rows: bad arguments");
    return launch_rows_io(const_cast<void *>(pool), type, is_v, rows, slot_block, slot_off, n_tok, n_head_kv, head_dim, block_size,
                          false, (hipStream_t) stream);
}

// ---- cache ---------------------------------------------------------------------------------------------------
pkv_cache * pkv_cache_create(const pkv_cache_params * p) {
    if (!p || p->n_blocks <= 0 || p->block_size <= 0 || p->n_seq_max <= 0) { set_error("pkv_cache_create: bad params"); return nullptr; }
    if (p->n_layer > 0) {
        if (ensure_device() != PKV_OK) return nullptr;
        if (!kv_type_ok(p->type_k) || !kv_type_ok(p->type_v) || p->n_head_kv <= 0 || (p->head_dim != 64 && p->head_dim != 128)) {
            set_error("pkv_cache_create: unsupported KV geometry/types");
            return nullptr;
        }
    }
    pkv_cache * c = new pkv_cache(*p);
    if (p->n_layer > 0) {
        if (hipSetDevice(p->device) != hipSuccess) { hip_fail(hipGetLastError(), "hipSetDevice"); delete c; return nullptr; }
        const size_t kb = pkv_pool_bytes(p->type_k, p->n_blocks, p->n_head_kv, p->block_size, p->head_dim);
        const size_t vb = pkv_pool_bytes(p->type_v, p->n_blocks, p->n_head_kv, p->block_size, p->head_dim);
        for (int l = 0; l < p->n_layer; ++l) {
            void * k = nullptr, * v = nullptr;
            // zero-filled like ggml_backend_buffer_clear at init (/ref/src/llama-kv-cache.cpp:118)
            if (hipMalloc(&k, kb) != hipSuccess || hipMalloc(&v, vb) != hipSuccess || hipMemset(k, 0, kb) != hipSuccess ||
                hipMemset(v, 0, vb) != hipSuccess) {
                hip_fail(hipGetLastError(), "pool allocation");
                if (k) (void) hipFree(k);
                if (v) (void) hipFree(v);
                pkv_cache_free(c);
                return nullptr;
            }
            c->dev.k.push_back(k);
            c->dev.v.push_back(v);
        }
    }
    return c;
}

void pkv_cache_free(pkv_cache * c) {
    if (!c) return;
    for (void * p : c->dev.k) (void) hipFree(p);
    for (void * p : c->dev.v) (void) hipFree(p);
    if (c->dev.scratch) (void) hipFree(c->dev.scratch);
    delete c;
}

int32_t pkv_cache_n_tokens(const pkv_cache * c) { return c ? c->host.get_n_tokens() : 0; }
int32_t pkv_cache_used_cells(const pkv_cache * c) { return c ? c->host.get_used_cells() : 0; }
void    pkv_cache_clear(pkv_cache * c) { if (c) c->host.clear(); }
int     pkv_cache_seq_rm(pkv_cache * c, int32_t s, int32_t p0, int32_t p1) { return c ? c->host.seq_rm(s, p0, p1) : 1; }
void    pkv_cache_seq_cp(pkv_cache * c, int32_t a, int32_t b, int32_t p0, int32_t p1) { if (c) c->host.seq_cp(a, b, p0, p1); }
void    pkv_cache_seq_keep(pkv_cache * c, int32_t s) { if (c) c->host.seq_keep(s); }
void    pkv_cache_seq_add(pkv_cache * c, int32_t s, int32_t p0, int32_t p1, int32_t d) { if (c) c->host.seq_add(s, p0, p1, d); }
void    pkv_cache_seq_div(pkv_cache * c, int32_t s, int32_t p0, int32_t p1, int32_t d) { if (c) c->host.seq_div(s, p0, p1, d); }
int32_t pkv_cache_seq_pos_max(pkv_cache * c, int32_t s) { return c ? c->host.seq_pos_max(s) : 0; }
void    pkv_cache_defrag(pkv_cache * c) { if (c) c->host.defrag(); }
int     pkv_cache_can_shift(const pkv_cache * c) { return c ? c->host.get_can_shift() : 0; }

int pkv_cache_find_slot(pkv_cache * c, int n_tok, const int32_t * seq_id, const int32_t * pos, int32_t * out_block,
                        int32_t * out_off) {
    PKV_REQUIRE(c && n_tok >= 0 && (n_tok == 0 || (seq_id && pos && out_block && out_off)), "pkv_cache_find_slot: bad arguments");
    const int rc = c->host.find_slot(n_tok, seq_id, pos, out_block, out_off);
    if (rc == PKV_ERR_NO_SPACE) set_error("pkv_cache_find_slot: failed to find a slot for %d tokens (%d free blocks)", n_tok, c->host.n_free_blocks());
    if (rc == PKV_ERR_INVALID) set_error("pkv_cache_find_slot: seq_id out of range");
    if (rc == PKV_OK) {   // the slot mapping goes straight into device kernels that only test blk < 0: never hand out an id outside the pool
        const int32_t nb = c->host.params().n_blocks, bs = c->host.params().block_size;
        for (int t = 0; t < n_tok; ++t)
            if (out_block[t] < 0 || out_block[t] >= nb || out_off[t] < 0 || out_off[t] >= bs) {
                set_error("pkv_cache_find_slot: internal error, slot (%d, %d) of token %d lies outside the pool", out_block[t], out_off[t], t);
                return PKV_ERR_INVALID;
            }
    }
    return rc;
}

int32_t pkv_cache_seq_len(const pkv_cache * c, int32_t s) { return c ? c->host.seq_len(s) : 0; }
int32_t pkv_cache_seq_n_blocks(const pkv_cache * c, int32_t s) { return c && c->host.valid(s) ? (int32_t) c->host.seq_blocks(s).size() : 0; }
int32_t pkv_cache_seq_blocks(const pkv_cache * c, int32_t s, int32_t * out, int32_t cap) {
    if (!c || !c->host.valid(s)) return 0;
    const auto & b = c->host.seq_blocks(s);
    for (int32_t i = 0; i < (int32_t) b.size() && i < cap; ++i) out[i] = b[i];
    return (int32_t) b.size();
}
int32_t pkv_cache_seq_is_regular(const pkv_cache * c, int32_t s) { return c ? c->host.seq_is_regular(s) : 1; }
int32_t pkv_cache_block_tables(const pkv_cache * c, int32_t n_seq, const int32_t * seq_ids, int32_t width, int32_t * table, int32_t * lens, int32_t * all_regular) {
    if (!c || n_seq < 0 || width <= 0 || !seq_ids || !table || !lens) return PKV_ERR_INVALID;
    int32_t mb = 0, reg = 1;
    for (int32_t i = 0; i < n_seq; ++i) {
        const int32_t s = seq_ids[i];
        int32_t * row = table + (size_t) i * width;
        int32_t nb = 0;
        if (c->host.valid(s)) {
            const auto & b = c->host.seq_blocks(s);
            nb = (int32_t) b.size();
            if (nb > width) { set_error("pkv_cache_block_tables: sequence %d has %d blocks, table width %d", s, nb, width); return PKV_ERR_INVALID; }
            for (int32_t j = 0; j < nb; ++j) row[j] = b[j];
        }
        for (int32_t j = nb; j < width; ++j) row[j] = 0;
        lens[i] = c->host.seq_len(s);
        reg &= c->host.seq_is_regular(s) ? 1 : 0;
        if (nb > mb) mb = nb;
    }
    if (all_regular) *all_regular = reg;
    return mb;
}
int32_t pkv_cache_seq_slot_pos(const pkv_cache * c, int32_t s, int32_t * out, int32_t cap) {
    if (!c || !c->host.valid(s)) return 0;
    const auto & p = c->host.seq_slot_pos(s);
    for (int32_t i = 0; i < (int32_t) p.size() && i < cap; ++i) out[i] = p[i];
    return (int32_t) p.size();
}
int32_t pkv_cache_n_free_blocks(const pkv_cache * c) { return c ? c->host.n_free_blocks() : 0; }
int32_t pkv_cache_block_refcount(const pkv_cache * c, int32_t b) { return c ? c->host.refcount(b) : -1; }
int32_t pkv_cache_has_shift(const pkv_cache * c) { return c ? c->host.has_shift() : 0; }
int32_t pkv_cache_n_pending(const pkv_cache * c) { return c ? (int32_t) c->host.pending.size() : 0; }
int32_t pkv_cache_pending(const pkv_cache * c, int32_t * out5, int32_t cap) {
    if (!c) return 0;
    const auto & q = c->host.pending;
    for (int32_t i = 0; i < (int32_t) q.size() && i < cap; ++i) {
        out5[i * 5 + 0] = q[i].cell; out5[i * 5 + 1] = q[i].src_block; out5[i * 5 + 2] = q[i].src_off;
        out5[i * 5 + 3] = q[i].dst_block; out5[i * 5 + 4] = q[i].dst_off;
    }
    return (int32_t) q.size();
}

void * pkv_cache_k_pool(pkv_cache * c, int layer) { return c && layer >= 0 && layer < (int) c->dev.k.size() ? c->dev.k[layer] : nullptr; }
void * pkv_cache_v_pool(pkv_cache * c, int layer) { return c && layer >= 0 && layer < (int) c->dev.v.size() ? c->dev.v[layer] : nullptr; }

static int upload(pkv_cache * c, const std::vector<int32_t> & h, hipStream_t st) {
    if (h.size() > c->dev.scratch_n) {
        if (c->dev.scratch) { PKV_HIP(hipStreamSynchronize(st)); PKV_HIP(hipFree(c->dev.scratch)); c->dev.scratch = nullptr; }
        c->dev.scratch_n = h.size() * 2 + 1024;
        PKV_HIP(hipMalloc((void **) &c->dev.scratch, c->dev.scratch_n * sizeof(int32_t)));
    }
    PKV_HIP(hipMemcpyAsync(c->dev.scratch, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    PKV_HIP(hipStreamSynchronize(st));  // h is pageable and dies with the caller's frame
    return PKV_OK;
}

int pkv_cache_update(pkv_cache * c, const pkv_rope_params * rope, void * stream) {
    PKV_REQUIRE(c != nullptr, "pkv_cache_update: null cache");
    const pkv_cache_params & p = c->host.params();
    hipStream_t st = (hipStream_t) stream;
    auto & q = c->host.pending;
    if (p.n_layer <= 0) {  // bookkeeping-only cache: nothing to run, just retire the queues
        q.clear();
        std::vector<int32_t> a, b, d;
        c->host.take_shifts(a, b, d);
        return PKV_OK;
    }
    if (int rc = ensure_device()) return rc;
    // (1) queued copies, in issue order, batched per run of equal kind
    size_t i = 0;
    while (i < q.size()) {
        size_t j = i;
        while (j < q.size() && q[j].cell == q[i].cell) ++j;
        const int n = (int) (j - i);
        std::vector<int32_t> h((size_t) n * 4);
        for (int k = 0; k < n; ++k) {
            h[k] = q[i + k].src_block; h[n + k] = q[i + k].src_off; h[2 * n + k] = q[i + k].dst_block; h[3 * n + k] = q[i + k].dst_off;
        }
        if (int rc = upload(c, h, st)) return rc;
        const int32_t * sb = c->dev.scratch, * so = sb + n, * db = so + n, * dof = db + n;
        for (int l = 0; l < p.n_layer; ++l) {
            int rc;
            if (!q[i].cell) {
                rc = launch_copy_blocks(c->dev.k[l], (size_t) p.n_head_kv * tile_bytes(p.type_k, p.block_size, p.head_dim), sb, db, n, st);
                if (!rc) rc = launch_copy_blocks(c->dev.v[l], (size_t) p.n_head_kv * tile_bytes(p.type_v, p.block_size, p.head_dim), sb, db, n, st);
            } else {
                rc = launch_copy_cells(c->dev.k[l], p.type_k, 0, sb, so, db, dof, n, p.n_head_kv, p.head_dim, p.block_size, st);
                if (!rc) rc = launch_copy_cells(c->dev.v[l], p.type_v, 1, sb, so, db, dof, n, p.n_head_kv, p.head_dim, p.block_size, st);
            }
            if (rc) return rc;
        }
        PKV_HIP(hipStreamSynchronize(st));  // scratch is reused by the next run
        i = j;
    }
    q.clear();
    // (2) pending K-shift (/ref/src/llama-context.cpp:719-760)
    if (c->host.has_shift()) {
        PKV_REQUIRE(rope != nullptr, "pkv_cache_update: a K-shift is pending but no rope parameters were given");
        RopeDev r;
        if (int rc = make_rope(rope, p.head_dim, &r)) return rc;
        std::vector<int32_t> blk, off, dlt;
        c->host.take_shifts(blk, off, dlt);
        const int n = (int) blk.size();
        if (n > 0) {
            std::vector<int32_t> h;
            h.insert(h.end(), blk.begin(), blk.end());
            h.insert(h.end(), off.begin(), off.end());
            h.insert(h.end(), dlt.begin(), dlt.end());
            if (int rc = upload(c, h, st)) return rc;
            for (int l = 0; l < p.n_layer; ++l)
                if (int rc = launch_k_shift(c->dev.k[l], p.type_k, c->dev.scratch, c->dev.scratch + n, c->dev.scratch + 2 * n, n,
                                            p.n_head_kv, p.head_dim, p.block_size, r, st))
                    return rc;
            PKV_HIP(hipStreamSynchronize(st));
        }
    }
    return PKV_OK;
}

// ---- first slice of the rest of the decoder layer ---------------------------------------------------------------
int pkv_rms_norm_f32(const float * x, float * y, int64_t n_rows, int64_t ne0, float eps, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && y && n_rows >= 0 && ne0 > 0 && eps >= 0.0f, "pkv_rms_norm_f32: bad arguments");
    return launch_rms_norm(x, y, n_rows, ne0, eps, (hipStream_t) stream);
}
int pkv_rms_norm_mul_f32(const float * x, const float * w, float * y_norm, float * y, int64_t n_rows, int64_t ne0, float eps, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && w && y && n_rows >= 0 && ne0 > 0 && eps >= 0.0f, "pkv_rms_norm_mul_f32: bad arguments");
    return launch_rms_norm_mul(x, nullptr, nullptr, w, y_norm, y, n_rows, ne0, eps, (hipStream_t) stream);
}
int pkv_add_rms_norm_mul_f32(const float * a, const float * b, float * sum, const float * w, float * y_norm, float * y, int64_t n_rows, int64_t ne0, float eps,
                             void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(a && b && sum && w && y && n_rows >= 0 && ne0 > 0 && eps >= 0.0f, "pkv_add_rms_norm_mul_f32: bad arguments");
    return launch_rms_norm_mul(a, b, sum, w, y_norm, y, n_rows, ne0, eps, (hipStream_t) stream);
}
int pkv_silu_mul_f32(const float * g, const float * u, float * y_silu, float * y, int64_t n, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(g && u && y && n >= 0, "pkv_silu_mul_f32: bad arguments");
    return launch_silu_mul(g, u, y_silu, y, n, (hipStream_t) stream);
}
int pkv_add_f32(const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(a && b && y && n_rows >= 0 && ne0 > 0 && b_rows > 0 && n_rows % b_rows == 0, "pkv_add_f32: bad arguments");
    return launch_binary(0, a, b, y, n_rows, ne0, b_rows, (hipStream_t) stream);
}
int pkv_mul_f32(const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(a && b && y && n_rows >= 0 && ne0 > 0 && b_rows > 0 && n_rows % b_rows == 0, "pkv_mul_f32: bad arguments");
    return launch_binary(1, a, b, y, n_rows, ne0, b_rows, (hipStream_t) stream);
}
int pkv_silu_f32(const float * x, float * y, int64_t n, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && y && n >= 0, "pkv_silu_f32: bad arguments");
    return launch_silu(x, y, n, (hipStream_t) stream);
}
int pkv_cpy_strided(const void * src, int src_type, const int64_t * ne_src, const int64_t * nb_src, void * dst, int dst_type, const int64_t * ne_dst, const int64_t * nb_dst,
                    void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(src && dst && ne_src && nb_src && ne_dst && nb_dst, "pkv_cpy_strided: null pointer");
    return launch_cpy_strided(src, src_type, ne_src, nb_src, dst, dst_type, ne_dst, nb_dst, (hipStream_t) stream);
}
int pkv_mul_mat_strided(const void * w, int type_w, const int64_t * ne_w, const int64_t * nb_w, const float * x, const int64_t * ne_x, const int64_t * nb_x, float * y,
                        void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(w && x && y && ne_w && nb_w && ne_x && nb_x, "pkv_mul_mat_strided: null pointer");
    return launch_mul_mat_strided(w, type_w, ne_w, nb_w, x, ne_x, nb_x, y, (hipStream_t) stream);
}
int pkv_soft_max_ext_f32(const float * x, const void * mask, int mask_f16, float * y, int64_t ne00, int64_t ne01, int64_t ne02, int64_t ne03, float scale,
                         float max_bias, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && y && ne00 > 0 && ne01 >= 0 && ne02 > 0 && ne03 > 0 && max_bias >= 0.0f, "pkv_soft_max_ext_f32: bad arguments");
    return launch_soft_max(x, mask, mask_f16, y, ne00, ne01, ne02, ne03, scale, max_bias, (hipStream_t) stream);
}
static bool packed_type_ok(int type_w) { const int bt = type_w & ~PKV_TYPE_PACKED; return type_w == bt || bt == PKV_TYPE_Q4_K || bt == PKV_TYPE_Q5_K || bt == PKV_TYPE_Q6_K; }
int pkv_mul_mat(const void * w, int type_wf, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(w && x && y && M >= 0 && K > 0 && n_cols >= 0 && packed_type_ok(type_wf), "pkv_mul_mat: bad arguments");
    const int type_w = type_wf & ~PKV_TYPE_PACKED;
    PKV_REQUIRE(K % (type_w == PKV_TYPE_F32 ? 4 : (type_w == PKV_TYPE_F16 ? 8 : ((type_w == PKV_TYPE_Q4_K || type_w == PKV_TYPE_Q5_K || type_w == PKV_TYPE_Q6_K) ? 256 : PKV_QK))) == 0 &&
                ((uintptr_t) x % 16) == 0 && ((uintptr_t) w % ((type_w == PKV_TYPE_Q8_0 || type_w == PKV_TYPE_Q6_K) ? 2 : 16)) == 0, "pkv_mul_mat: K / alignment not supported");
    return launch_mul_mat(w, type_wf, x, y, M, K, n_cols, (hipStream_t) stream);
}
int pkv_mul_mat_multi(int n, const void * const * w, int type_wf, const float * x, float * const * y, const int64_t * M, int64_t K, int64_t n_cols,
                      void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(n >= 1 && n <= PKV_MUL_MAT_MAX && w && x && y && M && K > 0 && n_cols >= 0 && packed_type_ok(type_wf), "pkv_mul_mat_multi: bad arguments");
    const int type_w = type_wf & ~PKV_TYPE_PACKED;
    PKV_REQUIRE(K % (type_w == PKV_TYPE_F32 ? 4 : (type_w == PKV_TYPE_F16 ? 8 : (type_w == PKV_TYPE_Q4_K ? 256 : PKV_QK))) == 0 && ((uintptr_t) x % 16) == 0,
                "pkv_mul_mat_multi: K / alignment not supported");
    for (int i = 0; i < n; ++i)
        PKV_REQUIRE(w[i] && y[i] && M[i] >= 0 && ((uintptr_t) w[i] % (type_w == PKV_TYPE_Q8_0 ? 2 : 16)) == 0, "pkv_mul_mat_multi: bad product %d", i);
    return launch_mul_mat_multi(n, w, type_wf, x, y, M, K, n_cols, (hipStream_t) stream);
}

size_t pkv_act_image_bytes(int64_t K) { return act_image_bytes(K); }
int pkv_quantize_act(const float * x, void * image, int64_t K, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && image && ((uintptr_t) x % 16) == 0 && ((uintptr_t) image % 16) == 0, "pkv_quantize_act: null or misaligned pointer");
    return launch_quantize_act(x, image, K, (hipStream_t) stream);
}
int pkv_rms_norm_mul_q(const float * x, const float * res, float * x_out, const float * w, float * y, void * image, int64_t ne0, float eps, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(x && w && image && ((res == nullptr) == (x_out == nullptr)) && eps >= 0.0f, "pkv_rms_norm_mul_q: bad arguments");
    PKV_REQUIRE(((uintptr_t) x % 16) == 0 && ((uintptr_t) w % 16) == 0 && ((uintptr_t) image % 16) == 0, "pkv_rms_norm_mul_q: pointers must be 16-byte aligned");
    return launch_rms_norm_mul_q(x, res, x_out, w, y, image, ne0, eps, (hipStream_t) stream);
}
int pkv_gemv_q(const pkv_gemv * g, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(g != nullptr, "pkv_gemv_q: null args");
    for (int i = 0; i < g->n && i < PKV_GEMV_MAX; ++i)
        PKV_REQUIRE(((uintptr_t) g->w[i] % 16) == 0, "pkv_gemv_q: weight matrix %d must be 16-byte aligned", i);
    return launch_gemv(*g, (hipStream_t) stream);
}
int pkv_weights_pack(const void * src, void * dst, int type, int64_t n_rows, int64_t K, int inverse, void * stream) {
    if (int rc = ensure_device()) return rc;
    return launch_pack_rows(src, dst, type, n_rows, K, inverse, (hipStream_t) stream);
}
int pkv_get_rows(const void * w, int type_w, const int32_t * ids, float * out, int64_t n_ids, int64_t K, int64_t n_rows_w, void * stream) {
    if (int rc = ensure_device()) return rc;
    PKV_REQUIRE(w && ids && out && n_ids >= 0 && K > 0 && n_rows_w > 0, "pkv_get_rows: bad arguments");
    { const int bt = type_w & ~PKV_TYPE_PACKED;
      PKV_REQUIRE(packed_type_ok(type_w) && K % (bt == PKV_TYPE_Q4_K || bt == PKV_TYPE_Q5_K || bt == PKV_TYPE_Q6_K ? 256 : (bt == PKV_TYPE_Q8_0 ? 32 : 1)) == 0, "pkv_get_rows: K does not fill whole blocks"); }
    return launch_get_rows(w, type_w, ids, out, n_ids, K, n_rows_w, (hipStream_t) stream);
}

// ---- session state: llama_kv_cache_unified::state_write / state_read in the reference's byte format ------------------------
// (/ref/src/llama-kv-cache.cpp:848-1221)   u32 cell_count | per cell: i32 pos, u32 n_seq_id, i32 seq_id[n_seq_id] |
//   u32 v_trans | u32 n_layer | per layer: i32 type_k, u64 k_size_row, K rows | per layer: i32 type_v, u64 v_size_row, V rows
// v_trans is written as 0 (the paged path is the flash-attention path, whose V cache is not transposed, :94); a file with
// v_trans = 1 (f16 V, `-fa 0` reference runs) is accepted and transposed back on the host.

int64_t pkv_cache_state_size(pkv_cache * c, int32_t seq_id) {
    if (!c) { set_error("pkv_cache_state_size: null cache"); return PKV_ERR_INVALID; }
    std::vector<KvCachePaged::StateCell> cells;
    if (c->host.state_cells(seq_id, cells) != PKV_OK) { set_error("pkv_cache_state_size: device work is pending, call pkv_cache_update first"); return PKV_ERR_INVALID; }
    return (int64_t) state_layout(c->host.params(), cells).total;
}

int64_t pkv_cache_state_write(pkv_cache * c, int32_t seq_id, void * dst, size_t cap, void * stream) {
    if (!c || !dst) { set_error("pkv_cache_state_write: null argument"); return PKV_ERR_INVALID; }
    const pkv_cache_params & p = c->host.params();
    hipStream_t st = (hipStream_t) stream;
    std::vector<KvCachePaged::StateCell> cells;
    if (c->host.state_cells(seq_id, cells) != PKV_OK) { set_error("pkv_cache_state_write: device work is pending, call pkv_cache_update first"); return PKV_ERR_INVALID; }
    const StateLayout L = state_layout(p, cells);
    if (cap < L.total) { set_error("pkv_cache_state_write: buffer too small (%zu < %zu)", cap, L.total); return PKV_ERR_INVALID; }
    char * w = (char *) dst;
    put<uint32_t>(w, (uint32_t) L.n);
    for (const auto & cell : cells) {                             // state_write_meta (:904-921)
        put<int32_t>(w, cell.pos);
        put<uint32_t>(w, (uint32_t) cell.seq_ids.size());
        for (int32_t id : cell.seq_ids) put<int32_t>(w, id);
    }
    put<uint32_t>(w, 0u);                                         // v_trans
    put<uint32_t>(w, (uint32_t) p.n_layer);
    if (p.n_layer > 0) {                                          // state_write_data (:923-1001): gather-on-save through the block table
        if (int rc = ensure_device()) return rc;
        std::vector<int32_t> h(2 * L.n + 2);
        for (size_t i = 0; i < L.n; ++i) { h[i] = cells[i].block; h[L.n + i] = cells[i].off; }
        if (int rc = upload(c, h, st)) return rc;
        void * rows = nullptr;
        const size_t rb = std::max<size_t>(16, L.n * std::max(L.k_row, L.v_row));
        PKV_HIP(hipMalloc(&rows, rb));
        int rc = PKV_OK;
        for (int v = 0; v < 2 && rc == PKV_OK; ++v) {
            const int type = v ? p.type_v : p.type_k;
            const size_t row = v ? L.v_row : L.k_row;
            for (int l = 0; l < p.n_layer && rc == PKV_OK; ++l) {
                put<int32_t>(w, (int32_t) type);                  // ggml_type ids: F16 1, Q4_0 2, Q8_0 8 == PKV_TYPE_*
                put<uint64_t>(w, (uint64_t) row);
                if (L.n == 0) continue;
                rc = launch_rows_io(v ? c->dev.v[l] : c->dev.k[l], type, v, rows, c->dev.scratch, c->dev.scratch + L.n, (int) L.n,
                                    p.n_head_kv, p.head_dim, p.block_size, false, st);
                if (rc == PKV_OK && hipMemcpyAsync(w, rows, L.n * row, hipMemcpyDeviceToHost, st) != hipSuccess) rc = hip_fail(hipGetLastError(), "state D2H");
                if (rc == PKV_OK && hipStreamSynchronize(st) != hipSuccess) rc = hip_fail(hipGetLastError(), "state sync");
                w += L.n * row;
            }
        }
        (void) hipFree(rows);
        if (rc != PKV_OK) return rc;
    }
    return (int64_t) (w - (char *) dst);
}

int pkv_cache_state_read(pkv_cache * c, int32_t seq_id, const void * src, size_t n_bytes, void * stream) {
    PKV_REQUIRE(c && src, "pkv_cache_state_read: null argument");
    const pkv_cache_params & p = c->host.params();
    hipStream_t st = (hipStream_t) stream;
    const char * r = (const char *) src, * end = r + n_bytes;
    // failure after the cache was touched leaves it as the reference does (:888-896): cleared / the sequence removed
    auto fail = [&](int code, const char * what) {
        if (seq_id == -1) c->host.clear(); else c->host.seq_rm(seq_id, -1, -1);
        set_error("pkv_cache_state_read: failed to restore kv cache (%s)", what);
        return code;
    };
    uint32_t cell_count = 0;
    if (!get(r, end, &cell_count)) return fail(PKV_ERR_INVALID, "truncated");
    if ((uint64_t) cell_count > (uint64_t) p.n_blocks * p.block_size) return fail(PKV_ERR_NO_SPACE, "not enough cells in kv cache");
    std::vector<int32_t> pos(cell_count);
    std::vector<std::vector<int32_t>> ids(cell_count);
    for (uint32_t i = 0; i < cell_count; ++i) {
        uint32_t n_seq_id = 0;
        if (!get(r, end, &pos[i]) || !get(r, end, &n_seq_id) || (size_t) (end - r) < (size_t) n_seq_id * 4) return fail(PKV_ERR_INVALID, "truncated");
        ids[i].resize(n_seq_id);
        for (uint32_t j = 0; j < n_seq_id; ++j) get(r, end, &ids[i][j]);
        if (seq_id != -1 && n_seq_id != 0) return fail(PKV_ERR_INVALID, "invalid seq_id-agnostic kv cell");
    }
    std::vector<KvCachePaged::StateTarget> tg;
    const int rc0 = c->host.state_restore(seq_id, pos, ids, tg);
    if (rc0 == PKV_ERR_NO_SPACE) return fail(rc0, "failed to find available cells in kv cache");
    if (rc0 != PKV_OK) return fail(rc0, "seq_id out of range");
    uint32_t v_trans = 0, n_layer = 0;
    if (!get(r, end, &v_trans) || !get(r, end, &n_layer)) return fail(PKV_ERR_INVALID, "truncated");
    if ((int32_t) n_layer != p.n_layer) return fail(PKV_ERR_INVALID, "mismatched layer count");
    if (v_trans > 1) return fail(PKV_ERR_INVALID, "incompatible V transposition");
    if (n_layer == 0) return PKV_OK;
    if (int rc = ensure_device()) return fail(rc, "no device");
    // find_slot may have queued a copy-on-write for a shared tail block: run it before the rows land
    if (!c->host.pending.empty()) { if (int rc = pkv_cache_update(c, nullptr, stream)) return fail(rc, "pending copies"); }
    const size_t nt = tg.size();
    const size_t k_row = row_size(p.type_k, (int64_t) p.n_head_kv * p.head_dim), v_row = row_size(p.type_v, (int64_t) p.n_head_kv * p.head_dim);
    std::vector<int32_t> h(2 * nt + 2);
    for (size_t i = 0; i < nt; ++i) { h[i] = tg[i].block; h[nt + i] = tg[i].off; }
    if (int rc = upload(c, h, st)) return fail(rc, "upload");
    void * rows = nullptr;
    if (hipMalloc(&rows, std::max<size_t>(16, nt * std::max(k_row, v_row))) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipMalloc"), "allocation");
    std::vector<char> stage;
    int rc = PKV_OK;
    const char * why = "";
    for (int v = 0; v < 2 && rc == PKV_OK; ++v) {
        const int type = v ? p.type_v : p.type_k;
        const size_t row = v ? v_row : k_row;
        const bool trans = v && v_trans;
        for (uint32_t l = 0; l < n_layer && rc == PKV_OK; ++l) {
            int32_t type_ref = 0;
            if (!get(r, end, &type_ref)) { rc = PKV_ERR_INVALID; why = "truncated"; break; }
            if (type_ref != type) { rc = PKV_ERR_INVALID; why = v ? "mismatched value type" : "mismatched key type"; break; }
            if (!trans) {
                uint64_t row_ref = 0;
                if (!get(r, end, &row_ref)) { rc = PKV_ERR_INVALID; why = "truncated"; break; }
                if (row_ref != row) { rc = PKV_ERR_INVALID; why = v ? "mismatched value row size" : "mismatched key row size"; break; }
            } else {                                              // transposed V (:1176-1215): element size + n_embd_v_gqa, then [n_embd][cells]
                uint32_t el = 0, n_embd = 0;
                if (!get(r, end, &el) || !get(r, end, &n_embd)) { rc = PKV_ERR_INVALID; why = "truncated"; break; }
                if (type != PKV_TYPE_F16 || el != 2) { rc = PKV_ERR_INVALID; why = "mismatched value element size"; break; }
                if (n_embd != (uint32_t) (p.n_head_kv * p.head_dim)) { rc = PKV_ERR_INVALID; why = "mismatched GQA embedding size"; break; }
            }
            const size_t bytes = (size_t) cell_count * row;
            if ((size_t) (end - r) < bytes) { rc = PKV_ERR_INVALID; why = "truncated"; break; }
            if (nt > 0) {
                stage.resize(nt * row);
                if (!trans) {
                    for (size_t i = 0; i < nt; ++i) memcpy(&stage[i * row], r + (size_t) tg[i].cell * row, row);
                } else {
                    const uint16_t * tv = (const uint16_t *) r;   // [n_embd][cell_count]
                    const size_t ne = row / 2;
                    for (size_t i = 0; i < nt; ++i)
                        for (size_t j = 0; j < ne; ++j) ((uint16_t *) &stage[i * row])[j] = tv[j * cell_count + (size_t) tg[i].cell];
                }
                if (hipMemcpyAsync(rows, stage.data(), nt * row, hipMemcpyHostToDevice, st) != hipSuccess) { rc = hip_fail(hipGetLastError(), "state H2D"); why = "copy"; break; }
                rc = launch_rows_io(v ? c->dev.v[l] : c->dev.k[l], type, v, rows, c->dev.scratch, c->dev.scratch + nt, (int) nt, p.n_head_kv,
                                    p.head_dim, p.block_size, true, st);
                if (rc == PKV_OK && hipStreamSynchronize(st) != hipSuccess) { rc = hip_fail(hipGetLastError(), "state sync"); why = "sync"; }
            }
            r += bytes;
        }
    }
    (void) hipFree(rows);
    if (rc != PKV_OK) return fail(rc, why);
    return PKV_OK;
}

int32_t pkv_cache_view_update(const pkv_cache * c, int32_t n_seq_max, int32_t * cell_pos, int32_t * cell_seqs, int32_t * stats) {
    if (!c || n_seq_max < 0) return 0;
    const pkv_cache_params & p = c->host.params();
    if (cell_pos || cell_seqs || stats) c->host.view(n_seq_max, cell_pos, cell_seqs, stats);
    return p.n_blocks * p.block_size;
}

// ---- fork prototype toys ---------------------------------------------------------------------------------------------
pkv_toy_block_manager * pkv_toy_bm_create(int n) { return new pkv_toy_block_manager(n); }
void pkv_toy_bm_free(pkv_toy_block_manager * bm) { delete bm; }
int  pkv_toy_bm_allocate(pkv_toy_block_manager * bm) { return bm ? bm->bm.allocateBlock() : -1; }
void pkv_toy_bm_release(pkv_toy_block_manager * bm, int b) { if (bm) bm->bm.freeBlock(b); }
int  pkv_toy_bm_status(const pkv_toy_block_manager * bm, uint8_t * out, int cap) {
    if (!bm) return 0;
    const auto & s = bm->bm.getBlockStatus();
    for (int i = 0; i < (int) s.size() && i < cap; ++i) out[i] = s[i];
    return (int) s.size();
}
pkv_toy_kv_manager * pkv_toy_kvm_create(int cap) { return new pkv_toy_kv_manager(cap); }
void pkv_toy_kvm_free(pkv_toy_kv_manager * m) { delete m; }
int  pkv_toy_kvm_insert(pkv_toy_kv_manager * m, int * b, int * s) { return m && m->m.insert(b, s) ? 0 : -1; }
int  pkv_toy_kvm_release(pkv_toy_kv_manager * m, int b, int s) { return m && m->m.release(b, s) ? 0 : -1; }

pkv_toy_block_kv_cache * pkv_toy_bkc_create(int nb, int bs, int h, int d) { return nb > 0 && bs > 0 && h > 0 && d > 0 ? new pkv_toy_block_kv_cache(nb, bs, h, d) : nullptr; }
void pkv_toy_bkc_free(pkv_toy_block_kv_cache * c) { delete c; }
void pkv_toy_bkc_init_cache(pkv_toy_block_kv_cache * c) { if (c) c->c.initCache(); }
void pkv_toy_bkc_write_to_cache(pkv_toy_block_kv_cache * c, const float * k, const float * v, const int * m, int n) {
    if (c) c->c.writeToCache(k, v, std::vector<int>(m, m + n));
}
void pkv_toy_bkc_swap_blocks(pkv_toy_block_kv_cache * c, pkv_toy_block_kv_cache * o, const int * m, int n) {
    if (c && o) c->c.swapBlocks(o->c, std::vector<int>(m, m + n));
}
void pkv_toy_bkc_copy_blocks(pkv_toy_block_kv_cache * c, const pkv_toy_block_kv_cache * src, const int * m, int n) {
    if (c && src) c->c.copyBlocks(src->c, std::vector<int>(m, m + n));
}
int pkv_toy_bkc_insert_kv_block(pkv_toy_block_kv_cache * c, const float * k, const float * v, int * pb, int * slot) {
    return c && c->c.insertKVBlock(k, v, pb, slot) ? 0 : -1;
}
int pkv_toy_bkc_free_kv_block(pkv_toy_block_kv_cache * c, int pb, int slot) { return c && c->c.freeKVBlock(pb, slot) ? 0 : -1; }
int pkv_toy_bkc_get(const pkv_toy_block_kv_cache * c, int which_v, float * out, int cap) {
    if (!c) return 0;
    const auto & x = which_v ? c->c.getValueCache() : c->c.getKeyCache();
    for (int i = 0; i < (int) x.size() && i < cap; ++i) out[i] = x[i];
    return (int) x.size();
}
void pkv_toy_attention_fwd(const float * q, int n, pkv_toy_block_kv_cache * c, const float * k, const float * v, const int * m,
                           int nm, float * out) {
    if (c) toyBlockKVAttentionFwd(q, (size_t) n, c->c, k, v, std::vector<int>(m, m + nm), out);
}
void pkv_toy_attention_paged_fwd(const float * q, int n, pkv_toy_block_kv_cache * c, const float * k, const float * v, float * out,
                                 int page) {
    if (c && page > 0) toyBlockKVAttentionPagedFwd(q, (size_t) n, c->c, k, v, out, (size_t) page);
}
void pkv_toy_attention_real_fwd(const float * q, int n, const float * k, const float * v, float * out) {
    toyBlockKVAttentionRealFwd(q, (size_t) n, k, v, out);
}

}  // extern "C"
