// oracle/ref_driver.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI shim (our own code) over the REFERENCE's compiled CPU ggml path
// (oracle/_ref/libggml_ref.so, built in place from /root/reference by oracle/Makefile).
// It builds one-op graphs through the reference's public API (ggml/include/ggml.h) and runs
// them with ggml_graph_compute_with_ctx (ggml/include/ggml-cpu.h:71), i.e. exactly the kernels
// the reference's CPU backend runs inside llama_decode:
//   ref_flash_attn_ext -> ggml_compute_forward_flash_attn_ext_f16 (ggml/src/ggml-cpu/ggml-cpu.c:12219)
//   ref_rope_ext       -> ggml_compute_forward_rope_f32            (ggml/src/ggml-cpu/ggml-cpu.c:10669)
//   ref_cpy            -> ggml_compute_forward_dup_f32 / from_float (ggml/src/ggml-cpu/ggml-cpu.c:3742)
//   ref_soft_max_ext   -> ggml_compute_forward_soft_max_f32        (ggml/src/ggml-cpu/ggml-cpu.c:10222)
// Used by tests/ to pin oracle/pkv_oracle.c and by tests/golden/make_golden.py to emit fixtures.
#include "ggml.h"
#include "ggml-cpu.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static struct ggml_context * ctx_new(size_t data_bytes) {
    struct ggml_init_params ip = {
        /*.mem_size   =*/ data_bytes + (size_t) 64 * 1024 * 1024,
        /*.mem_buffer =*/ NULL,
        /*.no_alloc   =*/ false,
    };
    return ggml_init(ip);
}

static int run1(struct ggml_context * ctx, struct ggml_tensor * node, int n_threads) {
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, node);
    return ggml_graph_compute_with_ctx(ctx, gf, n_threads) == GGML_STATUS_SUCCESS ? 0 : -1;
}

size_t ref_row_size(int type, int64_t ne) { return ggml_row_size((enum ggml_type) type, ne); }

// q   : f32  [n_head][n_q][D]            (ggml ne = {D, n_q, n_head})
// k,v : type [n_head_kv][n_kv][row(D)]   (ggml ne = {D, n_kv, n_head_kv}), reference block bytes
// mask: f16  [n_q_pad][n_kv] or NULL     (ggml ne = {n_kv, n_q_pad}), n_q_pad = GGML_PAD(n_q, GGML_KQ_MASK_PAD = 64)
// out : f32  [n_q][n_head][D]            (ggml ne = {D, n_head, n_q})
int ref_flash_attn_ext(const float * q, const void * k, const void * v, const uint16_t * mask, float * out,
                       int D, int n_q, int n_head, int n_kv, int n_head_kv, int n_q_pad,
                       int type_k, int type_v, float scale, float max_bias, float logit_softcap,
                       int n_threads) {
    const size_t kb = ggml_row_size((enum ggml_type) type_k, D) * (size_t) n_kv * n_head_kv;
    const size_t vb = ggml_row_size((enum ggml_type) type_v, D) * (size_t) n_kv * n_head_kv;
    const size_t qb = sizeof(float) * (size_t) D * n_q * n_head;
    struct ggml_context * ctx = ctx_new(kb + vb + 2 * qb + (size_t) 2 * n_kv * (n_q_pad + 1) + (size_t) n_threads * 8192);
    if (!ctx) return -2;
    struct ggml_tensor * tq = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, D, n_q, n_head);
    struct ggml_tensor * tk = ggml_new_tensor_3d(ctx, (enum ggml_type) type_k, D, n_kv, n_head_kv);
    struct ggml_tensor * tv = ggml_new_tensor_3d(ctx, (enum ggml_type) type_v, D, n_kv, n_head_kv);
    struct ggml_tensor * tm = NULL;
    memcpy(tq->data, q, qb);
    memcpy(tk->data, k, kb);
    memcpy(tv->data, v, vb);
    if (mask) {
        tm = ggml_new_tensor_2d(ctx, GGML_TYPE_F16, n_kv, n_q_pad);
        memcpy(tm->data, mask, (size_t) 2 * n_kv * n_q_pad);
    }
    struct ggml_tensor * r = ggml_flash_attn_ext(ctx, tq, tk, tv, tm, scale, max_bias, logit_softcap);
    ggml_flash_attn_ext_set_prec(r, GGML_PREC_F32);  // as src/llama-graph.cpp:1197
    int rc = run1(ctx, r, n_threads);
    if (rc == 0) memcpy(out, r->data, qb);
    ggml_free(ctx);
    return rc;
}

// x: f32 [n_tok][n_head][ne0] (ggml ne = {ne0, n_head, n_tok}); pos: i32 [n_tok]; ff: f32 [n_dims/2] or NULL
int ref_rope_ext(const float * x, const int32_t * pos, const float * ff, float * out,
                 int ne0, int n_head, int n_tok, int n_dims, int mode, int n_ctx_orig,
                 float freq_base, float freq_scale, float ext_factor, float attn_factor,
                 float beta_fast, float beta_slow, int n_threads) {
    const size_t xb = sizeof(float) * (size_t) ne0 * n_head * n_tok;
    struct ggml_context * ctx = ctx_new(2 * xb + 4096 * 16);
    if (!ctx) return -2;
    struct ggml_tensor * tx = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, ne0, n_head, n_tok);
    struct ggml_tensor * tp = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_tok);
    struct ggml_tensor * tf = NULL;
    memcpy(tx->data, x, xb);
    memcpy(tp->data, pos, sizeof(int32_t) * (size_t) n_tok);
    if (ff) {
        tf = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n_dims / 2);
        memcpy(tf->data, ff, sizeof(float) * (size_t) (n_dims / 2));
    }
    struct ggml_tensor * r = ggml_rope_ext(ctx, tx, tp, tf, n_dims, mode, n_ctx_orig, freq_base, freq_scale,
                                           ext_factor, attn_factor, beta_fast, beta_slow);
    int rc = run1(ctx, r, n_threads);
    if (rc == 0) memcpy(out, r->data, xb);
    ggml_free(ctx);
    return rc;
}

// ggml_cpy f32 -> type (the KV-store conversion, src/llama-graph.cpp:1379): x f32 [n_rows][ne0] -> dst bytes
int ref_cpy_f32(const float * x, void * dst, int ne0, int n_rows, int type_dst, int n_threads) {
    const size_t xb = sizeof(float) * (size_t) ne0 * n_rows;
    const size_t db = ggml_row_size((enum ggml_type) type_dst, ne0) * (size_t) n_rows;
    struct ggml_context * ctx = ctx_new(xb + db);
    if (!ctx) return -2;
    struct ggml_tensor * tx = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, ne0, n_rows);
    struct ggml_tensor * td = ggml_new_tensor_2d(ctx, (enum ggml_type) type_dst, ne0, n_rows);
    memcpy(tx->data, x, xb);
    struct ggml_tensor * r = ggml_cpy(ctx, tx, td);
    int rc = run1(ctx, r, n_threads);
    if (rc == 0) memcpy(dst, td->data, db);
    ggml_free(ctx);
    return rc;
}

// type -> f32 through the reference's to_float trait (ggml/src/ggml-quants.c:349 for q8_0)
int ref_to_float(const void * src, float * out, int64_t n, int type_src) {
    const struct ggml_type_traits * tt = ggml_get_type_traits((enum ggml_type) type_src);
    if (!tt || !tt->to_float) return -1;
    tt->to_float(src, out, n);
    return 0;
}

// deterministic "_ref" quantizers (ggml/src/ggml-quants.c:194 q8_0, :35 q4_0) via from_float_ref
int ref_from_float_ref(const float * x, void * dst, int64_t n, int type_dst) {
    const struct ggml_type_traits * tt = ggml_get_type_traits((enum ggml_type) type_dst);
    if (!tt || !tt->from_float_ref) return -1;
    tt->from_float_ref(x, dst, n);
    return 0;
}

// soft_max_ext: x f32 [n_rows][n_cols], mask f32 [n_rows_pad][n_cols] or NULL
int ref_soft_max_ext(const float * x, const float * mask, float * out, int n_cols, int n_rows, int n_rows_mask,
                     float scale, float max_bias, int n_threads) {
    const size_t xb = sizeof(float) * (size_t) n_cols * n_rows;
    struct ggml_context * ctx = ctx_new(3 * xb + sizeof(float) * (size_t) n_cols * (n_rows_mask + 1));
    if (!ctx) return -2;
    struct ggml_tensor * tx = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, n_cols, n_rows);
    struct ggml_tensor * tm = NULL;
    memcpy(tx->data, x, xb);
    if (mask) {
        tm = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, n_cols, n_rows_mask);
        memcpy(tm->data, mask, sizeof(float) * (size_t) n_cols * n_rows_mask);
    }
    struct ggml_tensor * r = ggml_soft_max_ext(ctx, tx, tm, scale, max_bias);
    int rc = run1(ctx, r, n_threads);
    if (rc == 0) memcpy(out, r->data, xb);
    ggml_free(ctx);
    return rc;
}
