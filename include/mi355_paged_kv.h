/* mi355_paged_kv.h — C ABI of the MI355X-native paged-KV-cache attention path for llama.cpp.
 *
 * Drop-in boundary for ONE hot path of the reference (KwanyoungCho/llama.cpp_test, "/ref" below =
 * /root/reference): KV store (+RoPE), block-table gather, paged single-token decode attention,
 * flash-style prefill attention, in-place K-shift, and the host block manager behind the
 * llama_kv_cache_* method set.  Everything here is `extern "C"`, plain pointers and sizes; device
 * pointers are raw HIP device addresses, `stream` is a hipStream_t passed as void*.
 * The library (libmi355_paged_kv.so) links only libamdhip64; there is NO CPU fallback: every compute
 * entry point returns PKV_ERR_NO_DEVICE when no gfx950 device/code object is available.
 *
 * Reference interfaces each group replaces are cited as /ref/<file>:<line>.
 *
 * ---------------------------------------------------------------------------------------------
 * Pool layout (private to this library; reference bytes only cross the boundary via
 * pkv_pool_scatter_rows / pkv_pool_gather_rows):
 *
 *   One pool per layer and per K/V, replacing the flat ring `k_l[il]` / `v_l[il]`
 *   (/ref/src/llama-kv-cache.cpp:100-101).  pool = [n_blocks][n_head_kv] tiles; tile(b,h) holds
 *   block_size (128) consecutive logical tokens of ONE kv head, so the per-head tile the decode kernel
 *   streams is contiguous (the reference's per-head view is 256 B every 2048 B,
 *   /ref/src/llama-graph.cpp:1414-1419):
 *     K pools (a lane of the decode kernel reads 16 contiguous bytes of ONE token row):
 *       f16  : half  e[tok][D]                                                              (32 KiB at 128x128)
 *       q8_0 : int8  qs[tok][D], then half d[tok][D/32]                                     (17 KiB) -- block_q8_0
 *              (/ref/ggml/src/ggml-common.h:203-208: 34 B, 2-B aligned) split so every 16-B load is aligned
 *       q4_0 : uint8 qs[tok][D/2] (per 32-block byte j = elem j | elem j+16 << 4, as block_q4_0,
 *              /ref/ggml/src/ggml-common.h:161-166), then half d[tok][D/32]                 ( 9 KiB)
 *     V pools (token-interleaved so that P.V needs 4 accumulators per lane and, for f16, is a pure
 *     v_dot2c_f32_f16 stream): uint32 w[tok/TL][D], the dword of one d holds its TL consecutive tokens,
 *       f16  : TL 2 (half2 = tokens t, t+1)        q8_0 : TL 4 (4 x int8)        q4_0 : TL 8 (8 nibbles)
 *       quantized types append half d[tok/TL][D/32][TL]
 *   tile offset = (b * n_head_kv + h) * pkv_tile_bytes(type, block_size, D); K and V tiles of a type have
 *   the same size; functions that touch tile contents take `is_v` to select the layout.
 *   Bytes per token are identical to the reference's row sizes (2048 / 1088 / 576 B per 1024 elems).
 * ---------------------------------------------------------------------------------------------
 */
#ifndef MI355_PAGED_KV_H
#define MI355_PAGED_KV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PKV_ABI_VERSION 3   /* 2 (r03): PKV_ATTN_Q_F16 / PKV_ATTN_OUT_F16, pkv_attn_decode_tiles; 3 (r03): pkv_flash_attn_ext_flat, pkv_attn_step_layers, PKV_TYPE_BF16 / _Q4_1 / _Q5_0 / _Q5_1 / _IQ4_NL */

/* values follow enum ggml_type (/ref/ggml/include/ggml.h:351-360) */
enum pkv_type { PKV_TYPE_F32 = 0, PKV_TYPE_F16 = 1, PKV_TYPE_Q4_0 = 2, PKV_TYPE_Q8_0 = 8,
                PKV_TYPE_Q4_K = 12, /* weights of pkv_mul_mat / pkv_gemv_q only (block_q4_K, /ref/ggml/src/ggml-common.h:279-290); not a KV type */
                PKV_TYPE_Q5_K = 13, /* weights of pkv_mul_mat / pkv_gemv_q / pkv_get_rows (block_q5_K, :291-308; Q5_K_M files); and (r03) a plane layout like the other two                  */
                PKV_TYPE_Q6_K = 14, /* weights of pkv_gemv_q / pkv_get_rows only (block_q6_K, :309-319): what a Q4_K_M file mixes in          */
                PKV_TYPE_BF16 = 30, /* K / V rows of pkv_flash_attn_ext_flat only (ggml_bf16_t, /ref/ggml/include/ggml.h:339-342): not a pool type */
                /* the remaining cache types of /ref/common/arg.cpp:172-182 — flat rows only: pkv_convert_f32 (the KV store) and pkv_flash_attn_ext_flat */
                PKV_TYPE_Q4_1 = 3, PKV_TYPE_Q5_0 = 6, PKV_TYPE_Q5_1 = 7, PKV_TYPE_IQ4_NL = 20 };

enum pkv_status {
    PKV_OK              =  0,
    PKV_ERR_INVALID     = -1, /* bad argument (shape/type/alignment) — nothing was launched        */
    PKV_ERR_NO_DEVICE   = -2, /* no HIP device / kernel image for this GPU — there is no CPU path   */
    PKV_ERR_HIP         = -3, /* a HIP call failed; pkv_last_error() has the text                   */
    PKV_ERR_NO_SPACE    = -4, /* block pool exhausted (find_slot failure, /ref/src/llama-kv-cache.cpp:654) */
    PKV_ERR_UNSUPPORTED = -5  /* valid in the reference but outside this path (e.g. ALiBi max_bias) */
};

int          pkv_abi_version(void);
const char * pkv_last_error(void);              /* thread-local text of the last non-OK status     */
int          pkv_device_count(void);            /* number of visible gfx950 devices (0 on CPU box) */
const char * pkv_device_arch(int device);       /* e.g. "gfx950"; NULL if no such device            */

size_t pkv_row_size  (int type, int64_t ne);                    /* = ggml_row_size, /ref/ggml/include/ggml.h:651 */
size_t pkv_tile_bytes(int type, int block_size, int head_dim);   /* see "Pool layout"               */
size_t pkv_pool_bytes(int type, int n_blocks, int n_head_kv, int block_size, int head_dim);

/* RoPE parameters = op_params of GGML_OP_ROPE (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10681-10693) */
typedef struct pkv_rope_params {
    int32_t n_dims;        /* op_params[1]                                   */
    int32_t mode;          /* op_params[2]; 0 = NORM (llama), 2 = NEOX        */
    int32_t n_ctx_orig;    /* op_params[4]                                   */
    float   freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow; /* [5..10]        */
    const float * freq_factors; /* device f32 [n_dims/2] or NULL (src[2], llama-3 rope_freqs)       */
} pkv_rope_params;

/* ---- kernels (stateless) ------------------------------------------------------------------- */

/* RoPE on activations: replaces ggml_rope_ext on Q/K in llm_build_llama (/ref/src/llama-model.cpp:4091-4102),
 * CPU kernel ggml_compute_forward_rope_f32 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10669).
 * x,out: device f32 [n_tok][n_head][head_dim] (out may alias x); pos: device i32 [n_tok]. */
int pkv_rope_f32(const float * x, const int32_t * pos, float * out, int n_tok, int n_head, int head_dim,
                 const pkv_rope_params * rope, int inverse, void * stream);

/* same with explicit input strides (in floats) for non-contiguous views of x (ggml views/permutes); out stays contiguous */
int pkv_rope_f32_strided(const float * x, int64_t x_stride_tok, int64_t x_stride_head, const int32_t * pos, float * out,
                         int n_tok, int n_head, int head_dim, const pkv_rope_params * rope, int inverse, void * stream);

/* RoPE on an f16 tensor (strides in halfs): f16 in, f32 math, f16 out = ggml_compute_forward_rope_f16
 * (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10855), the op build_rope_shift applies to an f16 K cache (/ref/src/llama-context.cpp:483-492) */
int pkv_rope_f16_strided(const void * x, int64_t x_stride_tok, int64_t x_stride_head, const int32_t * pos, void * out,
                         int n_tok, int n_head, int head_dim, const pkv_rope_params * rope, int inverse, void * stream);

/* GGML_OP_CPY f32 -> f16/q8_0/q4_0 into a CONTIGUOUS typed tensor in the reference's own byte format
 * (ggml_compute_forward_dup_f32, /ref/ggml/src/ggml-cpu/ggml-cpu.c:3742): x f32 [n_rows][ne0], ne0 % 64 == 0.
 * Also q4_1 / q5_0 / q5_1 / iq4_nl (ne0 % 32 == 0; the *_ref quantizers the CPU backend's from_float forwards to,
 * /ref/ggml/src/ggml-quants.c:68-190, :4573-4720): bit-exact, all-zero iq4_nl blocks excepted (the reference leaves their indices undefined). */
int pkv_convert_f32(const float * x, void * dst, int type, int64_t n_rows, int64_t ne0, void * stream);
/* the same store for n_cells KV rows ([n_cells][n_head_kv][head_dim] f32) that ALSO lands in a paged pool with identity placement
 * (cell c = block c / block_size, offset c % block_size; K or V layout by is_v): dst gets the reference's bytes for cells
 * [cell0, cell0 + n_cells) — dst points at the first of them — and the pool gets the same quantized bits, in one launch.  What the
 * ggml-backend plugin runs for the KV-store CPY of a step once a cache tensor has its paged mirror. */
/* (dst / k_dst / v_dst of the three *_paged entry points may be NULL: the paged rows are then the only copy — a cache tensor that IS its pool) */
int pkv_convert_f32_paged(const float * x, void * dst, int type, void * pool, int is_v, int64_t cell0, int64_t n_cells, int n_head_kv, int head_dim,
                          int block_size, void * stream);

/* ... and with the RoPE of the K rows in front: x = un-rotated k_cur, strided [tok][head][head_dim] (strides in elements, 0 = packed),
 * roped (optional, packed f32) = what GGML_OP_ROPE writes, dst / pool = what the CPY of it stores (K layout).  ROPE -> CPY of
 * build_attn (/ref/src/llama-graph.cpp:1375-1397) as one launch. */
int pkv_rope_convert_f32_paged(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, const pkv_rope_params * rope, float * roped, void * dst,
                               int type, void * pool, int64_t cell0, int64_t n_tok, int n_head_kv, int head_dim, int block_size, void * stream);

/* ... and all three per-token ops that sit between the q / k / v products and the attention of a llama layer as ONE launch: RoPE(Q)
 * out of place (q_out packed [tok][n_head][head_dim]), RoPE(K) + store, store(V) — K and V as in pkv_rope_convert_f32_paged /
 * pkv_convert_f32_paged (dst = reference bytes of cells cell0.., pool = paged rows, identity placement).  Strides in elements,
 * 0 = packed; v is packed [tok][n_head_kv][head_dim].  Q and K share pos and the rope parameters (build_attn's two ggml_rope_ext). */
typedef struct pkv_qkv_store_args {
    const float * q; int64_t q_stride_tok, q_stride_head; float * q_out;
    const float * k; int64_t k_stride_tok, k_stride_head; void * k_dst; void * k_pool; int64_t k_cell0;
    const float * v; void * v_dst; void * v_pool; int64_t v_cell0;
    const int32_t * pos; const pkv_rope_params * rope;
    int32_t type_k, type_v, n_tok, n_head, n_head_kv, head_dim, block_size;
} pkv_qkv_store_args;
int pkv_qkv_store_f32_paged(const pkv_qkv_store_args * a, void * stream);

/* KV store: replaces ggml_cpy(k_cur -> k_cache_view) / (v_cur -> v_cache_view) in build_attn
 * (/ref/src/llama-graph.cpp:1367-1398), CPU kernel ggml_compute_forward_dup_f32 + from_float
 * (/ref/ggml/src/ggml-cpu/ggml-cpu.c:3742, ggml-cpu-quants.c:738).
 * k_cur,v_cur: device f32 [n_tok][n_head_kv][head_dim].  Token t goes to block slot_block[t], offset
 * slot_off[t] (slot_block[t] < 0 = skip).  If `rope` != NULL, K is rotated with pos[t] first (fused
 * ggml_rope_ext + ggml_cpy).  Either of k_cur/v_cur may be NULL to store only the other. */
int pkv_kv_store(const float * k_cur, const float * v_cur, void * k_pool, void * v_pool,
                 int type_k, int type_v, const int32_t * slot_block, const int32_t * slot_off,
                 const int32_t * pos, const pkv_rope_params * rope,
                 int n_tok, int n_head_kv, int head_dim, int block_size, void * stream);

/* The three per-token ops that precede attention in a decoder layer, in ONE launch: RoPE(Q) in place
 * (ggml_rope_ext, /ref/src/llama-model.cpp:4091-4096) + RoPE(K) + KV store (= pkv_kv_store with `rope`).
 * q: device f32 [n_tok][n_head][head_dim], rotated in place with pos[t]. */
int pkv_qkv_prepare(float * q, int n_head, const float * k_cur, const float * v_cur, void * k_pool, void * v_pool,
                    int type_k, int type_v, const int32_t * slot_block, const int32_t * slot_off,
                    const int32_t * pos, const pkv_rope_params * rope,
                    int n_tok, int n_head_kv, int head_dim, int block_size, void * stream);

/* Paged attention, one launch for all sequences of a micro-batch: replaces the K/V "gather" views +
 * ggml_flash_attn_ext of build_attn / build_attn_mha (/ref/src/llama-graph.cpp:1404-1432, :1160-1250),
 * CPU kernel ggml_compute_forward_flash_attn_ext_f16 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:12219), and the dense
 * KQ_mask built on the host per step (/ref/src/llama-graph.cpp:403-468), which becomes (seq_lens, block_table). */
typedef struct pkv_attn_args {
    const float *   q;            /* device f32; token t, head h at q + t*q_stride_tok + h*q_stride_head      */
    int64_t         q_stride_tok, q_stride_head; /* in floats; contiguous [n_tok][n_head][D] = (n_head*D, D)    */
    float *         out;          /* device f32 [n_tok][n_head][head_dim]  (ggml dst [D, n_head, n_tok])       */
    const void *    k_pool;       /* this layer's pools, "Pool layout"                                         */
    const void *    v_pool;
    const int32_t * block_table;  /* device i32 [n_seq][max_blocks]: logical block -> physical block          */
    const int32_t * seq_lens;     /* device i32 [n_seq]: logical KV length INCLUDING this step's tokens        */
    const int32_t * q_start;      /* device i32 [n_seq+1] token offsets per sequence; NULL = one token per seq  */
    const int32_t * slot_pos;     /* device i32 [n_seq][max_blocks*block_size] cell positions for irregular     */
                                  /* sequences (holes after seq_rm / seq_add), -1 = empty; NULL = slot i has pos i */
    const int32_t * q_pos;        /* device i32 [n_tok] query positions (only read when slot_pos != NULL)       */
    void *          workspace;    /* device scratch, >= pkv_attn_workspace_bytes(...)                           */
    size_t          workspace_bytes;
    int32_t n_seq, n_tok, max_q_per_seq, n_head, n_head_kv, head_dim, block_size, max_blocks;
    int32_t type_k, type_v;
    float   scale;                /* op_params[0], 1/sqrt(head_dim) for llama (/ref/src/llama-model.cpp:4054)   */
    float   logit_softcap;        /* op_params[2]; 0 = off.  max_bias (ALiBi) != 0 is PKV_ERR_UNSUPPORTED       */
    int32_t max_seq_len;          /* upper bound of seq_lens[] known on the host (sizes the split-KV grid)      */
    int32_t flags;                /* PKV_ATTN_* bits                                                            */
    const void * mask;            /* optional additive mask (pkv_attn_decode: -inf / finite per cell, no ALiBi), device f16 [n_tok][mask_stride] (row = */
    int64_t mask_stride;          /* token, column = logical KV cell): the general KQ_mask of ggml_flash_attn_ext            */
    float   max_bias;             /* (/ref/ggml/include/ggml.h:1794-1807); max_bias > 0 = ALiBi slopes, needs a mask.      */
                                  /* NULL = the causal-by-position structure of llama (no mask tensor at all)              */
    void *  ev_start;             /* optional hipEvent_t pair bound to the split-KV decode launch itself          */
    void *  ev_stop;              /* (hipExtLaunchKernel): elapsed = the dispatch's own GPU duration, as rocprofv3 */
                                  /* --kernel-trace reports it.  NULL = plain launch.                              */
    void *  out_image;            /* pkv_attn_decode / _step, optional (r03): ALSO leave every out row as the q8_K activation image  */
                                  /* pkv_gemv_q takes as `xq` ([n_tok][pkv_act_image_bytes(n_head*head_dim)], the bytes pkv_quantize_act */
                                  /* makes of the row): the merge launch quantizes what it has just normalised, and the wo product of */
                                  /* the layer starts from 5 KB of integers instead of reading, scaling and rounding 16 KB of floats  */
                                  /* in every workgroup.  Needs 256 % head_dim == 0 and no PKV_ATTN_NO_COMBINE / _FUSED_COMBINE.      */
} pkv_attn_args;

/* leave the per-block (m, l, O) partials in `workspace` and skip the merge launch (measurement / callers that
 * merge elsewhere); `out` is not written */
#define PKV_ATTN_NO_COMBINE 1
/* merge the per-block partials INSIDE the decode launch (the last-arriving workgroup of each (sequence, kv head) does
 * it; arrival tickets live at the end of `workspace`).  Contract: `workspace` was zeroed once by
 * pkv_attn_workspace_init() and is used by one launch at a time.  Ignored (two-launch path) with softcap / slot_pos. */
#define PKV_ATTN_FUSED_COMBINE 2
/* pkv_attn_prefill: every query sees all seq_lens[s] cells (no causal structure); masking, if any, comes from `mask` */
#define PKV_ATTN_NONCAUSAL 4
/* pkv_attn_prefill / pkv_attn_prefill_step: q (out) is f16, strides in elements — half the boundary traffic of a prompt ubatch (32 x 512 tokens of an
 * 8B layer move 268 MB of f32 Q in and as much O out for 67 MB of KV).  For f16 K the reference rounds Q to f16 anyway (GGML_FP32_TO_FP16 in its
 * q_to_vec_dot, /ref/ggml/src/ggml-cpu/ggml-cpu.c:12343): an f16 Q changes nothing; an f16 O is the f32 result rounded once (relative 2^-11).  The
 * ggml-backend plugin keeps the f32 contract of ggml_flash_attn_ext; the own runner uses these. */
#define PKV_ATTN_Q_F16 8
#define PKV_ATTN_OUT_F16 16

size_t pkv_attn_workspace_bytes(int n_tok, int n_head, int head_dim, int max_seq_len, int block_size);
/* what pkv_attn_prefill alone can use (its split-KV mode for small launches over a long past; 0 above 2048 tokens).  A
 * smaller or NULL workspace is fine: the call then runs unsplit. */
size_t pkv_attn_prefill_workspace_bytes(int n_tok, int n_head, int head_dim);
/* zero a fresh workspace (asynchronously, on `stream`); required once before PKV_ATTN_FUSED_COMBINE is used with it */
int    pkv_attn_workspace_init(void * workspace, size_t bytes, void * stream);
/* single-token decode (one query token per sequence, q_start == NULL): split-KV over 128-token blocks.  Launches of at least four
 * rounds of resident workgroups take the multi-tile form: one workgroup walks several consecutive 128-token tiles of its (sequence,
 * kv head) with an online softmax and writes ONE partial; when a workgroup covers the whole sequence it writes `out`
 * itself (no partials, no merge launch — also under PKV_ATTN_NO_COMBINE, which then has nothing to leave behind). */
int pkv_attn_decode(const pkv_attn_args * args, void * stream);
/* the 128-token tiles one workgroup of pkv_attn_decode / pkv_attn_decode_step walks for a launch of this shape (1 = one
 * tile per workgroup + merge launch; >= ceil(max_seq_len / 128) = the form without partials).  Informational (bench.py). */
int pkv_attn_decode_tiles(int max_seq_len, int n_head, int n_head_kv, int n_seq);
/* One decode step of a layer in ONE launch (+ the merge): pkv_qkv_prepare + pkv_attn_decode fused, i.e. the whole
 * ggml_rope_ext(Qcur) / ggml_rope_ext(Kcur) / ggml_cpy(Kcur, Vcur -> cache) / ggml_flash_attn_ext run of
 * llm_build_llama + build_attn (/ref/src/llama-model.cpp:4091-4105, /ref/src/llama-graph.cpp:1367-1432) for a
 * one-token-per-sequence micro-batch.  The new token of sequence s is cell seq_lens[s]-1 (its block must already be
 * in block_table: find_slot ran).  Differences from the two calls: args->q is read un-rotated and NOT modified (the
 * rotated Q goes straight into the QK^T operand), and the results are bit-identical to pkv_qkv_prepare +
 * pkv_attn_decode, pools included.  PKV_ATTN_NO_COMBINE / PKV_ATTN_FUSED_COMBINE act as in pkv_attn_decode.
 * PKV_ERR_UNSUPPORTED with softcap or slot_pos (use the two calls). */
typedef struct pkv_step_args {
    const float *   k_cur;            /* device f32 [n_seq][n_head_kv][head_dim], this step's K rows, un-rotated  */
    const float *   v_cur;            /* device f32 [n_seq][n_head_kv][head_dim]                                  */
    const int32_t * pos;              /* device i32 [n_seq]: position of the new token (the RoPE angle); may be   */
                                      /* NULL when rope_cache is given                                            */
    const pkv_rope_params * rope;     /* NULL = no RoPE (Q and K are used as given)                               */
    const float *   rope_cache;       /* optional pkv_rope_cache() rows of these tokens, device f32               */
                                      /* [n_seq][head_dim/2][2]: computed ONCE per step and shared by all layers; */
                                      /* NULL = the call builds them itself from (pos, rope) with one more launch */
} pkv_step_args;
/* ggml_rope_cache_init (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10595-10609) for a micro-batch: cache[t][i] = (cos, sin) * mscale
 * of rotation pair i at pos[t] (identity past n_dims/2).  cache: device f32 [n_tok][head_dim/2][2], 16-byte aligned. */
int pkv_rope_cache(const int32_t * pos, int n_tok, int head_dim, const pkv_rope_params * rope, float * cache, void * stream);
int pkv_attn_decode_step(const pkv_attn_args * args, const pkv_step_args * step, void * stream);
/* The prefill analogue (a ubatch with several query tokens per sequence): RoPE(K) + KV store of the ubatch's rows into (slot_block, slot_off) — the
 * cells find_slot assigned, as in pkv_kv_store — then pkv_attn_prefill with RoPE(Q) fused into its Q fetch: args->q is read un-rotated and NOT
 * modified, the rotated Q goes straight into the QK^T operand.  Replaces pkv_qkv_prepare + pkv_attn_prefill (/ref/src/llama-model.cpp:4091-4105,
 * /ref/src/llama-graph.cpp:1367-1432 for n_tokens > 1) with the same pools and the same output bits, without the in-place rotation pass over Q
 * (2/3 of that call's traffic).  step->rope_cache (pkv_rope_cache rows of the ubatch's tokens, computed once per ubatch and shared by all layers)
 * is REQUIRED with rope; step->pos is still read by the K rotation.  PKV_ERR_UNSUPPORTED for NEOX pairs (use the two calls). */
int pkv_attn_prefill_step(const pkv_attn_args * args, const pkv_step_args * step, const int32_t * slot_block, const int32_t * slot_off, void * stream);
/* One ubatch through ALL layers in one call (r03): the per-layer loop of llm_build_llama's attention block (/ref/src/llama-model.cpp:4066-4110) as the host
 * side runs it — pkv_attn_decode_step (slot_block == NULL: one token per sequence) or pkv_attn_prefill_step (slot_block / slot_off given) for layer
 * l = 0 .. n_layer - 1 with args / step as given except the per-layer pointers: q[l] / k_cur[l] / v_cur[l] / out[l] (device tensors of that layer) and
 * k_pool[l] / v_pool[l].  Same launches, same results; one trip through the binding instead of n_layer (a Python caller spends ~18 us per call: at 32
 * layers that is the whole 0.6 ms of a 32-sequence decode step).  Stops at the first error. */
int pkv_attn_step_layers(const pkv_attn_args * args, const pkv_step_args * step, int n_layer, const void * const * q, const void * const * k_cur,
                         const void * const * v_cur, void * const * out, const void * const * k_pool, const void * const * v_pool,
                         const int32_t * slot_block, const int32_t * slot_off, void * stream);
/* DIAGNOSTIC build of the decode-step kernel with in-kernel phase stamps (f16 KV, head_dim 128, gqa 4 / 8; the two-launch step):
 * same arguments as pkv_attn_decode_step; `stamps` = device u64 [n_workgroups][8 waves][PKV_DECODE_NSTAMP], workgroup index
 * (seq * n_head_kv + kv_head) * n_splits + split.  Slots: 0 = s_memrealtime (100 MHz) at entry, 1..10 = s_memtime (shader clock) at
 * entry / scalars loaded / loads issued / Q operand ready / K landed / logits + softmax done / V landed / PV done /
 * barrier passed / partial stores issued, 11 = s_memrealtime after the stores were acknowledged.  mode 1 = with a forced
 * `s_waitcnt vmcnt` in front of the "K landed" / "V landed" stamps (clean phase shares); mode 2 = stamps only, the waits stay where the
 * compiler put them (how far the product build overlaps the KV stream with QK^T / PV).  Its run time is not a
 * measurement (the stamps fence overlaps the product kernel has); read shares.  Never used by the product path. */
#define PKV_DECODE_NSTAMP 12
int pkv_attn_decode_stamps(const pkv_attn_args * args, const pkv_step_args * step, uint64_t * stamps, int mode, void * stream);
/* measurement aid: raw hipEvent_t handles for pkv_attn_args.ev_start / ev_stop */
void * pkv_event_create(void);
void   pkv_event_destroy(void * ev);
int    pkv_event_synchronize(void * ev);
int    pkv_event_elapsed_ms(void * ev_start, void * ev_stop, float * ms);
/* batched prefill (many query tokens per sequence, causal): MFMA QK^T / PV */
int pkv_attn_prefill(const pkv_attn_args * args, void * stream);

/* GGML_OP_FLASH_ATTN_EXT on the reference's OWN tensor layout, for what the paged kernels above do not take (r03): head sizes other than 64 / 128 (even,
 * <= 256; a multiple of 32 with quantized rows), bf16 K / V, any gqa ratio, K / V as strided views of flat tensors — the rest of the domain of
 * ggml_compute_forward_flash_attn_ext_f16 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:12219-12432) that /ref/tests/test-backend-ops.cpp:4367-4390 tests (head sizes
 * 64 / 80 / 128 / 256 x f16 / bf16 / q8_0 / q4_0) and `-ctk / -ctv bf16` runs (/ref/common/arg.cpp:172-182).  No pool: K / V rows are read where they lie, in
 * the reference's bytes (block_q8_0 / block_q4_0, 16-bit elements), Q rounded through the K type's vec_dot_type as the reference does (f16 / bf16 /
 * q8_0 with the AVX2 rounding).  One workgroup per (query row, head): a coverage kernel, not a roofline one — callers ask the paged kernels first (the
 * ggml-backend plugin does).
 *   q    : device f32, row (t, h) at q + t*q_stride_tok + h*q_stride_head (floats)       [ggml q: ne = {D, n_q, n_head}]
 *   k, v : device bytes, row (cell c, kv head g) at k + c*k_stride_tok + g*k_stride_head (BYTES = ggml nb[1], nb[2]), types PKV_TYPE_F16 / _BF16 / _Q8_0 / _Q4_0 / _Q4_1 / _Q5_0 / _Q5_1 / _IQ4_NL
 *   mask : device f16 [n_q][mask_stride] or NULL (no causal structure is implied: what is not masked is seen); max_bias > 0 = ALiBi slopes on the mask
 *   out  : device f32 [n_q][n_head][head_dim]                                             [ggml dst: ne = {D, n_head, n_q}]                                  */
typedef struct pkv_flat_attn_args {
    const float * q;    int64_t q_stride_tok, q_stride_head;
    const void *  k;    int64_t k_stride_tok, k_stride_head;
    const void *  v;    int64_t v_stride_tok, v_stride_head;
    const void *  mask; int64_t mask_stride;
    float *       out;
    int32_t n_q, n_kv, n_head, n_head_kv, head_dim, type_k, type_v;
    float   scale, max_bias, logit_softcap;           /* op_params[0..2] */
    void *  workspace;                                /* optional device scratch (pkv_flash_attn_ext_flat_workspace_bytes): a launch of few query rows over a */
    size_t  workspace_bytes;                          /* long K / V (a decode step) is then cut along the cells (split-KV + a merge launch); NULL = unsplit    */
} pkv_flat_attn_args;
size_t pkv_flash_attn_ext_flat_workspace_bytes(int n_q, int n_head, int head_dim, int n_kv);   /* 0: a launch of this shape never splits */
int pkv_flash_attn_ext_flat(const pkv_flat_attn_args * args, void * stream);

/* In-place K-shift: replaces build_kv_self_shift / build_rope_shift (/ref/src/llama-context.cpp:447-571).
 * For each listed cell rotate the cached K row of every kv head by `delta` positions (f16 in place; quantized:
 * dequantize -> rope -> requantize, as :469-487).  cell i = (cell_block[i], cell_off[i]), shift cell_delta[i]. */
int pkv_k_shift(void * k_pool, int type_k, const int32_t * cell_block, const int32_t * cell_off,
                const int32_t * cell_delta, int n_cells, int n_head_kv, int head_dim, int block_size,
                const pkv_rope_params * rope, void * stream);

/* Block copy (copy-on-write / defrag moves): replaces the ggml_cpy row-range moves of build_kv_self_defrag
 * (/ref/src/llama-context.cpp:653-711).  Copies whole blocks src[i] -> dst[i] inside one pool. */
int pkv_copy_blocks(void * pool, int type, const int32_t * src, const int32_t * dst, int n,
                    int n_head_kv, int head_dim, int block_size, void * stream);

/* Reference-format row I/O: what ggml_backend_tensor_set/get on k_l[il]/v_l[il] do for session files
 * (/ref/src/llama-kv-cache.cpp:904-1001 write, :1105-1221 read) and for test-backend-ops
 * (/ref/tests/test-backend-ops.cpp:471).  rows: device bytes [n_tok][n_head_kv][pkv_row_size(type, head_dim)],
 * i.e. one KV-cache row per token exactly as the reference lays it out. */
int pkv_pool_scatter_rows(void * pool, int type, int is_v, const void * rows, const int32_t * slot_block,
                          const int32_t * slot_off, int n_tok, int n_head_kv, int head_dim, int block_size,
                          void * stream);
/* scatter with explicit byte strides between tokens / heads of `rows` (any ggml view of a K/V tensor);
 * slot_block == NULL = identity placement (token t -> block t / block_size, offset t % block_size) */
int pkv_pool_scatter_rows_strided(void * pool, int type, int is_v, const void * rows, int64_t row_stride_tok,
                                  int64_t row_stride_head, const int32_t * slot_block, const int32_t * slot_off,
                                  int n_tok, int n_head_kv, int head_dim, int block_size, void * stream);
int pkv_pool_gather_rows(const void * pool, int type, int is_v, void * rows, const int32_t * slot_block,
                         const int32_t * slot_off, int n_tok, int n_head_kv, int head_dim, int block_size,
                         void * stream);

/* ---- host block manager: the llama_kv_cache method set -------------------------------------- */
/* Opaque paged cache = llama_kv_cache_unified re-designed around a block table
 * (/ref/src/llama-kv-cache.h:63-178).  Owns the per-layer device pools unless created with n_layer == 0
 * (host-only bookkeeping, usable without a GPU). */
typedef struct pkv_cache pkv_cache;

typedef struct pkv_cache_params {
    int32_t n_layer;         /* 0 = bookkeeping only, no device memory                             */
    int32_t n_head_kv, head_dim;
    int32_t type_k, type_v;  /* cparams type_k/type_v (/ref/include/llama.h:313-346)                */
    int32_t block_size;      /* tokens per block; 128                                               */
    int32_t n_blocks;        /* pool capacity; kv_size = n_blocks * block_size cells                */
    int32_t n_seq_max;       /* llama_context_params.n_seq_max                                      */
    int32_t device;          /* HIP device ordinal                                                  */
    int32_t evict_when_full; /* 0: allocation failure is reported (default); 1: the fork prototype's
                                BlockManager policy — evict the lowest allocated block               */
} pkv_cache_params;

pkv_cache * pkv_cache_create(const pkv_cache_params * p);
void        pkv_cache_free  (pkv_cache * c);

/* llama_kv_cache_* C wrappers (/ref/src/llama-kv-cache.cpp:1227-1332): same names, argument meaning
 * and NULL-tolerance. p0 < 0 means 0, p1 < 0 means +inf (/ref/src/llama-kv-cache.cpp:176-182). */
int32_t pkv_cache_n_tokens   (const pkv_cache * c);
int32_t pkv_cache_used_cells (const pkv_cache * c);
void    pkv_cache_clear      (pkv_cache * c);
int     pkv_cache_seq_rm     (pkv_cache * c, int32_t seq_id, int32_t p0, int32_t p1);     /* bool */
void    pkv_cache_seq_cp     (pkv_cache * c, int32_t seq_src, int32_t seq_dst, int32_t p0, int32_t p1);
void    pkv_cache_seq_keep   (pkv_cache * c, int32_t seq_id);
void    pkv_cache_seq_add    (pkv_cache * c, int32_t seq_id, int32_t p0, int32_t p1, int32_t delta);
void    pkv_cache_seq_div    (pkv_cache * c, int32_t seq_id, int32_t p0, int32_t p1, int32_t d);
int32_t pkv_cache_seq_pos_max(pkv_cache * c, int32_t seq_id);
void    pkv_cache_defrag     (pkv_cache * c);
int     pkv_cache_can_shift  (const pkv_cache * c);                                       /* bool */

/* find_slot (/ref/src/llama-kv-cache.cpp:453-672, non-recurrent branch): place the n_tok tokens of a
 * ubatch (token t belongs to sequence seq_id[t] at position pos[t]) and return their cells.
 * out_block/out_off (host, n_tok each) receive the physical block and in-block offset of every token.
 * Returns PKV_OK or PKV_ERR_NO_SPACE (nothing is modified on failure, as the reference's find_slot). */
int pkv_cache_find_slot(pkv_cache * c, int n_tok, const int32_t * seq_id, const int32_t * pos,
                        int32_t * out_block, int32_t * out_off);

/* block-table introspection (host) */
int32_t pkv_cache_seq_len      (const pkv_cache * c, int32_t seq_id);              /* logical cells   */
int32_t pkv_cache_seq_n_blocks (const pkv_cache * c, int32_t seq_id);
int32_t pkv_cache_seq_blocks   (const pkv_cache * c, int32_t seq_id, int32_t * out, int32_t cap);
int32_t pkv_cache_seq_is_regular(const pkv_cache * c, int32_t seq_id);            /* pos[i]==i ∀i     */
/* the per-ubatch device tables in one call (what set_inputs builds per ubatch in the reference, /ref/src/llama-graph.cpp:403-468, instead of the dense
 * mask): table[n_seq][width] = the sequences' block lists padded with 0, lens[n_seq] = their logical lengths, *all_regular = every sequence has
 * pos[i] == i.  Returns the longest block list (>= 0) or PKV_ERR_INVALID (a list longer than `width`). */
int32_t pkv_cache_block_tables(const pkv_cache * c, int32_t n_seq, const int32_t * seq_ids, int32_t width, int32_t * table, int32_t * lens, int32_t * all_regular);
int32_t pkv_cache_seq_slot_pos (const pkv_cache * c, int32_t seq_id, int32_t * out, int32_t cap);
int32_t pkv_cache_n_free_blocks(const pkv_cache * c);
int32_t pkv_cache_block_refcount(const pkv_cache * c, int32_t block);
int32_t pkv_cache_has_shift    (const pkv_cache * c);
/* device work queued by host edits and not yet applied by pkv_cache_update: rows of
 * {is_cell_copy, src_block, src_off, dst_block, dst_off} (whole-block copy-on-write copies have offsets 0) */
int32_t pkv_cache_n_pending    (const pkv_cache * c);
int32_t pkv_cache_pending      (const pkv_cache * c, int32_t * out5, int32_t cap);

/* device side of the cache (n_layer > 0) */
void * pkv_cache_k_pool(pkv_cache * c, int layer);
void * pkv_cache_v_pool(pkv_cache * c, int layer);
/* kv_self_update (/ref/src/llama-context.cpp:719-803): apply pending copy-on-write block copies and the
 * pending K-shift (has_shift) to every layer on `stream`, then clear the deltas. */
int pkv_cache_update(pkv_cache * c, const pkv_rope_params * rope, void * stream);

/* ---- FIRST SLICE of the rest of the decoder layer (correct and parity-checked, not yet tuned) ----
 * What a llama layer needs besides the attention path, so that the reference runtime can keep whole layers on the device
 * through the ggml-backend plugin.  All tensors device, f32, contiguous rows.
 *   pkv_rms_norm_f32 : ggml_rms_norm, y[r] = x[r] / sqrt(mean(x[r]^2) + eps)   (/ref/ggml/src/ggml-cpu/ggml-cpu.c:8215-8260)
 *   pkv_add_f32 / pkv_mul_f32 : y[r][i] = a[r][i] (+|*) b[r % b_rows][i]      (ggml_add / ggml_mul with row broadcast)
 *   pkv_silu_f32     : x / (1 + exp(-x))                                       (ggml_silu, :2015)
 *   pkv_mul_mat      : ggml_mul_mat, y[c][m] = sum_k W[m][k] x[c][k]; W = M rows of K elements in f32 / f16 / q8_0 / q4_K (reference
 *                      row format; q4_K dequantized as dequantize_row_q4_K, /ref/ggml/src/ggml-quants.c:1280-1302), x = n_cols rows
 *                      of K f32, y = n_cols rows of M f32   (:7467-7640) */
int pkv_rms_norm_f32(const float * x, float * y, int64_t n_rows, int64_t ne0, float eps, void * stream);
/* the two pairs every llama layer runs back to back, as one launch each: y = rms_norm(x) * w (w = one row of ne0 weights) and
 * y = silu(g) * u; y_norm / y_silu receive the intermediate when non-NULL (a graph that still names it) */
int pkv_rms_norm_mul_f32(const float * x, const float * w, float * y_norm, float * y, int64_t n_rows, int64_t ne0, float eps, void * stream);
int pkv_silu_mul_f32(const float * g, const float * u, float * y_silu, float * y, int64_t n, void * stream);
/* the residual add in front of the norm in the same launch: sum = a + b (same shape; sum may be a or b: in-place ADD of ggml-alloc),
 * y = rms_norm(sum) * w — ffn_inp = attn_out + inpL -> ffn_norm, l_out = ffn_out + ffn_inp -> the next layer's attn_norm */
int pkv_add_rms_norm_mul_f32(const float * a, const float * b, float * sum, const float * w, float * y_norm, float * y, int64_t n_rows, int64_t ne0, float eps,
                             void * stream);
int pkv_add_f32(const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows, void * stream);
int pkv_mul_f32(const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows, void * stream);
int pkv_silu_f32(const float * x, float * y, int64_t n, void * stream);
/* ggml_soft_max_ext (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10222-10318; the softmax of the non-flash-attention path, /ref/src/llama-graph.cpp:1201-1239):
 * y[i3][i2][i1][:] = softmax(x[...][:] * scale + slope(i2) * mask[i1][:]); x, y f32 contiguous [ne03][ne02][ne01][ne00] (y may alias x); mask NULL or
 * [>= ne01][ne00] rows of f32 (mask_f16 = 0) or f16 (1), broadcast over i2, i3; slope = ALiBi (max_bias > 0: m0^(h+1) / m1^(2(h-n_head_log2)+1), h = i2) else 1 */
/* The two generic ops a non-flash-attention graph needs beside the softmax (`-fa 0`: /ref/src/llama-graph.cpp:1201-1239, transposed V store :1387-1394).
 * Shapes / strides as ggml's ne[4] / nb[4] (elements / BYTES, dimension 0 first).
 *   pkv_cpy_strided     : ggml_cpy / ggml_cont between strided f32 / f16 tensors of equal element count (element i of src in its index order -> element i of dst)
 *   pkv_mul_mat_strided : ggml_mul_mat with a strided, batched, broadcast f16 / f32 first operand [K, M, b2, b3] (unit stride in K) and an f32 second operand
 *                         [K, N, B2, B3] (unit stride in K; B2 % b2 == 0, B3 % b3 == 0); y f32 contiguous [M, N, B2, B3] */
int pkv_cpy_strided(const void * src, int src_type, const int64_t * ne_src, const int64_t * nb_src, void * dst, int dst_type, const int64_t * ne_dst, const int64_t * nb_dst,
                    void * stream);
int pkv_mul_mat_strided(const void * w, int type_w, const int64_t * ne_w, const int64_t * nb_w, const float * x, const int64_t * ne_x, const int64_t * nb_x, float * y,
                        void * stream);
int pkv_soft_max_ext_f32(const float * x, const void * mask, int mask_f16, float * y, int64_t ne00, int64_t ne01, int64_t ne02, int64_t ne03, float scale,
                         float max_bias, void * stream);
int pkv_mul_mat(const void * w, int type_w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, void * stream);
/* n (1..PKV_MUL_MAT_MAX) products that share x, K, n_cols and the weight type — wq / wk / wv of a layer, gate / up of its FFN
 * (/ref/src/llama-model.cpp build_attn inputs, build_ffn LLM_FFN_PAR) — results identical to n pkv_mul_mat calls; q4_K and q8_0
 * weights run as ONE launch (at decode a product is mostly its launch-and-latency floor) */
#define PKV_MUL_MAT_MAX 4
int pkv_mul_mat_multi(int n, const void * const * w, int type_w, const float * x, float * const * y, const int64_t * M, int64_t K, int64_t n_cols,
                      void * stream);

/* ---- decode-step weight products (one activation column) for K-quant weights: csrc/pkv_gemv.hip ------------------------------
 * The activation column is quantized once to int8 per 32 elements (the reference's CPU backend quantizes it to q8_K per 256 before a
 * q4_K / q5_K / q6_K product, /ref/ggml/src/ggml-cpu/ggml-cpu.c:7467-7640) into an "activation image" of pkv_act_image_bytes(K) bytes:
 * by pkv_quantize_act, or by pkv_rms_norm_mul_q (rms_norm(x [+ res]) * w straight into the image: build_norm feeding a product).
 * pkv_gemv_q then streams the rows of up to three matrices that share the column in ONE launch (wq + wk + wv; gate + up), with the
 * epilogues a llama layer needs: y = residual + W x (attn_out / ffn_down onto the residual stream, :4120,:4160) or
 * y = silu(W0 x) * (W1 x) (LLM_FFN_SILU + LLM_FFN_PAR of build_ffn).  Rows are block_q4_K / block_q5_K / block_q6_K in the reference's format
 * (per matrix: a Q4_K_M / Q5_K_M file keeps attn_v / ffn_down / output in q6_K), K % 256 == 0, M even; the matrices of one launch may differ in type. */
#define PKV_GEMV_MAX 3
typedef struct pkv_gemv {
    int32_t       n;                   /* products that share the activation image (1..PKV_GEMV_MAX)                          */
    int32_t       silu_mul;            /* 1: n == 2, M[0] == M[1]: y[0][m] = silu(W0 x)[m] * (W1 x)[m]; y[1] is not written    */
    int32_t       type_w[PKV_GEMV_MAX];
    const void *  w[PKV_GEMV_MAX];     /* device rows, reference format                                                       */
    float *       y[PKV_GEMV_MAX];     /* device f32 [M[i]]                                                                   */
    int64_t       M[PKV_GEMV_MAX];
    int64_t       K;
    const void *  xq;                  /* activation image, or NULL: the kernel builds it itself from `x` (below)              */
    const float * residual;            /* optional (n == 1): y[0][m] = residual[m] + (W x)[m]; may alias y[0]                 */
    const float * x;                   /* xq == NULL: the f32 activation column [K] (16-byte aligned); every workgroup quantizes   */
    const float * norm_w;              /*   it into LDS under its first weight loads — after rms_norm(x) * norm_w when norm_w != NULL */
    float         eps;                 /*   (build_norm feeding the product: no norm launch, no quantize launch; K <= 8192 then)   */
    float *       norm_out;            /* optional, with norm_w: rms_norm(x) [K] and                                              */
    float *       col_out;             /*   rms_norm(x) * norm_w [K] are ALSO written (by one workgroup): a graph that names them     */
} pkv_gemv;
size_t pkv_act_image_bytes(int64_t K);
int pkv_quantize_act(const float * x, void * image, int64_t K, void * stream);
/* x_out = x + res (res, x_out both or neither; x_out may alias x), image = quantize(rms_norm(x_out) * w); y (optional) = the f32 row */
int pkv_rms_norm_mul_q(const float * x, const float * res, float * x_out, const float * w, float * y, void * image, int64_t ne0, float eps, void * stream);
int pkv_gemv_q(const pkv_gemv * g, void * stream);
/* PLANE layout of K-quant weight rows (csrc/pkv_wlayout.h): the same bytes re-laid INSIDE each row at upload so that a wave instruction of the
 * decode-step products reads 1 KiB of contiguous bytes per plane (q4_K row: [16 B of low-half nibbles per 64-element group][16 B of the other half]
 * [16-byte block headers]; q5_K row: the same two planes, then [qh 0..15][qh 16..31][headers] per block; q6_K row: [ql lo][ql hi][qh][scales][d]) — the GGUF block order puts every lane's load at a 144- / 210-byte stride.
 * Precedent: the reference's CPU backend repacks weights on set_tensor (/ref/ggml/src/ggml-cpu/ggml-cpu-aarch64.cpp:3663-3810, :4145-4172).
 * A packed tensor is named by `type | PKV_TYPE_PACKED` in pkv_gemv.type_w[], pkv_mul_mat, pkv_mul_mat_multi and pkv_get_rows; results are bit-identical
 * to the GGUF-order tensor's.  pkv_weights_pack converts n_rows rows of K elements (src != dst; inverse = 1: back to GGUF order, byte-exact). */
#define PKV_TYPE_PACKED 0x100
int pkv_weights_pack(const void * src, void * dst, int type, int64_t n_rows, int64_t K, int inverse, void * stream);
/* ggml_get_rows (/ref/src/llama-model.cpp:4047: inp_embd = get_rows(tok_embd, tokens)): out[i][:] = dequantize(row ids[i] of w);
 * w = n_rows_w rows of K elements in f32 / f16 / q8_0 / q4_K / q5_K / q6_K; ids device i32 [n_ids]; out device f32 [n_ids][K] */
int pkv_get_rows(const void * w, int type_w, const int32_t * ids, float * out, int64_t n_ids, int64_t K, int64_t n_rows_w, void * stream);

/* Session state of the KV cache in the reference's byte format: llama_kv_cache_unified::state_write / state_read
 * (/ref/src/llama-kv-cache.cpp:848-1221; reached through llama_state_get_data / llama_state_seq_get_data,
 * /ref/include/llama.h:793-876).  Layout: u32 cell_count | per cell {i32 pos, u32 n_seq_id, i32 seq_id[n_seq_id]} |
 * u32 v_trans | u32 n_layer | per layer {i32 type_k, u64 k_size_row, K rows} | per layer {i32 type_v, u64 v_size_row,
 * V rows}; one row = one cell = row_size(type, n_head_kv * head_dim) bytes exactly as the reference stores it.
 * seq_id >= 0: that sequence's cells (n_seq_id written as 0); seq_id == -1: every occupied cell with its sequence set,
 * in ascending (block, offset) order.  The rows are gathered through the block table on save and scattered on load
 * ("gather-on-save"), so files are interchangeable with the reference's flat cell ring.  v_trans is written as 0 (the
 * flash-attention layout); files with v_trans = 1 (f16 V of `-fa 0` runs) are read and transposed back.
 * host pointers; a bookkeeping-only cache (n_layer == 0) reads/writes the metadata alone and needs no GPU.
 * state_size / state_write return the byte count (>= 0) or a negative PKV_ERR_*; pending device work (copies, K-shift)
 * must have been applied by pkv_cache_update first.  state_read on failure leaves the cache cleared (seq_id == -1) or
 * the sequence removed, like the reference (:888-896). */
int64_t pkv_cache_state_size(pkv_cache * c, int32_t seq_id);
int64_t pkv_cache_state_write(pkv_cache * c, int32_t seq_id, void * dst, size_t capacity, void * stream);
int     pkv_cache_state_read(pkv_cache * c, int32_t seq_id, const void * src, size_t n_bytes, void * stream);

/* Debug view of the cache: llama_kv_cache_view_update (/ref/src/llama-kv-cache.cpp:1364-1433, struct
 * /ref/include/llama.h:593-630) over the PHYSICAL cells, cell i = (block i / block_size, offset i % block_size).
 * cell_pos: i32 [n_cells] (pos + pending delta, -1 = empty); cell_seqs: i32 [n_cells][n_seq_max] padded with -1 (ids beyond
 * n_seq_max are not shown, as in the reference); stats: i32 [5] = {n_cells, token_count, used_cells, max_contiguous,
 * max_contiguous_idx} with the reference's meaning (longest run of empty cells, -1 when the cache is full).
 * Returns n_cells (= n_blocks * block_size); arrays may be NULL to query the size only. */
int32_t pkv_cache_view_update(const pkv_cache * c, int32_t n_seq_max, int32_t * cell_pos, int32_t * cell_seqs, int32_t * stats);

/* ---- the fork prototype's toy classes (sources absent from /ref; semantics recovered from
 * /ref/tests/test_block_manager.cpp, test_kv_cache.cpp, test_block_kv_cache.cpp and the ELF symbols,
 * SURVEY.md §8(a-toy)) ---------------------------------------------------------------------------- */
typedef struct pkv_toy_block_manager pkv_toy_block_manager;
pkv_toy_block_manager * pkv_toy_bm_create(int n_blocks);
void pkv_toy_bm_free    (pkv_toy_block_manager * bm);
int  pkv_toy_bm_allocate(pkv_toy_block_manager * bm);            /* BlockManager::allocateBlock  */
void pkv_toy_bm_release (pkv_toy_block_manager * bm, int block); /* BlockManager::freeBlock      */
int  pkv_toy_bm_status  (const pkv_toy_block_manager * bm, uint8_t * out, int cap); /* getBlockStatus */

typedef struct pkv_toy_kv_manager pkv_toy_kv_manager;            /* KVCacheManager + PhysicalBlock */
pkv_toy_kv_manager * pkv_toy_kvm_create(int slot_capacity);
void pkv_toy_kvm_free  (pkv_toy_kv_manager * m);
int  pkv_toy_kvm_insert(pkv_toy_kv_manager * m, int * physical_block, int * slot); /* VirtualBlockEntry */
int  pkv_toy_kvm_release(pkv_toy_kv_manager * m, int physical_block, int slot);

/* BlockKVCache + block_kv_attention toys (/ref/tests/test_block_kv_cache.cpp, test_block_kv_attention.cpp,
 * test_paged_attention.cpp): host float storage, block element count be = block_size*num_kv_heads*head_size */
typedef struct pkv_toy_block_kv_cache pkv_toy_block_kv_cache;
pkv_toy_block_kv_cache * pkv_toy_bkc_create(int num_blocks, int block_size, int num_kv_heads, int head_size);
void pkv_toy_bkc_free(pkv_toy_block_kv_cache * c);
void pkv_toy_bkc_init_cache(pkv_toy_block_kv_cache * c);
void pkv_toy_bkc_write_to_cache(pkv_toy_block_kv_cache * c, const float * k, const float * v, const int * slot_mapping, int n);
void pkv_toy_bkc_swap_blocks(pkv_toy_block_kv_cache * c, pkv_toy_block_kv_cache * other, const int * mapping, int n);
void pkv_toy_bkc_copy_blocks(pkv_toy_block_kv_cache * c, const pkv_toy_block_kv_cache * src, const int * mapping, int n);
int  pkv_toy_bkc_insert_kv_block(pkv_toy_block_kv_cache * c, const float * k, const float * v, int * physical_block, int * slot);
int  pkv_toy_bkc_free_kv_block(pkv_toy_block_kv_cache * c, int physical_block, int slot);
int  pkv_toy_bkc_get(const pkv_toy_block_kv_cache * c, int which_v, float * out, int cap);   /* getKeyCache / getValueCache */
void pkv_toy_attention_fwd(const float * q, int n, pkv_toy_block_kv_cache * c, const float * k, const float * v,
                           const int * slot_mapping, int n_map, float * out);               /* blockKVAttentionFwd      */
void pkv_toy_attention_paged_fwd(const float * q, int n, pkv_toy_block_kv_cache * c, const float * k, const float * v,
                                 float * out, int page_size);                               /* blockKVAttentionPagedFwd */
void pkv_toy_attention_real_fwd(const float * q, int n, const float * k, const float * v, float * out); /* ...RealFwd   */

#ifdef __cplusplus
}
#endif
#endif /* MI355_PAGED_KV_H */
