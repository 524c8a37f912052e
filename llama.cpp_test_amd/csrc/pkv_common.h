// pkv_common.h — shared host/device helpers for the MI355X paged-KV path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mi355_paged_kv.h"

#define PKV_QK 32  // elements per quant block: QK8_0 == QK4_0 (/ref/ggml/src/ggml-common.h:155,197)

namespace pkv {

// ---- status / error text -------------------------------------------------------------------------
void set_error(const char * fmt, ...);
int  hip_fail(hipError_t e, const char * what);
#define PKV_HIP(call)                                                       \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return ::pkv::hip_fail(e_, #call);            \
    } while (0)
#define PKV_REQUIRE(cond, ...)                                              \
    do {                                                                    \
        if (!(cond)) { ::pkv::set_error(__VA_ARGS__); return PKV_ERR_INVALID; } \
    } while (0)

int ensure_device();  // PKV_OK or PKV_ERR_NO_DEVICE

// ---- pool layout ("Pool layout" in include/mi355_paged_kv.h) -------------------------------------
__host__ __device__ inline size_t tile_quant_bytes(int type, int block_size, int D) {
    return type == PKV_TYPE_F16  ? (size_t) block_size * D * 2
         : type == PKV_TYPE_Q8_0 ? (size_t) block_size * D
         : type == PKV_TYPE_Q4_0 ? (size_t) block_size * D / 2 : 0;
}
__host__ __device__ inline size_t tile_bytes(int type, int block_size, int D) {
    const size_t q = tile_quant_bytes(type, block_size, D);
    return type == PKV_TYPE_F16 ? q : (q ? q + (size_t) block_size * (D / PKV_QK) * 2 : 0);
}
__host__ __device__ inline size_t row_size(int type, int64_t ne) {
    return type == PKV_TYPE_F32  ? (size_t) ne * 4
         : type == PKV_TYPE_F16  ? (size_t) ne * 2
         : type == PKV_TYPE_Q8_0 ? (size_t) (ne / PKV_QK) * 34
         : type == PKV_TYPE_Q4_0 ? (size_t) (ne / PKV_QK) * 18
         : type == PKV_TYPE_Q4_1 ? (size_t) (ne / PKV_QK) * 20
         : type == PKV_TYPE_Q5_0 ? (size_t) (ne / PKV_QK) * 22
         : type == PKV_TYPE_Q5_1 ? (size_t) (ne / PKV_QK) * 24
         : type == PKV_TYPE_IQ4_NL ? (size_t) (ne / PKV_QK) * 18
         : type == PKV_TYPE_BF16 ? (size_t) ne * 2
         : type == PKV_TYPE_Q4_K ? (size_t) (ne / 256) * 144
         : type == PKV_TYPE_Q5_K ? (size_t) (ne / 256) * 176
         : type == PKV_TYPE_Q6_K ? (size_t) (ne / 256) * 210 : 0;
}
inline bool kv_type_ok(int t) { return t == PKV_TYPE_F16 || t == PKV_TYPE_Q8_0 || t == PKV_TYPE_Q4_0; }
inline bool flat_only_type(int t) { return t == PKV_TYPE_Q4_1 || t == PKV_TYPE_Q5_0 || t == PKV_TYPE_Q5_1 || t == PKV_TYPE_IQ4_NL; }   // cache types without a pool layout

// ---- canonical sum of squares of an RMS norm -------------------------------------------------------
// rms_norm_mul_kernel (pkv_layer_ops.hip) and the norm prologue of the products (pkv_gemv.hip) must produce the same bits whatever their
// workgroup shape.  ggml_compute_forward_rms_norm_f32 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:8215-8260) adds (ggml_float)(x * x): f32 squares,
// f64 sum, mean = (float)(sum / ne00), scale = 1 / sqrtf(mean + eps).  Same here, in a fixed order: per 256-element block (one wave, one
// float4 per lane) x, y, z, w in order, a butterfly over the wave (partners 1, 2, 4, 8, 16, 32 lanes apart), then the blocks in index order.
#ifdef __HIPCC__
// all-reduce sum of a double over the wave without LDS (__shfl_xor on a double is two ds_bpermute round trips per step: 0.6 us per block on the
// critical path of every norm): DPP inside the rows of 16, permlane swaps across them; lanes 1, 2, 4, 8, 16, 32 apart in that order
template <int CTRL> __device__ __forceinline__ double dpp_d(double x) {
    const unsigned long long u = __double_as_longlong(x);
    const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (u >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double x) {
    x += dpp_d<0xB1>(x); x += dpp_d<0x4E>(x); x += dpp_d<0x141>(x); x += dpp_d<0x140>(x);
    {
        const unsigned long long u = __double_as_longlong(x);
        auto l = __builtin_amdgcn_permlane16_swap((unsigned) u, (unsigned) u, false, false);
        auto h = __builtin_amdgcn_permlane16_swap((unsigned) (u >> 32), (unsigned) (u >> 32), false, false);
        x = __longlong_as_double(((unsigned long long) h[0] << 32) | l[0]) + __longlong_as_double(((unsigned long long) h[1] << 32) | l[1]);
    }
    {
        const unsigned long long u = __double_as_longlong(x);
        auto l = __builtin_amdgcn_permlane32_swap((unsigned) u, (unsigned) u, false, false);
        auto h = __builtin_amdgcn_permlane32_swap((unsigned) (u >> 32), (unsigned) (u >> 32), false, false);
        x = __longlong_as_double(((unsigned long long) h[0] << 32) | l[0]) + __longlong_as_double(((unsigned long long) h[1] << 32) | l[1]);
    }
    return x;
}
__device__ __forceinline__ double block_sumsq(const float4 v) {
    double s = (double) __fmul_rn(v.x, v.x);
    s += (double) __fmul_rn(v.y, v.y);
    s += (double) __fmul_rn(v.z, v.z);
    s += (double) __fmul_rn(v.w, v.w);
    return wave_sum_d(s);
}
// (n_blocks <= 64.)  ONE LDS read per lane — lane l holds block l — then the blocks are added in index order out of registers (v_readlane): the
// serial loop over LDS paid one read latency per block, 0.6 us for the 16 blocks of a 4096-element row on the critical path of every norm
// (profiles/r03_gemv_stamps.txt).  Same additions in the same order: the same bits.
__device__ __forceinline__ float rms_scale(const double * block_sums, int n_blocks, int64_t ne0, float eps) {
    const int l_ = (int) (threadIdx.x & 63);
    const unsigned long long mine = __double_as_longlong(l_ < n_blocks ? block_sums[l_] : 0.0);
    const int lo = (int) (unsigned) mine, hi = (int) (unsigned) (mine >> 32);
    double t = 0.0;
    for (int b = 0; b < n_blocks; ++b) {
        const unsigned lb = (unsigned) __builtin_amdgcn_readlane(lo, b), hb = (unsigned) __builtin_amdgcn_readlane(hi, b);
        t += __longlong_as_double(((unsigned long long) hb << 32) | lb);
    }
    const float mean = (float) (t / (double) ne0);
    return 1.0f / sqrtf(mean + eps);
}
#endif

// ---- how a decode launch is cut into workgroups (host logic) ------------------------------------------
// Tiles per workgroup of the multi-tile form (1 = the single-tile form).  A launch of one 128-token tile per workgroup pays its fixed work — scalars, the Q
// operand, the cross-wave merge, a partial, a share of the merge launch — once per 128 tokens; but it also keeps two workgroups per CU resident (gqa <= 4: 118
// VGPRs; gqa 8: 167, one workgroup), and a launch of a few hundred multi-tile workgroups has LESS in flight per CU than the same launch cut into single tiles.
// Measured (profiles/r03_decode_tiles.txt, r03_decode_rule_sweep.txt; ms per step of all layers): 70B geometry ctx 16384 (gqa 8, 1024 tiles) 1.99 at 1 tile per
// workgroup, 1.77 at 2, 1.46 at 4 (= 256 workgroups), 2.00 at 8; 8B geometry ctx 4096: 32 sequences 3.09 -> 2.85 (8 tiles), 16 sequences 1.58 -> 1.54, but 3 / 6
// sequences 0.44 -> 0.56 / 0.72 -> 0.88 with 3 / 6 tiles (264 uneven workgroups); 32 sequences x ctx 640: 0.66 -> 0.52 with one workgroup per (sequence, kv
// head) (no partials, no merge launch), 24 sequences with 4 + 1 tiles 0.55 -> 0.67.  Rule: (1) enough streams to fill the chip and few tiles: one workgroup per
// stream; (2) otherwise several tiles only from four rounds of resident workgroups on, a power of two that leaves at least one round and divides the
// sequence's tiles evenly (or nearly so), at most DECODE_MAX_TILES (the bound keeps one long sequence among short ones from serialising the launch).
constexpr int DECODE_CUS = 256, DECODE_MAX_TILES = 8, DECODE_TILE_TOKENS = 128;
inline int decode_tiles_per_wg(int n_splits, int n_head_kv, int n_seq, int gqa) {
    if (const char * e = getenv("PKV_DECODE_NTILE")) { const int v = atoi(e); if (v >= 1) return v < n_splits ? v : n_splits; }   // experiments
    const long streams = (long) n_head_kv * n_seq, wg1 = streams * n_splits;
    if (n_splits <= 1) return 1;
    if (streams >= DECODE_CUS && n_splits <= DECODE_MAX_TILES) return n_splits;
    const long slots = (long) DECODE_CUS * (gqa <= 4 ? 2 : 1);           // workgroups resident at a time
    if (wg1 < 4 * slots) return 1;
    const long lim = wg1 / slots;
    for (int nt = DECODE_MAX_TILES; nt >= 2; nt >>= 1)
        if (nt <= lim && (n_splits % nt == 0 || n_splits >= 8 * nt)) return nt;
    return 1;
}
inline int decode_tiles_for(int max_seq_len, int n_head, int n_head_kv, int n_seq) {   // pkv_attn_decode_tiles
    int g = n_head_kv > 0 ? n_head / n_head_kv : 1;
    if (g == 16 || g == 32) { n_head_kv *= g / 8; g = 8; }                // (run as gqa 8 over virtual kv heads)
    return decode_tiles_per_wg((max_seq_len + DECODE_TILE_TOKENS - 1) / DECODE_TILE_TOKENS, n_head_kv, n_seq, g);
}

// ---- kernel launchers (pkv_kernels_*.hip) ---------------------------------------------------------
struct RopeDev {  // pkv_rope_params resolved on the host (corr dims = ggml_rope_yarn_corr_dims, /ref/ggml/src/ggml.c:3739)
    int   n_dims, mode;
    float freq_scale, ext_factor, attn_factor, theta_scale, corr0, corr1, yarn_mscale;
    const float * freq_factors;
};
int make_rope(const pkv_rope_params * p, int head_dim, RopeDev * out);

int launch_rope_f32(const float * x, const int32_t * pos, float * out, int n_tok, int n_head, int D,
                    const RopeDev & r, int inverse, hipStream_t st, int64_t xs_tok = 0, int64_t xs_head = 0);
int launch_rope_cache(const int32_t * pos, float * cache, int n_tok, int D, const RopeDev & r, hipStream_t st);
int launch_rope_f16(const void * x, const int32_t * pos, void * out, int n_tok, int n_head, int D, const RopeDev & r,
                    int inverse, hipStream_t st, int64_t xs_tok = 0, int64_t xs_head = 0);
int launch_kv_store(float * q, int n_head_q, const float * k_cur, const float * v_cur, void * k_pool, void * v_pool,
                    int type_k, int type_v, const int32_t * slot_block, const int32_t * slot_off, const int32_t * pos,
                    const RopeDev * rope, int n_tok, int n_head_kv, int D, int block_size, hipStream_t st, const float * rope_cache = nullptr);
int launch_attn_decode(const pkv_attn_args & a, hipStream_t st, const pkv_step_args * step = nullptr);
int launch_attn_prefill(const pkv_attn_args & a, hipStream_t st, const float * rope_cache = nullptr);   // rope_cache: RoPE(Q) fused in (NORM pairs)
int launch_attn_flat(const pkv_flat_attn_args & a, hipStream_t st);   // pkv_attn_flat.hip
size_t flat_workspace_bytes(int n_q, int n_head, int D, int n_kv);
int launch_quantize_blocks(const float * x, void * dst, int type, int64_t n_rows, int64_t ne0, hipStream_t st);   // q4_1 / q5_0 / q5_1 / iq4_nl (pkv_attn_flat.hip)
int launch_attn_decode_stamps(const pkv_attn_args & a, const pkv_step_args * step, unsigned long long * stamps, int mode, hipStream_t st);
size_t decode_workspace_bytes(int n_tok, int n_head, int D, int max_seq_len);
size_t prefill_workspace_bytes(int n_tok, int n_head, int D);
void set_prefill_stamps(void * dev_u64);  // diagnostic
int launch_k_shift(void * k_pool, int type_k, const int32_t * cell_block, const int32_t * cell_off,
                   const int32_t * cell_delta, int n_cells, int n_head_kv, int D, int block_size, const RopeDev & r,
                   hipStream_t st);
int launch_copy_blocks(void * pool, size_t block_bytes, const int32_t * src, const int32_t * dst, int n, hipStream_t st);
int launch_copy_cells(void * pool, int type, int is_v, const int32_t * sb, const int32_t * so, const int32_t * db,
                      const int32_t * dof, int n, int n_head_kv, int D, int block_size, hipStream_t st);
int launch_rows_io(void * pool, int type, int is_v, void * rows, const int32_t * slot_block, const int32_t * slot_off, int n_tok,
                   int n_head_kv, int D, int block_size, bool scatter, hipStream_t st, int64_t rs_tok = 0, int64_t rs_head = 0);
// first slice of the rest of the decoder layer (pkv_layer_ops.hip)
int launch_rms_norm(const float * x, float * y, int64_t n_rows, int64_t ne0, float eps, hipStream_t st);
int launch_rms_norm_mul(const float * x, const float * res, float * x_out, const float * w, float * y_norm, float * y, int64_t n_rows, int64_t ne0, float eps,
                        hipStream_t st);
int launch_silu_mul(const float * g, const float * u, float * y_silu, float * y, int64_t n, hipStream_t st);
int launch_binary(int op, const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows, hipStream_t st);
int launch_silu(const float * x, float * y, int64_t n, hipStream_t st);
int launch_cpy_strided(const void * src, int src_type, const int64_t * ne_s, const int64_t * nb_s, void * dst, int dst_type, const int64_t * ne_d, const int64_t * nb_d, hipStream_t st);
int launch_mul_mat_strided(const void * w, int type_w, const int64_t * ne_w, const int64_t * nb_w, const float * x, const int64_t * ne_x, const int64_t * nb_x, float * y,
                           hipStream_t st);
int launch_soft_max(const float * x, const void * mask, int mask_f16, float * y, int64_t ne00, int64_t ne01, int64_t ne02, int64_t ne03, float scale, float max_bias,
                    hipStream_t st);
int launch_mul_mat(const void * w, int type_w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, hipStream_t st);
int launch_mul_mat_multi(int n, const void * const * w, int type_w, const float * x, float * const * y, const int64_t * M, int64_t K, int64_t n_cols,
                         hipStream_t st);
// batched products on the matrix cores (pkv_gemm.hip)
bool gemm_supported(int type_w, int64_t M, int64_t K);
int launch_gemm(const void * w, int type_w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, hipStream_t st);
int launch_gemm_multi(int n, const void * const * w, int type_w, const float * x, float * const * y, const int64_t * M, int64_t K, int64_t n_cols, hipStream_t st);   // n <= 4 matrices of one type, same x
int launch_convert_rows(const float * x, void * dst, int type, int64_t n_rows, int64_t ne0, hipStream_t st);
// decode-step products for K-quant weights (pkv_gemv.hip)
size_t act_image_bytes(int64_t K);
int launch_quantize_act(const float * x, void * img, int64_t K, hipStream_t st);
int launch_gemv(const pkv_gemv & g, hipStream_t st);
int launch_pack_rows(const void * src, void * dst, int type, int64_t n_rows, int64_t K, int inverse, hipStream_t st);
int launch_get_rows(const void * w, int type_w, const int32_t * ids, float * out, int64_t n_ids, int64_t K, int64_t n_rows_w, hipStream_t st);
int launch_rms_norm_mul_q(const float * x, const float * res, float * x_out, const float * w, float * y, void * img, int64_t ne0, float eps, hipStream_t st);
int launch_qkv_store_mirror(const pkv_qkv_store_args * p, const RopeDev & r, hipStream_t st);
int launch_rope_convert_mirror(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, float * roped, void * dst, int type, void * pool,
                               int64_t cell0, int64_t n_tok, int n_head_kv, int D, int BS, const RopeDev & r, hipStream_t st);
int launch_convert_rows_mirror(const float * x, void * dst, int type, void * pool, int is_v, int64_t cell0, int64_t n_head_rows, int n_head_kv, int D, int BS,
                               hipStream_t st);

}  // namespace pkv
