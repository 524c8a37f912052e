// pkv_kv_ops.hip — RoPE, fused KV store (RoPE + f16/q8_0/q4_0 convert + scatter through the slot mapping),
// in-place K-shift, reference-format row I/O and block copies for the paged pools.  gfx950 only.
//
// One 64-lane wave handles one (token, kv head) row; lane i owns elements (2i, 2i+1), i.e. one RoPE pair in
// NORM mode (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10807-10821).  Quantization block reductions (32 elements = 16
// lanes = one DPP row) use DPP only.
#include "pkv_common.h"
#include <cstdlib>

// every kernel of this file must round a given expression the same way (bit-exact parity between the fused and
// the separate launches): no implicit FMA contraction
#pragma clang fp contract(off)

#include "pkv_rowcodec.h"

namespace pkv {
namespace {

// reference-format row: f16[D] / block_q8_0[D/32] / block_q4_0[D/32] (/ref/ggml/src/ggml-common.h:161-166,203-208)
template <int T> __device__ __forceinline__ void ref_write(char * row, int lane, const Q2 & v, int D) {
    const int e0 = 2 * lane;
    if constexpr (T == PKV_TYPE_Q4_0) {
        const int p0 = __shfl_xor(v.q0, 8), p1 = __shfl_xor(v.q1, 8);
        if (e0 >= D) return;
        char * b = row + (size_t) (e0 / PKV_QK) * 18;
        if ((lane & 15) < 8) *reinterpret_cast<uint16_t *>(b + 2 + 2 * (lane & 15)) = (uint16_t) ((v.q0 | (p0 << 4)) | ((v.q1 | (p1 << 4)) << 8));
        if ((lane & 15) == 0) *reinterpret_cast<uint16_t *>(b) = v.d;
    } else {
        if (e0 >= D) return;
        if constexpr (T == PKV_TYPE_F16) {
            *reinterpret_cast<uint32_t *>(row + (size_t) e0 * 2) = (uint32_t) (v.q0 & 0xFFFF) | ((uint32_t) v.q1 << 16);
        } else {
            char * b = row + (size_t) (e0 / PKV_QK) * 34;
            *reinterpret_cast<uint16_t *>(b + 2 + (e0 & 31)) = (uint16_t) ((v.q0 & 0xFF) | ((v.q1 & 0xFF) << 8));
            if ((lane & 15) == 0) *reinterpret_cast<uint16_t *>(b) = v.d;
        }
    }
}
template <int T> __device__ __forceinline__ Q2 ref_read(const char * row, int lane, int D) {
    Q2 r{0, 0, 0};
    const int e0 = 2 * lane;
    if (e0 >= D) return r;
    if constexpr (T == PKV_TYPE_F16) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(row + (size_t) e0 * 2);
        r.q0 = w & 0xFFFF; r.q1 = w >> 16;
    } else if constexpr (T == PKV_TYPE_Q8_0) {
        const char *   b  = row + (size_t) (e0 / PKV_QK) * 34;
        const uint16_t pk = *reinterpret_cast<const uint16_t *>(b + 2 + (e0 & 31));
        r.q0 = (int) (signed char) (pk & 0xFF); r.q1 = (int) (signed char) (pk >> 8);
        r.d  = *reinterpret_cast<const uint16_t *>(b);
    } else {
        const char *   b  = row + (size_t) (e0 / PKV_QK) * 18;
        const uint16_t pk = *reinterpret_cast<const uint16_t *>(b + 2 + 2 * ((lane & 15) & 7));
        const int      sh = (lane & 15) < 8 ? 0 : 4;
        r.q0 = (pk >> sh) & 0xF; r.q1 = (pk >> (8 + sh)) & 0xF;
        r.d  = *reinterpret_cast<const uint16_t *>(b);
    }
    return r;
}

// ---- kernels ----------------------------------------------------------------------------------------------
// ggml_rope_cache_init (:10595-10609) for every token of a micro-batch: cache[t][ip] = (cos, sin) * mscale of pair ip at
// pos[t]; entries past n_dims/2 are the identity rotation.  One row serves all heads and all layers of the step.
__global__ void __launch_bounds__(64) rope_cache_kernel(const int32_t * pos, float2 * cache, int half_d, const RopeDev r) {
    const int t = blockIdx.x;
    for (int ip = threadIdx.x; ip < half_d; ip += 64)
        cache[(size_t) t * half_d + ip] = 2 * ip < r.n_dims ? rope_cs(pos[t], ip, r, 1.0f) : make_float2(1.0f, 0.0f);
}

struct StoreArgs {
    float * q; int n_head_q;
    const float * k_cur; const float * v_cur;
    char * k_pool; char * v_pool;
    const int32_t * slot_block; const int32_t * slot_off; const int32_t * pos;
    int n_head_kv, block_size, has_rope;
    RopeDev rope;
    const float * rope_cache;   // optional pkv_rope_cache rows of these tokens [n_tok][D/2][2] (NORM pairs): read instead of evaluated (kv_store_tok_kernel)
};

// grid (max(n_head_kv, n_head_q), n_tok, 2 or 3): z = 0 stores K (optionally roped), z = 1 stores V,
// z = 2 rotates the Q heads in place (pkv_qkv_prepare: the three per-token ops of a decode step in ONE launch)
template <int KT, int VT, int D>
__global__ void __launch_bounds__(64) kv_store_kernel(const StoreArgs a) {
    const int h = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    if (blockIdx.z == 2) {
        if (h >= a.n_head_q || 2 * lane >= D) return;
        float2 * qp = reinterpret_cast<float2 *>(a.q + ((size_t) t * a.n_head_q + h) * D + 2 * lane);
        float2   x  = *qp;
        if ((a.rope.mode & 2) == 0) {
            if (2 * lane < a.rope.n_dims) *qp = rope_row<D>(x, lane, a.pos[t], a.rope, 1.0f);
        } else {
            *qp = rope_row<D>(x, lane, a.pos[t], a.rope, 1.0f);
        }
        return;
    }
    if (h >= a.n_head_kv) return;
    const bool is_v = blockIdx.z == 1;
    const float * cur = is_v ? a.v_cur : a.k_cur;
    if (!cur) return;
    const int blk = a.slot_block[t];
    if (blk < 0) return;
    const int off = a.slot_off[t];
    float2 x = make_float2(0.0f, 0.0f);
    if (2 * lane < D) x = *reinterpret_cast<const float2 *>(cur + ((size_t) t * a.n_head_kv + h) * D + 2 * lane);
    if (!is_v) {
        if (a.has_rope) x = rope_row<D>(x, lane, a.pos[t], a.rope, 1.0f);
        char * tile = a.k_pool + ((size_t) blk * a.n_head_kv + h) * tile_bytes(KT, a.block_size, D);
        tile_write<KT, false>(tile, off, lane, quantize<KT>(x, lane), a.block_size, D);
    } else {
        char * tile = a.v_pool + ((size_t) blk * a.n_head_kv + h) * tile_bytes(VT, a.block_size, D);
        tile_write<VT, true>(tile, off, lane, quantize<VT>(x, lane), a.block_size, D);
    }
}

// grid (n_head_kv, n_cells): dequantize -> rope(delta) -> requantize in place (/ref/src/llama-context.cpp:447-496)
template <int KT, int D>
__global__ void __launch_bounds__(64) k_shift_kernel(char * k_pool, const int32_t * cell_block, const int32_t * cell_off,
                                                     const int32_t * cell_delta, int n_head_kv, int block_size,
                                                     const RopeDev rope) {
    const int h = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const int delta = cell_delta[c];
    if (delta == 0 || cell_block[c] < 0) return;
    char * tile = k_pool + ((size_t) cell_block[c] * n_head_kv + h) * tile_bytes(KT, block_size, D);
    float2 x = dequantize<KT>(tile_read<KT, false>(tile, cell_off[c], lane, block_size, D));
    x = rope_row<D>(x, lane, delta, rope, 1.0f);
    tile_write<KT, false>(tile, cell_off[c], lane, quantize<KT>(x, lane), block_size, D);
}

// grid (n_head_kv, n_tok): reference-format rows <-> tiles
template <int T, bool IS_V, bool SCATTER, int D>
__global__ void __launch_bounds__(64) rows_io_kernel(char * pool, char * rows, const int32_t * slot_block,
                                                     const int32_t * slot_off, int n_head_kv, int block_size,
                                                     int64_t rs_tok, int64_t rs_head) {
    const int h = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    // slot_block == NULL: identity mapping, token t lives in block t / block_size at offset t % block_size
    const int blk = slot_block ? slot_block[t] : t / block_size, off = slot_block ? slot_off[t] : t % block_size;
    if (blk < 0) return;
    char * tile = pool + ((size_t) blk * n_head_kv + h) * tile_bytes(T, block_size, D);
    char * row  = rows + (size_t) t * rs_tok + (size_t) h * rs_head;
    if (SCATTER) tile_write<T, IS_V>(tile, off, lane, ref_read<T>(row, lane, D), block_size, D);
    else         ref_write<T>(row, lane, tile_read<T, IS_V>(tile, off, lane, block_size, D), D);
}

// grid (n_head_kv, n_cells): move single cells between blocks (seq_cp of unaligned ranges, defrag)
template <int T, bool IS_V, int D>
__global__ void __launch_bounds__(64) copy_cells_kernel(char * pool, const int32_t * sb, const int32_t * so,
                                                        const int32_t * db, const int32_t * dof, int n_head_kv,
                                                        int block_size) {
    const int h = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    if (sb[c] < 0 || db[c] < 0) return;
    const size_t tb = tile_bytes(T, block_size, D);
    const Q2 v = tile_read<T, IS_V>(pool + ((size_t) sb[c] * n_head_kv + h) * tb, so[c], lane, block_size, D);
    tile_write<T, IS_V>(pool + ((size_t) db[c] * n_head_kv + h) * tb, dof[c], lane, v, block_size, D);
}

__global__ void __launch_bounds__(256) copy_blocks_kernel(char * pool, size_t block_bytes, const int32_t * src,
                                                          const int32_t * dst) {
    const int i = blockIdx.y;
    if (src[i] < 0 || dst[i] < 0 || src[i] == dst[i]) return;
    const uint4 * s = reinterpret_cast<const uint4 *>(pool + (size_t) src[i] * block_bytes);
    uint4 *       d = reinterpret_cast<uint4 *>(pool + (size_t) dst[i] * block_bytes);
    const size_t  n = block_bytes / 16;
    for (size_t j = (size_t) blockIdx.x * 256 + threadIdx.x; j < n; j += (size_t) gridDim.x * 256) d[j] = s[j];
}

}  // namespace

// ---- host launchers -------------------------------------------------------------------------------------------
int make_rope(const pkv_rope_params * p, int head_dim, RopeDev * out) {
    PKV_REQUIRE(p != nullptr, "rope params missing");
    PKV_REQUIRE(p->n_dims > 0 && p->n_dims % 2 == 0 && p->n_dims <= head_dim, "rope: bad n_dims %d", p->n_dims);
    if (p->mode != 0 && p->mode != 2) {
        set_error("rope: mode %d not supported (NORM = 0, NEOX = 2)", p->mode);
        return PKV_ERR_UNSUPPORTED;
    }
    out->n_dims = p->n_dims; out->mode = p->mode;
    out->freq_scale = p->freq_scale; out->ext_factor = p->ext_factor; out->attn_factor = p->attn_factor;
    out->theta_scale = powf(p->freq_base, -2.0f / p->n_dims);
    // ggml_rope_yarn_corr_dims (/ref/ggml/src/ggml.c:3735-3747)
    auto corr = [&](float n_rot) { return p->n_dims * logf(p->n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(p->freq_base)); };
    out->corr0 = fmaxf(0.0f, floorf(corr(p->beta_fast)));
    out->corr1 = fminf((float) (p->n_dims - 1), ceilf(corr(p->beta_slow)));
    out->yarn_mscale = 1.0f + 0.1f * logf(1.0f / p->freq_scale);   // rope_yarn's mscale factor (:10590), by the host's logf like the reference's
    out->freq_factors = p->freq_factors;
    return PKV_OK;
}

int launch_rope_f32(const float * x, const int32_t * pos, float * out, int n_tok, int n_head, int D, const RopeDev & r,
                    int inverse, hipStream_t st, int64_t xs_tok, int64_t xs_head) {
    if (n_tok == 0) return PKV_OK;
    if (xs_tok == 0) { xs_tok = (int64_t) n_head * D; xs_head = D; }
    hipLaunchKernelGGL(rope_kernel<float>, dim3(n_head, n_tok), dim3(64), 0, st, x, pos, out, n_head, D, r, inverse ? -1.0f : 1.0f, xs_tok, xs_head);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_rope_cache(const int32_t * pos, float * cache, int n_tok, int D, const RopeDev & r, hipStream_t st) {
    if (n_tok == 0) return PKV_OK;
    hipLaunchKernelGGL(rope_cache_kernel, dim3(n_tok), dim3(64), 0, st, pos, reinterpret_cast<float2 *>(cache), D / 2, r);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_rope_f16(const void * x, const int32_t * pos, void * out, int n_tok, int n_head, int D, const RopeDev & r,
                    int inverse, hipStream_t st, int64_t xs_tok, int64_t xs_head) {
    if (n_tok == 0) return PKV_OK;
    if (xs_tok == 0) { xs_tok = (int64_t) n_head * D; xs_head = D; }
    hipLaunchKernelGGL(rope_kernel<_Float16>, dim3(n_head, n_tok), dim3(64), 0, st, (const _Float16 *) x, pos, (_Float16 *) out, n_head,
                       D, r, inverse ? -1.0f : 1.0f, xs_tok, xs_head);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

// The same work for a prefill ubatch: ONE workgroup per token (4 waves), every wave walking rows q heads | k heads | v heads with
// stride 4.  kv_store_kernel spends a one-wave workgroup and a libm-exact (cos, sin) evaluation on every 512-byte row: a 2048-token
// ubatch of the 8B geometry is 98,304 rows — 101 us per layer, four times the prefill attention kernel behind it.  A token's angles
// do not depend on the head: here each wave evaluates them once and rotates its 10 roped rows with them (same rope_cs, same rot0 /
// rot1: the stored bits are those of kv_store_kernel), and its rows' loads are issued four ahead.
template <int D>
__device__ __forceinline__ float2 rope_apply(float2 x, int lane, float2 c0, float2 c1, const RopeDev & r) {
    const int l = lane % (D / 2);
    if ((r.mode & 2) == 0) return 2 * l < r.n_dims ? make_float2(rot0(x.x, x.y, c0.x, c0.y), rot1(x.x, x.y, c0.x, c0.y)) : x;
    constexpr int HL = D / 4;
    const float px = __shfl_xor(x.x, HL), py = __shfl_xor(x.y, HL);
    return l < HL ? make_float2(rot0(x.x, px, c0.x, c0.y), rot0(x.y, py, c1.x, c1.y))
                  : make_float2(rot1(px, x.x, c0.x, c0.y), rot1(py, x.y, c1.x, c1.y));
}
template <int KT, int VT, int D>
__global__ void __launch_bounds__(256) kv_store_tok_kernel(const StoreArgs a) {
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = a.q ? a.n_head_q : 0, nk = a.k_cur ? a.n_head_kv : 0, nv = a.v_cur ? a.n_head_kv : 0;
    const int blk = a.slot_block[t], off = a.slot_off[t];
    float2 c0 = make_float2(1.0f, 0.0f), c1 = c0;
    if (a.has_rope) {
        const int pos = a.pos[t], e0 = 2 * (lane % (D / 2));
        if ((a.rope.mode & 2) == 0) {
            if (a.rope_cache)             c0 = reinterpret_cast<const float2 *>(a.rope_cache)[(size_t) t * (D / 2) + e0 / 2];   // the same values, computed once per ubatch
            else if (e0 < a.rope.n_dims)  c0 = rope_cs(pos, e0 / 2, a.rope, 1.0f);
        } else {
            const int ic = (lane % (D / 2)) < D / 4 ? e0 : e0 - D / 2;
            c0 = rope_cs(pos, ic, a.rope, 1.0f);
            c1 = rope_cs(pos, ic + 1, a.rope, 1.0f);
        }
    }
    const bool act = 2 * lane < D;                     // D = 64: the upper half-wave idles
    const int n_rows = nq + nk + nv;
    auto src_of = [&](int row) -> const float * {
        if (row < nq)      return a.q + ((size_t) t * a.n_head_q + row) * D;
        if (row < nq + nk) return a.k_cur + ((size_t) t * a.n_head_kv + (row - nq)) * D;
        return a.v_cur + ((size_t) t * a.n_head_kv + (row - nq - nk)) * D;
    };
    for (int r0 = wave; r0 < n_rows; r0 += 16) {
        float2 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 4 * j;
            x[j] = make_float2(0.0f, 0.0f);
            if (row < n_rows && act) x[j] = *reinterpret_cast<const float2 *>(src_of(row) + 2 * lane);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 4 * j;
            if (row >= n_rows) break;                  // wave-uniform
            if (row < nq) {
                const float2 y = a.has_rope ? rope_apply<D>(x[j], lane, c0, c1, a.rope) : x[j];
                if (act) *reinterpret_cast<float2 *>(a.q + ((size_t) t * a.n_head_q + row) * D + 2 * lane) = y;
            } else if (blk >= 0) {
                if (row < nq + nk) {
                    const int h = row - nq;
                    const float2 y = a.has_rope ? rope_apply<D>(x[j], lane, c0, c1, a.rope) : x[j];
                    tile_write<KT, false>(a.k_pool + ((size_t) blk * a.n_head_kv + h) * tile_bytes(KT, a.block_size, D), off, lane, quantize<KT>(y, lane), a.block_size, D);
                } else {
                    const int h = row - nq - nk;
                    tile_write<VT, true>(a.v_pool + ((size_t) blk * a.n_head_kv + h) * tile_bytes(VT, a.block_size, D), off, lane, quantize<VT>(x[j], lane), a.block_size, D);
                }
            }
        }
    }
}
template <int KT, int VT>
static int store_d(const StoreArgs & a, int n_tok, int D, hipStream_t st) {
    if (n_tok >= 64 && !getenv("PKV_KV_STORE_ROWS")) {   // a prefill ubatch: one workgroup per token (kv_store_tok_kernel)
        if (D == 128)      hipLaunchKernelGGL((kv_store_tok_kernel<KT, VT, 128>), dim3(n_tok), dim3(256), 0, st, a);
        else if (D == 64)  hipLaunchKernelGGL((kv_store_tok_kernel<KT, VT, 64>), dim3(n_tok), dim3(256), 0, st, a);
        else { set_error("pkv_kv_store: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
        PKV_HIP(hipGetLastError());
        return PKV_OK;
    }
    dim3 grid(a.q ? (a.n_head_q > a.n_head_kv ? a.n_head_q : a.n_head_kv) : a.n_head_kv, n_tok, a.q ? 3 : 2);
    if (D == 128)      hipLaunchKernelGGL((kv_store_kernel<KT, VT, 128>), grid, dim3(64), 0, st, a);
    else if (D == 64)  hipLaunchKernelGGL((kv_store_kernel<KT, VT, 64>), grid, dim3(64), 0, st, a);
    else { set_error("pkv_kv_store: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
template <int KT>
static int store_v(const StoreArgs & a, int type_v, int n_tok, int D, hipStream_t st) {
    switch (type_v) {
        case PKV_TYPE_F16:  return store_d<KT, PKV_TYPE_F16>(a, n_tok, D, st);
        case PKV_TYPE_Q8_0: return store_d<KT, PKV_TYPE_Q8_0>(a, n_tok, D, st);
        case PKV_TYPE_Q4_0: return store_d<KT, PKV_TYPE_Q4_0>(a, n_tok, D, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

int launch_kv_store(float * q, int n_head_q, const float * k_cur, const float * v_cur, void * k_pool, void * v_pool,
                    int type_k, int type_v, const int32_t * slot_block, const int32_t * slot_off, const int32_t * pos,
                    const RopeDev * rope, int n_tok, int n_head_kv, int D, int block_size, hipStream_t st, const float * rope_cache) {
    if (n_tok == 0) return PKV_OK;
    StoreArgs a{};
    a.rope_cache = rope_cache;
    a.q = q; a.n_head_q = n_head_q;
    a.k_cur = k_cur; a.v_cur = v_cur; a.k_pool = (char *) k_pool; a.v_pool = (char *) v_pool;
    a.slot_block = slot_block; a.slot_off = slot_off; a.pos = pos;
    a.n_head_kv = n_head_kv; a.block_size = block_size; a.has_rope = rope != nullptr;
    if (rope) {
        a.rope = *rope;
        if ((rope->mode & 2) && rope->n_dims != D) { set_error("fused NEOX rope needs n_dims == head_dim"); return PKV_ERR_UNSUPPORTED; }
    }
    switch (type_k) {
        case PKV_TYPE_F16:  return store_v<PKV_TYPE_F16>(a, type_v, n_tok, D, st);
        case PKV_TYPE_Q8_0: return store_v<PKV_TYPE_Q8_0>(a, type_v, n_tok, D, st);
        case PKV_TYPE_Q4_0: return store_v<PKV_TYPE_Q4_0>(a, type_v, n_tok, D, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

template <int KT>
static int shift_d(void * k_pool, const int32_t * cb, const int32_t * co, const int32_t * cd, int n_cells, int n_head_kv,
                   int D, int block_size, const RopeDev & r, hipStream_t st) {
    dim3 grid(n_head_kv, n_cells);
    if (D == 128)     hipLaunchKernelGGL((k_shift_kernel<KT, 128>), grid, dim3(64), 0, st, (char *) k_pool, cb, co, cd, n_head_kv, block_size, r);
    else if (D == 64) hipLaunchKernelGGL((k_shift_kernel<KT, 64>), grid, dim3(64), 0, st, (char *) k_pool, cb, co, cd, n_head_kv, block_size, r);
    else { set_error("pkv_k_shift: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_k_shift(void * k_pool, int type_k, const int32_t * cell_block, const int32_t * cell_off,
                   const int32_t * cell_delta, int n_cells, int n_head_kv, int D, int block_size, const RopeDev & r,
                   hipStream_t st) {
    if (n_cells == 0) return PKV_OK;
    if ((r.mode & 2) && r.n_dims != D) { set_error("K-shift NEOX rope needs n_dims == head_dim"); return PKV_ERR_UNSUPPORTED; }
    switch (type_k) {
        case PKV_TYPE_F16:  return shift_d<PKV_TYPE_F16>(k_pool, cell_block, cell_off, cell_delta, n_cells, n_head_kv, D, block_size, r, st);
        case PKV_TYPE_Q8_0: return shift_d<PKV_TYPE_Q8_0>(k_pool, cell_block, cell_off, cell_delta, n_cells, n_head_kv, D, block_size, r, st);
        case PKV_TYPE_Q4_0: return shift_d<PKV_TYPE_Q4_0>(k_pool, cell_block, cell_off, cell_delta, n_cells, n_head_kv, D, block_size, r, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

int launch_copy_blocks(void * pool, size_t block_bytes, const int32_t * src, const int32_t * dst, int n, hipStream_t st) {
    if (n == 0) return PKV_OK;
    const int gx = (int) ((block_bytes / 16 + 255) / 256);
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(gx < 64 ? gx : 64, n), dim3(256), 0, st, (char *) pool, block_bytes, src, dst);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

template <int T, bool IS_V>
static int cells_d(void * pool, const int32_t * sb, const int32_t * so, const int32_t * db, const int32_t * dof, int n,
                   int n_head_kv, int D, int block_size, hipStream_t st) {
    dim3 grid(n_head_kv, n);
    if (D == 128)     hipLaunchKernelGGL((copy_cells_kernel<T, IS_V, 128>), grid, dim3(64), 0, st, (char *) pool, sb, so, db, dof, n_head_kv, block_size);
    else if (D == 64) hipLaunchKernelGGL((copy_cells_kernel<T, IS_V, 64>), grid, dim3(64), 0, st, (char *) pool, sb, so, db, dof, n_head_kv, block_size);
    else { set_error("cell copy: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_copy_cells(void * pool, int type, int is_v, const int32_t * sb, const int32_t * so, const int32_t * db,
                      const int32_t * dof, int n, int n_head_kv, int D, int block_size, hipStream_t st) {
    if (n == 0) return PKV_OK;
    switch (type) {
        case PKV_TYPE_F16:  return is_v ? cells_d<PKV_TYPE_F16, true>(pool, sb, so, db, dof, n, n_head_kv, D, block_size, st)
                                        : cells_d<PKV_TYPE_F16, false>(pool, sb, so, db, dof, n, n_head_kv, D, block_size, st);
        case PKV_TYPE_Q8_0: return is_v ? cells_d<PKV_TYPE_Q8_0, true>(pool, sb, so, db, dof, n, n_head_kv, D, block_size, st)
                                        : cells_d<PKV_TYPE_Q8_0, false>(pool, sb, so, db, dof, n, n_head_kv, D, block_size, st);
        case PKV_TYPE_Q4_0: return is_v ? cells_d<PKV_TYPE_Q4_0, true>(pool, sb, so, db, dof, n, n_head_kv, D, block_size, st)
                                        : cells_d<PKV_TYPE_Q4_0, false>(pool, sb, so, db, dof, n, n_head_kv, D, block_size, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

template <int T, bool IS_V, bool SC>
static int rows_d(void * pool, void * rows, const int32_t * sb, const int32_t * so, int n_tok, int n_head_kv, int D,
                  int block_size, hipStream_t st, int64_t rs_tok, int64_t rs_head) {
    dim3 grid(n_head_kv, n_tok);
    if (rs_tok == 0) { rs_head = (int64_t) row_size(T, D); rs_tok = rs_head * n_head_kv; }
    if (D == 128)     hipLaunchKernelGGL((rows_io_kernel<T, IS_V, SC, 128>), grid, dim3(64), 0, st, (char *) pool, (char *) rows, sb, so, n_head_kv, block_size, rs_tok, rs_head);
    else if (D == 64) hipLaunchKernelGGL((rows_io_kernel<T, IS_V, SC, 64>), grid, dim3(64), 0, st, (char *) pool, (char *) rows, sb, so, n_head_kv, block_size, rs_tok, rs_head);
    else { set_error("row I/O: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
template <int T>
static int rows_t(void * pool, void * rows, const int32_t * sb, const int32_t * so, int n_tok, int n_head_kv, int D,
                  int block_size, bool is_v, bool scatter, hipStream_t st, int64_t rs_tok, int64_t rs_head) {
    if (is_v) return scatter ? rows_d<T, true, true>(pool, rows, sb, so, n_tok, n_head_kv, D, block_size, st, rs_tok, rs_head)
                             : rows_d<T, true, false>(pool, rows, sb, so, n_tok, n_head_kv, D, block_size, st, rs_tok, rs_head);
    return scatter ? rows_d<T, false, true>(pool, rows, sb, so, n_tok, n_head_kv, D, block_size, st, rs_tok, rs_head)
                   : rows_d<T, false, false>(pool, rows, sb, so, n_tok, n_head_kv, D, block_size, st, rs_tok, rs_head);
}

int launch_rows_io(void * pool, int type, int is_v, void * rows, const int32_t * slot_block, const int32_t * slot_off,
                   int n_tok, int n_head_kv, int D, int block_size, bool scatter, hipStream_t st, int64_t rs_tok, int64_t rs_head) {
    if (n_tok == 0) return PKV_OK;
    switch (type) {
        case PKV_TYPE_F16:  return rows_t<PKV_TYPE_F16>(pool, rows, slot_block, slot_off, n_tok, n_head_kv, D, block_size, is_v, scatter, st, rs_tok, rs_head);
        case PKV_TYPE_Q8_0: return rows_t<PKV_TYPE_Q8_0>(pool, rows, slot_block, slot_off, n_tok, n_head_kv, D, block_size, is_v, scatter, st, rs_tok, rs_head);
        case PKV_TYPE_Q4_0: return rows_t<PKV_TYPE_Q4_0>(pool, rows, slot_block, slot_off, n_tok, n_head_kv, D, block_size, is_v, scatter, st, rs_tok, rs_head);
    }
    return PKV_ERR_UNSUPPORTED;
}

// f32 rows -> reference-format bytes (what GGML_OP_CPY f32 -> f16/q8_0/q4_0 stores, /ref/ggml/src/ggml-cpu/ggml-cpu.c:3742):
// grid (ne0 / CH, n_rows), one wave per CH-element chunk of a row
template <int T, int CH>
__global__ void __launch_bounds__(64) convert_rows_kernel(const float * x, char * dst, int64_t ne0) {
    const int64_t row = blockIdx.y;
    const int ch = blockIdx.x, lane = threadIdx.x;
    float2 v = make_float2(0.0f, 0.0f);
    if (2 * lane < CH) v = *reinterpret_cast<const float2 *>(x + row * ne0 + (int64_t) ch * CH + 2 * lane);
    ref_write<T>(dst + row * (int64_t) row_size(T, ne0) + (int64_t) ch * row_size(T, CH), lane, quantize<T>(v, lane), CH);
}

// the same conversion for KV rows of head_dim D that also land in a paged pool (identity placement: cell c = block c / BS, offset
// c % BS): flat[g] and the pool row of (cell0 + g / n_head_kv, head g % n_head_kv) receive the SAME quantized bits, so a flat cache
// tensor and its paged mirror are written by one launch.  grid = number of head rows.
template <int T, bool IS_V, int D>
__global__ void __launch_bounds__(64) convert_rows_mirror_kernel(const float * x, char * dst, char * pool, int64_t cell0, int n_head_kv, int BS) {
    const int64_t g = blockIdx.x;
    const int lane = threadIdx.x;
    float2 v = make_float2(0.0f, 0.0f);
    if (2 * lane < D) v = *reinterpret_cast<const float2 *>(x + g * D + 2 * lane);
    const Q2 q = quantize<T>(v, lane);
    if (dst) ref_write<T>(dst + g * (int64_t) row_size(T, D), lane, q, D);   // dst == NULL: the paged rows are the only copy
    const int64_t cell = cell0 + g / n_head_kv;
    const int h = (int) (g % n_head_kv);
    char * tile = pool + ((size_t) (cell / BS) * n_head_kv + h) * tile_bytes(T, BS, D);
    tile_write<T, IS_V>(tile, (int) (cell % BS), lane, q, BS, D);
}
// ... with the RoPE of the K rows in front (ROPE -> CPY of build_attn, /ref/src/llama-graph.cpp:1375-1397): x is the un-rotated k_cur
// (strided [tok][head][D]), `roped` (optional) receives what GGML_OP_ROPE would have written, flat + pool what the CPY of it stores
template <int T, int D>
__global__ void __launch_bounds__(64) rope_convert_mirror_kernel(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, float * roped, char * dst,
                                                                 char * pool, int64_t cell0, int n_head_kv, int BS, const RopeDev r) {
    const int64_t g = blockIdx.x, t = g / n_head_kv;
    const int h = (int) (g % n_head_kv), lane = threadIdx.x;
    float2 v = make_float2(0.0f, 0.0f);
    if (2 * lane < D) v = *reinterpret_cast<const float2 *>(x + t * xs_tok + h * xs_head + 2 * lane);
    v = rope_row<D>(v, lane, pos[t], r, 1.0f);
    if (roped && 2 * lane < D) *reinterpret_cast<float2 *>(roped + g * D + 2 * lane) = v;
    const Q2 q = quantize<T>(v, lane);
    if (dst) ref_write<T>(dst + g * (int64_t) row_size(T, D), lane, q, D);   // dst == NULL: the paged rows are the only copy
    const int64_t cell = cell0 + t;
    char * tile = pool + ((size_t) (cell / BS) * n_head_kv + h) * tile_bytes(T, BS, D);
    tile_write<T, false>(tile, (int) (cell % BS), lane, q, BS, D);
}
// the three per-token ops between the q / k / v products and the attention of a llama layer in ONE launch: RoPE(Q) out of place,
// RoPE(K) + store (flat + pool), store V (flat + pool).  grid (max(n_head_q, n_head_kv), n_tok, 3): z = 0 K, 1 V, 2 Q.
struct QkvStoreArgs {
    const float * q; int64_t q_s_tok, q_s_head; float * q_out; int n_head_q;
    const float * k; int64_t k_s_tok, k_s_head; char * k_dst; char * k_pool; int64_t k_cell0;
    const float * v; char * v_dst; char * v_pool; int64_t v_cell0;
    const int32_t * pos; RopeDev rope; int n_head_kv, BS;
};
template <int KT, int VT, int D>
__global__ void __launch_bounds__(64) qkv_store_mirror_kernel(const QkvStoreArgs a) {
    const int h = blockIdx.x, lane = threadIdx.x;
    const int64_t t = blockIdx.y;
    if (blockIdx.z == 2) {
        if (h >= a.n_head_q) return;
        float2 x = make_float2(0.0f, 0.0f);
        if (2 * lane < D) x = *reinterpret_cast<const float2 *>(a.q + t * a.q_s_tok + h * a.q_s_head + 2 * lane);
        x = rope_row<D>(x, lane, a.pos[t], a.rope, 1.0f);
        if (2 * lane < D) *reinterpret_cast<float2 *>(a.q_out + (t * a.n_head_q + h) * D + 2 * lane) = x;
        return;
    }
    if (h >= a.n_head_kv) return;
    const int64_t g = t * a.n_head_kv + h;
    if (blockIdx.z == 0) {
        float2 x = make_float2(0.0f, 0.0f);
        if (2 * lane < D) x = *reinterpret_cast<const float2 *>(a.k + t * a.k_s_tok + h * a.k_s_head + 2 * lane);
        x = rope_row<D>(x, lane, a.pos[t], a.rope, 1.0f);
        const Q2 q = quantize<KT>(x, lane);
        if (a.k_dst) ref_write<KT>(a.k_dst + g * (int64_t) row_size(KT, D), lane, q, D);
        const int64_t cell = a.k_cell0 + t;
        tile_write<KT, false>(a.k_pool + ((size_t) (cell / a.BS) * a.n_head_kv + h) * tile_bytes(KT, a.BS, D), (int) (cell % a.BS), lane, q, a.BS, D);
    } else {
        float2 x = make_float2(0.0f, 0.0f);
        if (2 * lane < D) x = *reinterpret_cast<const float2 *>(a.v + g * D + 2 * lane);
        const Q2 q = quantize<VT>(x, lane);
        if (a.v_dst) ref_write<VT>(a.v_dst + g * (int64_t) row_size(VT, D), lane, q, D);
        const int64_t cell = a.v_cell0 + t;
        tile_write<VT, true>(a.v_pool + ((size_t) (cell / a.BS) * a.n_head_kv + h) * tile_bytes(VT, a.BS, D), (int) (cell % a.BS), lane, q, a.BS, D);
    }
}
// the prompt-processing form (one workgroup per token, the token's angles evaluated once per wave: see kv_store_tok_kernel)
template <int KT, int VT, int D>
__global__ void __launch_bounds__(256) qkv_store_mirror_tok_kernel(const QkvStoreArgs a) {
    const int64_t t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = a.n_head_q, nk = a.n_head_kv, n_rows = nq + 2 * nk;
    float2 c0, c1;
    {
        const int pos = a.pos[t], l = lane % (D / 2), e0 = 2 * l;
        c0 = c1 = make_float2(1.0f, 0.0f);
        if ((a.rope.mode & 2) == 0) {
            if (e0 < a.rope.n_dims) c0 = rope_cs(pos, e0 / 2, a.rope, 1.0f);
        } else {
            const int ic = l < D / 4 ? e0 : e0 - D / 2;
            c0 = rope_cs(pos, ic, a.rope, 1.0f);
            c1 = rope_cs(pos, ic + 1, a.rope, 1.0f);
        }
    }
    const bool act = 2 * lane < D;
    auto src_of = [&](int row) -> const float * {
        if (row < nq)      return a.q + t * a.q_s_tok + (int64_t) row * a.q_s_head;
        if (row < nq + nk) return a.k + t * a.k_s_tok + (int64_t) (row - nq) * a.k_s_head;
        return a.v + (t * a.n_head_kv + (row - nq - nk)) * D;
    };
    for (int r0 = wave; r0 < n_rows; r0 += 16) {
        float2 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 4 * j;
            x[j] = make_float2(0.0f, 0.0f);
            if (row < n_rows && act) x[j] = *reinterpret_cast<const float2 *>(src_of(row) + 2 * lane);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 4 * j;
            if (row >= n_rows) break;                  // wave-uniform
            if (row < nq) {
                const float2 y = rope_apply<D>(x[j], lane, c0, c1, a.rope);
                if (act) *reinterpret_cast<float2 *>(a.q_out + (t * a.n_head_q + row) * D + 2 * lane) = y;
            } else if (row < nq + nk) {
                const int h = row - nq;
                const Q2 q = quantize<KT>(rope_apply<D>(x[j], lane, c0, c1, a.rope), lane);
                if (a.k_dst) ref_write<KT>(a.k_dst + (t * a.n_head_kv + h) * (int64_t) row_size(KT, D), lane, q, D);
                const int64_t cell = a.k_cell0 + t;
                tile_write<KT, false>(a.k_pool + ((size_t) (cell / a.BS) * a.n_head_kv + h) * tile_bytes(KT, a.BS, D), (int) (cell % a.BS), lane, q, a.BS, D);
            } else {
                const int h = row - nq - nk;
                const Q2 q = quantize<VT>(x[j], lane);
                if (a.v_dst) ref_write<VT>(a.v_dst + (t * a.n_head_kv + h) * (int64_t) row_size(VT, D), lane, q, D);
                const int64_t cell = a.v_cell0 + t;
                tile_write<VT, true>(a.v_pool + ((size_t) (cell / a.BS) * a.n_head_kv + h) * tile_bytes(VT, a.BS, D), (int) (cell % a.BS), lane, q, a.BS, D);
            }
        }
    }
}
template <int KT, int VT>
static int qkv_store_d(const QkvStoreArgs & a, int n_tok, int D, hipStream_t st) {
    if (n_tok >= 64 && !getenv("PKV_KV_STORE_ROWS")) {
        if (D == 128)     hipLaunchKernelGGL((qkv_store_mirror_tok_kernel<KT, VT, 128>), dim3(n_tok), dim3(256), 0, st, a);
        else if (D == 64) hipLaunchKernelGGL((qkv_store_mirror_tok_kernel<KT, VT, 64>), dim3(n_tok), dim3(256), 0, st, a);
        else { set_error("qkv store: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
        PKV_HIP(hipGetLastError());
        return PKV_OK;
    }
    const dim3 grid(a.n_head_q > a.n_head_kv ? a.n_head_q : a.n_head_kv, n_tok, 3);
    if (D == 128)     hipLaunchKernelGGL((qkv_store_mirror_kernel<KT, VT, 128>), grid, dim3(64), 0, st, a);
    else if (D == 64) hipLaunchKernelGGL((qkv_store_mirror_kernel<KT, VT, 64>), grid, dim3(64), 0, st, a);
    else { set_error("qkv store: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
template <int KT>
static int qkv_store_v(const QkvStoreArgs & a, int type_v, int n_tok, int D, hipStream_t st) {
    switch (type_v) {
        case PKV_TYPE_F16:  return qkv_store_d<KT, PKV_TYPE_F16>(a, n_tok, D, st);
        case PKV_TYPE_Q8_0: return qkv_store_d<KT, PKV_TYPE_Q8_0>(a, n_tok, D, st);
        case PKV_TYPE_Q4_0: return qkv_store_d<KT, PKV_TYPE_Q4_0>(a, n_tok, D, st);
    }
    return PKV_ERR_UNSUPPORTED;
}
int launch_qkv_store_mirror(const pkv_qkv_store_args * p, const RopeDev & r, hipStream_t st) {
    if (p->n_tok == 0) return PKV_OK;
    QkvStoreArgs a;
    a.q = p->q; a.q_s_tok = p->q_stride_tok ? p->q_stride_tok : (int64_t) p->n_head * p->head_dim; a.q_s_head = p->q_stride_tok ? p->q_stride_head : p->head_dim;
    a.q_out = p->q_out; a.n_head_q = p->n_head;
    a.k = p->k; a.k_s_tok = p->k_stride_tok ? p->k_stride_tok : (int64_t) p->n_head_kv * p->head_dim; a.k_s_head = p->k_stride_tok ? p->k_stride_head : p->head_dim;
    a.k_dst = (char *) p->k_dst; a.k_pool = (char *) p->k_pool; a.k_cell0 = p->k_cell0;
    a.v = p->v; a.v_dst = (char *) p->v_dst; a.v_pool = (char *) p->v_pool; a.v_cell0 = p->v_cell0;
    a.pos = p->pos; a.rope = r; a.n_head_kv = p->n_head_kv; a.BS = p->block_size;
    switch (p->type_k) {
        case PKV_TYPE_F16:  return qkv_store_v<PKV_TYPE_F16>(a, p->type_v, p->n_tok, p->head_dim, st);
        case PKV_TYPE_Q8_0: return qkv_store_v<PKV_TYPE_Q8_0>(a, p->type_v, p->n_tok, p->head_dim, st);
        case PKV_TYPE_Q4_0: return qkv_store_v<PKV_TYPE_Q4_0>(a, p->type_v, p->n_tok, p->head_dim, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

template <int T>
static int rope_convert_mirror_t(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, float * roped, void * dst, void * pool, int64_t cell0,
                                 int64_t n_tok, int n_head_kv, int D, int BS, const RopeDev & r, hipStream_t st) {
    const dim3 grid((unsigned) (n_tok * n_head_kv));
    if (D == 128)     hipLaunchKernelGGL((rope_convert_mirror_kernel<T, 128>), grid, dim3(64), 0, st, x, xs_tok, xs_head, pos, roped, (char *) dst, (char *) pool, cell0, n_head_kv, BS, r);
    else if (D == 64) hipLaunchKernelGGL((rope_convert_mirror_kernel<T, 64>), grid, dim3(64), 0, st, x, xs_tok, xs_head, pos, roped, (char *) dst, (char *) pool, cell0, n_head_kv, BS, r);
    else { set_error("rope + convert + mirror: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_rope_convert_mirror(const float * x, int64_t xs_tok, int64_t xs_head, const int32_t * pos, float * roped, void * dst, int type, void * pool,
                               int64_t cell0, int64_t n_tok, int n_head_kv, int D, int BS, const RopeDev & r, hipStream_t st) {
    if (n_tok == 0) return PKV_OK;
    if (xs_tok == 0) { xs_tok = (int64_t) n_head_kv * D; xs_head = D; }
    switch (type) {
        case PKV_TYPE_F16:  return rope_convert_mirror_t<PKV_TYPE_F16>(x, xs_tok, xs_head, pos, roped, dst, pool, cell0, n_tok, n_head_kv, D, BS, r, st);
        case PKV_TYPE_Q8_0: return rope_convert_mirror_t<PKV_TYPE_Q8_0>(x, xs_tok, xs_head, pos, roped, dst, pool, cell0, n_tok, n_head_kv, D, BS, r, st);
        case PKV_TYPE_Q4_0: return rope_convert_mirror_t<PKV_TYPE_Q4_0>(x, xs_tok, xs_head, pos, roped, dst, pool, cell0, n_tok, n_head_kv, D, BS, r, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

template <int T>
static int convert_mirror_t(const float * x, void * dst, void * pool, int is_v, int64_t cell0, int64_t n_head_rows, int n_head_kv, int D, int BS, hipStream_t st) {
    const dim3 grid((unsigned) n_head_rows);
#define PKV_CM(V, DD) hipLaunchKernelGGL((convert_rows_mirror_kernel<T, V, DD>), grid, dim3(64), 0, st, x, (char *) dst, (char *) pool, cell0, n_head_kv, BS)
    if (D == 128) { if (is_v) PKV_CM(true, 128); else PKV_CM(false, 128); }
    else if (D == 64) { if (is_v) PKV_CM(true, 64); else PKV_CM(false, 64); }
    else { set_error("convert + mirror: unsupported head_dim %d", D); return PKV_ERR_UNSUPPORTED; }
#undef PKV_CM
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_convert_rows_mirror(const float * x, void * dst, int type, void * pool, int is_v, int64_t cell0, int64_t n_head_rows, int n_head_kv, int D, int BS,
                               hipStream_t st) {
    if (n_head_rows == 0) return PKV_OK;
    switch (type) {
        case PKV_TYPE_F16:  return convert_mirror_t<PKV_TYPE_F16>(x, dst, pool, is_v, cell0, n_head_rows, n_head_kv, D, BS, st);
        case PKV_TYPE_Q8_0: return convert_mirror_t<PKV_TYPE_Q8_0>(x, dst, pool, is_v, cell0, n_head_rows, n_head_kv, D, BS, st);
        case PKV_TYPE_Q4_0: return convert_mirror_t<PKV_TYPE_Q4_0>(x, dst, pool, is_v, cell0, n_head_rows, n_head_kv, D, BS, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

int launch_convert_rows(const float * x, void * dst, int type, int64_t n_rows, int64_t ne0, hipStream_t st) {
    if (n_rows == 0) return PKV_OK;
    const int CH = ne0 % 128 == 0 ? 128 : 64;
    if (ne0 % 64 != 0) { set_error("convert: row length %lld is not a multiple of 64", (long long) ne0); return PKV_ERR_UNSUPPORTED; }
    dim3 grid((unsigned) (ne0 / CH), (unsigned) n_rows);
#define PKV_CVT(T) do { if (CH == 128) hipLaunchKernelGGL((convert_rows_kernel<T, 128>), grid, dim3(64), 0, st, x, (char *) dst, ne0); \
                        else           hipLaunchKernelGGL((convert_rows_kernel<T, 64>), grid, dim3(64), 0, st, x, (char *) dst, ne0); } while (0)
    switch (type) {
        case PKV_TYPE_F16:  PKV_CVT(PKV_TYPE_F16); break;
        case PKV_TYPE_Q8_0: PKV_CVT(PKV_TYPE_Q8_0); break;
        case PKV_TYPE_Q4_0: PKV_CVT(PKV_TYPE_Q4_0); break;
        default: return PKV_ERR_UNSUPPORTED;
    }
#undef PKV_CVT
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

}  // namespace pkv
