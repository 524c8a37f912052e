// pkv_attn_flat.hip — GGML_OP_FLASH_ATTN_EXT on the reference's OWN tensor layout (gfx950), for what the paged kernels do not take.
//
// The paged decode / prefill kernels (pkv_attn_decode.hip, pkv_attn_prefill.hip) are cut for the north-star geometries: head size 64 / 128, f16 / q8_0 / q4_0
// tiles of a block pool.  The operator they stand in for — ggml_compute_forward_flash_attn_ext_f16 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:12219-12432) — is
// wider: any head size, bf16 K / V (CLI-whitelisted cache type, /ref/common/arg.cpp:172-182), any gqa ratio, K / V as strided views of flat tensors
// (/ref/tests/test-backend-ops.cpp:4367-4390 tests head sizes 64 / 80 / 128 / 256 x f16 / bf16 / q8_0 / q4_0), and the cache types q4_1 / q5_0 / q5_1 / iq4_nl
// the CLI also accepts (their KV store — GGML_OP_CPY f32 -> type — is quantize_blocks_kernel below).  This file is that width: one kernel that reads
// the reference's rows where they lie (block_q8_0 34 B, block_q4_0 18 B, 2-byte aligned; 16-bit rows by 16-byte loads when aligned) — no pool, no
// conversion pass, bf16 stays bf16.
//
// Arithmetic follows the reference's operand roundings: Q goes through the K type's vec_dot_type first (:12343 — f16 for f16 K, bf16 for bf16 K
// (ggml_compute_fp32_to_bf16: nearest even), q8_0 with the AVX2 rounding for q8_0 / q4_0 K: id = 127 / amax, ties to even, d kept as f16); integer
// dots of the quantized types are exact in f32 (|sum| <= 32 * 127 * 127); scale, softcap (tanhf), mask * ALiBi slope as :12349-12365; V rows are
// accumulated in f32 (the reference: f32 for every V type but f16, where its f16 accumulator drifts from exact arithmetic by itself).
//
// Shape: one workgroup of 4 waves (16 for rows of >= 1024 cells) per (query row, head); wave w walks the 64-token chunks w, w + 4, ..: lane = token for QK^T (the Q operand sits in LDS,
// every lane reads the same address: a broadcast), online softmax per wave, then lane = column for PV (consecutive lanes read consecutive elements of one V
// row); one cross-wave merge through LDS.  A coverage kernel, not a roofline one: the hot geometries never come here (pkv_flash_attn_ext_flat's callers
// ask the paged kernels first).
#include "pkv_common.h"
#include <algorithm>

namespace pkv {
namespace {

constexpr int FLAT_NW = 4, FLAT_NW_LONG = 16, FLAT_DMAX = 256, FLAT_MAX_SPLIT = 32;   // waves per workgroup: 4, or 16 once a row has >= 16 chunks of 64 cells to walk

struct FlatArgs {
    const float * q; int64_t q_s_tok, q_s_head;
    const char * k, * v; int64_t k_s_tok, k_s_head, v_s_tok, v_s_head;
    const _Float16 * mask; int64_t mask_stride;
    float * out;
    int n_q, n_kv, n_head, n_head_kv, D;
    float scale, softcap, max_bias, m0, m1;
    unsigned n_head_log2;
    int k_al16, v_al16;
    int n_split, split_len;            // split-KV: workgroup z walks the cells [z * split_len, (z + 1) * split_len) and leaves an (m, l, O) partial
    float * part;                      // [n_q][n_head][n_split][D + 2]
};

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned) h << 16); }
__device__ __forceinline__ float h2f(unsigned short h) { return (float) __builtin_bit_cast(_Float16, h); }
// ggml_compute_fp32_to_bf16 (/ref/ggml/src/ggml-impl.h:538-551) and back
__device__ __forceinline__ float bf_round(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return __uint_as_float(((u >> 16) | 64u) << 16);
    return __uint_as_float(((u + (0x7fffu + ((u >> 16) & 1u))) >> 16) << 16);
}
template <int T> __device__ __forceinline__ float e16(unsigned short h) { return T == PKV_TYPE_BF16 ? bf2f(h) : h2f(h); }

// the other CLI-whitelisted cache types (/ref/common/arg.cpp:172-182), blocks of 32 as the reference stores them (/ref/ggml/src/ggml-common.h:168-203, :421-426):
//   q4_1 {half d, m; u8 qs[16]} 20 B   q5_0 {half d; u8 qh[4]; u8 qs[16]} 22 B   q5_1 {half d, m; u8 qh[4]; u8 qs[16]} 24 B   iq4_nl {half d; u8 qs[16]} 18 B
// element e: low nibbles are elements 0..15, high nibbles 16..31; q5: bit e of qh is the fifth bit; iq4_nl: the nibble indexes kvalues_iq4nl
template <int T> struct Blk;
template <> struct Blk<PKV_TYPE_Q4_1>   { static constexpr int B = 20, QS = 4, QH = -1, M = 2;  };
template <> struct Blk<PKV_TYPE_Q5_0>   { static constexpr int B = 22, QS = 6, QH = 2,  M = -1; };
template <> struct Blk<PKV_TYPE_Q5_1>   { static constexpr int B = 24, QS = 8, QH = 4,  M = 2;  };
template <> struct Blk<PKV_TYPE_IQ4_NL> { static constexpr int B = 18, QS = 2, QH = -1, M = -1; };
template <int T> constexpr bool is_new_blk = T == PKV_TYPE_Q4_1 || T == PKV_TYPE_Q5_0 || T == PKV_TYPE_Q5_1 || T == PKV_TYPE_IQ4_NL;
template <int T> constexpr bool is_q81 = T == PKV_TYPE_Q4_1 || T == PKV_TYPE_Q5_1;        // K types whose vec_dot_type is q8_1 (type_traits_cpu)
__device__ __forceinline__ int iq4nl_value(int nib) {                                   // kvalues_iq4nl (/ref/ggml/src/ggml-quants.c:2434), two 64-bit tables
    const unsigned long long lo = 0xF6EADDCFBFAD9881ull, hi = 0x715945352619'0D01ull;
    return (int) (signed char) ((nib < 8 ? lo >> (8 * nib) : hi >> (8 * (nib - 8))) & 0xff);
}
__device__ __forceinline__ unsigned ld_u32_2(const char * p) {                          // 4 bytes at a 2-byte aligned address
    return (unsigned) *reinterpret_cast<const unsigned short *>(p) | ((unsigned) *reinterpret_cast<const unsigned short *>(p + 2) << 16);
}
// the integer value of element e of a block (q5_0: minus 16; iq4_nl: the table value) given its nibble and fifth bit
template <int T> __device__ __forceinline__ int blk_value(int nib, unsigned qh, int e) {
    if constexpr (T == PKV_TYPE_Q4_1)      return nib;
    else if constexpr (T == PKV_TYPE_IQ4_NL) return iq4nl_value(nib);
    else if constexpr (T == PKV_TYPE_Q5_0) return (nib | (int) (((qh >> e) & 1u) << 4)) - 16;
    else                                   return nib | (int) (((qh >> e) & 1u) << 4);
}

// the workgroup's result for column `col` of (query row, head): the output itself (a row that sees no cell: zeros, as the paged kernels), or — split-KV — its
// (m, l, unnormalised O) partial
__device__ __forceinline__ void flat_store(const FlatArgs & a, int iq, int h, int col, float M, float L, float o) {
    if (a.n_split == 1) { a.out[((size_t) iq * a.n_head + h) * a.D + col] = L > 0.0f ? o * (1.0f / L) : 0.0f; return; }
    float * p = a.part + (((size_t) iq * a.n_head + h) * a.n_split + blockIdx.z) * (a.D + 2);
    p[2 + col] = o;
    if (col == 0) { p[0] = M; p[1] = L; }
}
// grid (n_q, n_head) x 256: out = sum_z O_z e^{m_z - M} / sum_z l_z e^{m_z - M}
__global__ void __launch_bounds__(FLAT_DMAX) flat_merge_kernel(const FlatArgs a) {
    const int iq = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
    if (d >= a.D) return;
    const float * p = a.part + ((size_t) iq * a.n_head + h) * a.n_split * (a.D + 2);
    float M = -INFINITY;
    for (int z = 0; z < a.n_split; ++z) M = fmaxf(M, p[(size_t) z * (a.D + 2)]);
    float L = 0.0f, o = 0.0f;
    if (M != -INFINITY) {
        for (int z = 0; z < a.n_split; ++z) {
            const float * pz = p + (size_t) z * (a.D + 2);
            const float f = expf(pz[0] - M);
            L = fmaf(pz[1], f, L);
            o = fmaf(pz[2 + d], f, o);
        }
    }
    a.out[((size_t) iq * a.n_head + h) * a.D + d] = L > 0.0f ? o * (1.0f / L) : 0.0f;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// the dot of ONE K row with the Q operand in LDS (sQ: the rounded Q as floats — integers for quantized K; sQd: its block scales)
template <int KT>
__device__ __forceinline__ float k_dot(const char * row, const float * sQ, const float * sQd, int D, bool al16) {
    float acc = 0.0f;
    if constexpr (KT == PKV_TYPE_F16 || KT == PKV_TYPE_BF16) {
        if (al16) {
            for (int c = 0; c < D / 8; ++c) {
                const uint4 w = *reinterpret_cast<const uint4 *>(row + 16 * c);
                const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc = fmaf(e16<KT>((unsigned short) (u[i] & 0xffffu)), sQ[8 * c + 2 * i], acc);
                    acc = fmaf(e16<KT>((unsigned short) (u[i] >> 16)), sQ[8 * c + 2 * i + 1], acc);
                }
            }
            for (int d = D & ~7; d < D; ++d) acc = fmaf(e16<KT>(*reinterpret_cast<const unsigned short *>(row + 2 * d)), sQ[d], acc);
        } else {
            for (int d = 0; d < D; ++d) acc = fmaf(e16<KT>(*reinterpret_cast<const unsigned short *>(row + 2 * d)), sQ[d], acc);
        }
    } else if constexpr (KT == PKV_TYPE_Q8_0) {                      // block_q8_0 {half d; int8 qs[32]} (/ref/ggml/src/ggml-common.h:203-208)
        for (int b = 0; b < D / PKV_QK; ++b) {
            const char * blk = row + 34 * b;
            const float dk = h2f(*reinterpret_cast<const unsigned short *>(blk));
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned short w = *reinterpret_cast<const unsigned short *>(blk + 2 + 2 * j);
                s = fmaf((float) (signed char) (w & 0xff), sQ[32 * b + 2 * j], s);
                s = fmaf((float) (signed char) (w >> 8), sQ[32 * b + 2 * j + 1], s);
            }
            acc = fmaf(dk * sQd[b], s, acc);                         // (ggml_vec_dot_q8_0_q8_0: d_x * d_y * sum of integer products)
        }
    } else if constexpr (is_new_blk<KT>) {                           // q4_1 / q5_1 against q8_1 (+ m_k * s_q per block), q5_0 / iq4_nl against q8_0
        using BK = Blk<KT>;
        float summs = 0.0f;
        for (int b = 0; b < D / PKV_QK; ++b) {
            const char * blk = row + BK::B * b;
            const float dk = h2f(*reinterpret_cast<const unsigned short *>(blk));
            unsigned qh = 0;
            if constexpr (BK::QH >= 0) qh = ld_u32_2(blk + BK::QH);
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w = *reinterpret_cast<const unsigned short *>(blk + BK::QS + 2 * j);
                s = fmaf((float) blk_value<KT>((int) (w & 0xf), qh, 2 * j), sQ[32 * b + 2 * j], s);
                s = fmaf((float) blk_value<KT>((int) ((w >> 4) & 0xf), qh, 2 * j + 16), sQ[32 * b + 2 * j + 16], s);
                s = fmaf((float) blk_value<KT>((int) ((w >> 8) & 0xf), qh, 2 * j + 1), sQ[32 * b + 2 * j + 1], s);
                s = fmaf((float) blk_value<KT>((int) (w >> 12), qh, 2 * j + 17), sQ[32 * b + 2 * j + 17], s);
            }
            acc = fmaf(dk * sQd[b], s, acc);
            if constexpr (BK::M >= 0) summs = __fadd_rn(summs, __fmul_rn(h2f(*reinterpret_cast<const unsigned short *>(blk + BK::M)), sQd[FLAT_DMAX / PKV_QK + b]));
        }
        acc += summs;
    } else {                                                         // block_q4_0 {half d; uint8 qs[16]}: byte j = elem j | elem j + 16 << 4, minus 8 (:161-166)
        for (int b = 0; b < D / PKV_QK; ++b) {
            const char * blk = row + 18 * b;
            const float dk = h2f(*reinterpret_cast<const unsigned short *>(blk));
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w = *reinterpret_cast<const unsigned short *>(blk + 2 + 2 * j);
                s = fmaf((float) ((int) (w & 0xf) - 8), sQ[32 * b + 2 * j], s);
                s = fmaf((float) ((int) ((w >> 4) & 0xf) - 8), sQ[32 * b + 2 * j + 16], s);
                s = fmaf((float) ((int) ((w >> 8) & 0xf) - 8), sQ[32 * b + 2 * j + 1], s);
                s = fmaf((float) ((int) (w >> 12) - 8), sQ[32 * b + 2 * j + 17], s);
            }
            acc = fmaf(dk * sQd[b], s, acc);
        }
    }
    return acc;
}

// element d of ONE V row as the reference's to_float gives it (dequantize_row_q8_0 / _q4_0: quant * d)
template <int VT>
__device__ __forceinline__ float v_elem(const char * row, int d) {
    if constexpr (VT == PKV_TYPE_F16 || VT == PKV_TYPE_BF16) {
        return e16<VT>(*reinterpret_cast<const unsigned short *>(row + 2 * d));
    } else if constexpr (VT == PKV_TYPE_Q8_0) {
        const char * blk = row + 34 * (d >> 5);
        return (float) *reinterpret_cast<const signed char *>(blk + 2 + (d & 31)) * h2f(*reinterpret_cast<const unsigned short *>(blk));
    } else if constexpr (is_new_blk<VT>) {                           // dequantize_row_q4_1 / _q5_0 / _q5_1 / _iq4_nl: q * d (+ m), product and sum rounded separately
        using BV = Blk<VT>;
        const char * blk = row + BV::B * (d >> 5);
        const int e = d & 31;
        const unsigned byte = *reinterpret_cast<const unsigned char *>(blk + BV::QS + (e & 15));
        unsigned qh = 0;
        if constexpr (BV::QH >= 0) qh = ld_u32_2(blk + BV::QH);
        const float x = __fmul_rn((float) blk_value<VT>((e & 16) ? (int) (byte >> 4) : (int) (byte & 0xf), qh, e), h2f(*reinterpret_cast<const unsigned short *>(blk)));
        if constexpr (BV::M >= 0) return __fadd_rn(x, h2f(*reinterpret_cast<const unsigned short *>(blk + BV::M)));
        else return x;
    } else {
        const char * blk = row + 18 * (d >> 5);
        const unsigned byte = *reinterpret_cast<const unsigned char *>(blk + 2 + (d & 15));
        const int nib = (d & 16) ? (int) (byte >> 4) : (int) (byte & 0xf);
        return (float) (nib - 8) * h2f(*reinterpret_cast<const unsigned short *>(blk));
    }
}

// grid (n_q, n_head) x 256 (1024) threads
template <int KT, int VT>
__global__ void __launch_bounds__(64 * FLAT_NW_LONG) attn_flat_kernel(const FlatArgs a) {
    const int iq = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int D = a.D, hk = h / (a.n_head / a.n_head_kv), nw = (int) (blockDim.x >> 6);
    __shared__ float sQ[FLAT_DMAX], sQd[2 * FLAT_DMAX / PKV_QK], sAcc[FLAT_NW_LONG][FLAT_DMAX], sM[FLAT_NW_LONG], sL[FLAT_NW_LONG];

    // ---- the Q operand, rounded as the K type's vec_dot_type rounds it (:12343)
    {
        const float x = tid < D ? a.q[(size_t) iq * a.q_s_tok + (size_t) h * a.q_s_head + tid] : 0.0f;
        if (tid >= FLAT_DMAX) {}                                     // (the long form has more threads than columns)
        else if constexpr (KT == PKV_TYPE_F16)  sQ[tid] = (float) (_Float16) x;
        else if constexpr (KT == PKV_TYPE_BF16) sQ[tid] = bf_round(x);
        else {                                                       // quantize_row_q8_0, AVX2 branch (/ref/ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862)
            float am = fabsf(x);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));   // the 32 lanes of one block
            const float id = am != 0.0f ? 127.0f / am : 0.0f;
            const float qi = rintf(x * id);
            sQ[tid] = qi;
            if ((tid & 31) == 0) sQd[tid >> 5] = (float) (_Float16) (am / 127.0f);
            if constexpr (is_q81<KT>) {                              // quantize_row_q8_1 (:1138-1180): s = f16(d * sum of the block's quants), d NOT yet rounded to f16
                float sm = qi;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) sm += __shfl_xor(sm, o);
                if ((tid & 31) == 0) sQd[FLAT_DMAX / PKV_QK + (tid >> 5)] = (float) (_Float16) __fmul_rn(am / 127.0f, sm);
            }
        }
    }
    __syncthreads();

    const float slope = a.max_bias > 0.0f ? ((unsigned) h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2 * (h - (int) a.n_head_log2) + 1))) : 1.0f;   // :12315-12316
    const _Float16 * mrow = a.mask ? a.mask + (size_t) iq * a.mask_stride : nullptr;
    const char * kb = a.k + (size_t) hk * a.k_s_head, * vb = a.v + (size_t) hk * a.v_s_head;
    const bool ksafe = a.k_al16 != 0;
    float m = -INFINITY, l = 0.0f, acc[FLAT_DMAX / 64] = {0.0f, 0.0f, 0.0f, 0.0f};

    const int kv_beg = (int) blockIdx.z * a.split_len, kv_end = min(a.n_kv, kv_beg + a.split_len);
    for (int c0 = kv_beg + wave * 64; c0 < kv_end; c0 += nw * 64) {
        const int t = c0 + lane;
        float s = -INFINITY;
        if (t < kv_end) {
            const float mv = mrow ? slope * (float) mrow[t] : 0.0f;
            if (mv != -INFINITY) {                                    // :12349-12352: a masked cell is skipped
                s = k_dot<KT>(kb + (size_t) t * a.k_s_tok, sQ, sQd, D, ksafe) * a.scale;
                if (a.softcap != 0.0f) s = a.softcap * tanhf(s);
                s += mv;
            }
        }
        const float cm = wave_max(s);
        if (cm == -INFINITY) continue;                               // (wave-uniform) nothing visible in this chunk
        const float mn = fmaxf(m, cm);
        const float alpha = expf(m - mn), p = expf(s - mn);          // exp(-inf) = 0: the first chunk, masked cells
        l = l * alpha + wave_sum(p);
        m = mn;
#pragma unroll
        for (int i = 0; i < FLAT_DMAX / 64; ++i) acc[i] *= alpha;
        const int nt = min(64, kv_end - c0);
        for (int j = 0; j < nt; ++j) {
            const float pj = __shfl(p, j);
            if (pj == 0.0f) continue;                                // (wave-uniform)
            const char * vrow = vb + (size_t) (c0 + j) * a.v_s_tok;
#pragma unroll
            for (int i = 0; i < FLAT_DMAX / 64; ++i) {
                const int d = lane + 64 * i;
                if (d < D) acc[i] = fmaf(v_elem<VT>(vrow, d), pj, acc[i]);
            }
        }
    }

    // ---- one merge across the waves
    if (lane == 0) { sM[wave] = m; sL[wave] = l; }
#pragma unroll
    for (int i = 0; i < FLAT_DMAX / 64; ++i) sAcc[wave][lane + 64 * i] = acc[i];
    __syncthreads();
    if (tid < D) {
        float M = sM[0];
        for (int w = 1; w < nw; ++w) M = fmaxf(M, sM[w]);
        float L = 0.0f, o = 0.0f;
        if (M != -INFINITY) {
            for (int w = 0; w < nw; ++w) {
                const float f = expf(sM[w] - M);
                L = fmaf(sL[w], f, L);
                o = fmaf(sAcc[w][tid], f, o);
            }
        }
        flat_store(a, iq, h, tid, M, L, o);
    }
}

// ---- the same operator for 16-bit rows (f16 / bf16 K and V, 16-byte aligned): LPR = D / 8 (rounded up to a power of two) lanes share a row, 16 bytes each, so
//      one wave instruction reads 64 / LPR whole rows and every byte of a line is used (the lane-per-row form above keeps the texture path busy with 64 lines
//      per K instruction and 2-byte V loads: profiles/r03_flat_bench.txt).  The lane's eight Q values stay in registers; the partial dots of a row are
//      summed across its LPR lanes, lane j of a row slot keeps the logit of the slot's j-th row — one logit per lane, as above —, the weights go back to the
//      rows' lanes by a shuffle for PV, and the R = 64 / LPR row slots' accumulators are added at the end.
template <int KT, int VT, int LPR>
__global__ void __launch_bounds__(64 * FLAT_NW_LONG) attn_flat16_kernel(const FlatArgs a) {
    constexpr int R = 64 / LPR;
    const int iq = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int D = a.D, hk = h / (a.n_head / a.n_head_kv), nw = (int) (blockDim.x >> 6);
    const int g = lane / LPR, j = lane % LPR;
    const bool active = 8 * j < D;
    __shared__ float sAcc[FLAT_NW_LONG][FLAT_DMAX], sM[FLAT_NW_LONG], sL[FLAT_NW_LONG];

    float qv[8];
    {
        const float * qrow = a.q + (size_t) iq * a.q_s_tok + (size_t) h * a.q_s_head + 8 * j;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = active ? qrow[e] : 0.0f;
            qv[e] = KT == PKV_TYPE_F16 ? (float) (_Float16) x : bf_round(x);
        }
    }
    const float slope = a.max_bias > 0.0f ? ((unsigned) h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2 * (h - (int) a.n_head_log2) + 1))) : 1.0f;
    const _Float16 * mrow = a.mask ? a.mask + (size_t) iq * a.mask_stride : nullptr;
    const char * kb = a.k + (size_t) hk * a.k_s_head + 16 * j, * vb = a.v + (size_t) hk * a.v_s_head + 16 * j;
    float m = -INFINITY, l = 0.0f, acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    auto dot8 = [&](const uint4 & w) {
        const unsigned u[4] = {w.x, w.y, w.z, w.w};
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s = fmaf(e16<KT>((unsigned short) (u[i] & 0xffffu)), qv[2 * i], s);
            s = fmaf(e16<KT>((unsigned short) (u[i] >> 16)), qv[2 * i + 1], s);
        }
        return s;
    };

    const int kv_beg = (int) blockIdx.z * a.split_len, kv_end = min(a.n_kv, kv_beg + a.split_len);
    for (int c0 = kv_beg + wave * 64; c0 < kv_end; c0 += nw * 64) {
        // ---- logits: instruction i covers the rows c0 + i * R + g; lane j of a slot keeps row j's
        float sraw = 0.0f;
#pragma unroll 4
        for (int i = 0; i < LPR; ++i) {
            const int t = c0 + i * R + g;
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (active && t < kv_end) w = *reinterpret_cast<const uint4 *>(kb + (size_t) t * a.k_s_tok);
            float part = dot8(w);
#pragma unroll
            for (int o = LPR / 2; o >= 1; o >>= 1) part += __shfl_xor(part, o);
            if (i == j) sraw = part;
        }
        const int t = c0 + j * R + g;                                 // this lane's cell
        float s = -INFINITY;
        if (t < kv_end) {
            const float mv = mrow ? slope * (float) mrow[t] : 0.0f;
            if (mv != -INFINITY) {
                s = sraw * a.scale;
                if (a.softcap != 0.0f) s = a.softcap * tanhf(s);
                s += mv;
            }
        }
        const float cm = wave_max(s);
        if (cm == -INFINITY) continue;                               // (wave-uniform)
        const float mn = fmaxf(m, cm);
        const float alpha = expf(m - mn), p = expf(s - mn);
        l = l * alpha + wave_sum(p);
        m = mn;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= alpha;
        // ---- PV: the weight of row c0 + i * R + g sits in lane i of this slot
#pragma unroll 4
        for (int i = 0; i < LPR; ++i) {
            const int tr = c0 + i * R + g;
            const float pr = __shfl(p, (lane & ~(LPR - 1)) | i);
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (active && tr < kv_end) w = *reinterpret_cast<const uint4 *>(vb + (size_t) tr * a.v_s_tok);
            const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc[2 * x]     = fmaf(e16<VT>((unsigned short) (u[x] & 0xffffu)), pr, acc[2 * x]);
                acc[2 * x + 1] = fmaf(e16<VT>((unsigned short) (u[x] >> 16)), pr, acc[2 * x + 1]);
            }
        }
    }
    // the row slots of the wave, then the waves
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o);
    if (lane == 0) { sM[wave] = m; sL[wave] = l; }
    if (g == 0 && active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sAcc[wave][8 * j + e] = acc[e];
    }
    __syncthreads();
    if (tid < D) {
        float M = sM[0];
        for (int w = 1; w < nw; ++w) M = fmaxf(M, sM[w]);
        float L = 0.0f, o = 0.0f;
        if (M != -INFINITY) {
            for (int w = 0; w < nw; ++w) {
                const float f = expf(sM[w] - M);
                L = fmaf(sL[w], f, L);
                o = fmaf(sAcc[w][tid], f, o);
            }
        }
        flat_store(a, iq, h, tid, M, L, o);
    }
}

template <int KT, int VT>
int launch_flat16(const FlatArgs & fa, hipStream_t st) {
    const dim3 grid(fa.n_q, fa.n_head, fa.n_split), block(64 * (fa.split_len >= 64 * FLAT_NW_LONG ? FLAT_NW_LONG : FLAT_NW));
    const int lanes = (fa.D + 7) / 8;
    if (lanes <= 4)       hipLaunchKernelGGL((attn_flat16_kernel<KT, VT, 4>), grid, block, 0, st, fa);
    else if (lanes <= 8)  hipLaunchKernelGGL((attn_flat16_kernel<KT, VT, 8>), grid, block, 0, st, fa);
    else if (lanes <= 16) hipLaunchKernelGGL((attn_flat16_kernel<KT, VT, 16>), grid, block, 0, st, fa);
    else                  hipLaunchKernelGGL((attn_flat16_kernel<KT, VT, 32>), grid, block, 0, st, fa);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

template <int KT>
int launch_v(const FlatArgs & fa, int type_v, hipStream_t st) {
    const dim3 grid(fa.n_q, fa.n_head, fa.n_split), block(64 * (fa.split_len >= 64 * FLAT_NW_LONG ? FLAT_NW_LONG : FLAT_NW));
    switch (type_v) {
#define PKV_FLAT_V(VT) case VT: hipLaunchKernelGGL((attn_flat_kernel<KT, VT>), grid, block, 0, st, fa); break
        PKV_FLAT_V(PKV_TYPE_F16); PKV_FLAT_V(PKV_TYPE_BF16); PKV_FLAT_V(PKV_TYPE_Q8_0); PKV_FLAT_V(PKV_TYPE_Q4_0);
        PKV_FLAT_V(PKV_TYPE_Q4_1); PKV_FLAT_V(PKV_TYPE_Q5_0); PKV_FLAT_V(PKV_TYPE_Q5_1); PKV_FLAT_V(PKV_TYPE_IQ4_NL);
#undef PKV_FLAT_V
        default: return PKV_ERR_UNSUPPORTED;
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

bool flat_blocked(int t) { return t == PKV_TYPE_Q8_0 || t == PKV_TYPE_Q4_0 || t == PKV_TYPE_Q4_1 || t == PKV_TYPE_Q5_0 || t == PKV_TYPE_Q5_1 || t == PKV_TYPE_IQ4_NL; }
bool flat_type_ok(int t) { return t == PKV_TYPE_F16 || t == PKV_TYPE_BF16 || flat_blocked(t); }
size_t flat_row_bytes(int t, int D) {
    const size_t nb = (size_t) D / PKV_QK;
    switch (t) {
        case PKV_TYPE_Q8_0: return nb * 34;
        case PKV_TYPE_Q4_0: case PKV_TYPE_IQ4_NL: return nb * 18;
        case PKV_TYPE_Q4_1: return nb * 20;
        case PKV_TYPE_Q5_0: return nb * 22;
        case PKV_TYPE_Q5_1: return nb * 24;
    }
    return (size_t) D * 2;
}

// ---- GGML_OP_CPY f32 -> q4_1 / q5_0 / q5_1 / iq4_nl (the KV store of a cache of those types): the CPU backend's from_float of each is the *_ref quantizer
//      (/ref/ggml/src/ggml-quants.c:68-190, :4573-4720).  One thread per block of 32, the reference's scalar loops with every product and sum rounded
//      separately (the reference is compiled without contraction): the same bytes.
template <int T>
__global__ void __launch_bounds__(256) quantize_blocks_kernel(const float * x, char * dst, int64_t n_blocks) {
    const int64_t ib = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (ib >= n_blocks) return;
    const float * xb = x + ib * PKV_QK;
    char * y = dst + ib * Blk<T>::B;
    float v[PKV_QK];
#pragma unroll
    for (int j = 0; j < PKV_QK; j += 4) { const float4 t = *reinterpret_cast<const float4 *>(xb + j); v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w; }
    // (the value passes through an opaque register first: hipcc otherwise folds a scale's last multiplication into the conversion as v_fma_mixlo_f16(x, c, +0),
    //  and (-0) + (+0) = +0 where the reference stores the -0 that max / -16 of an all-zero block is)
    auto put_h = [&](int off, float f) { asm volatile("" : "+v"(f)); *reinterpret_cast<_Float16 *>(y + off) = (_Float16) f; };
    unsigned char qs[16];
    unsigned qh = 0;
    if constexpr (T == PKV_TYPE_Q4_1 || T == PKV_TYPE_Q5_1) {
        constexpr int LV = T == PKV_TYPE_Q4_1 ? 15 : 31;
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
#pragma unroll
        for (int j = 0; j < PKV_QK; ++j) { if (v[j] < mn) mn = v[j]; if (v[j] > mx) mx = v[j]; }
        const float d = __fdiv_rn(__fsub_rn(mx, mn), (float) LV), id = d ? __fdiv_rn(1.0f, d) : 0.0f;
        put_h(0, d); put_h(2, mn);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float x0 = __fmul_rn(__fsub_rn(v[j], mn), id), x1 = __fmul_rn(__fsub_rn(v[j + 16], mn), id);
            unsigned xi0, xi1;
            if constexpr (T == PKV_TYPE_Q4_1) { const int a0 = (int) (signed char) (int) __fadd_rn(x0, 0.5f), a1 = (int) (signed char) (int) __fadd_rn(x1, 0.5f); xi0 = (unsigned) min(15, a0); xi1 = (unsigned) min(15, a1); }
            else                              { xi0 = (unsigned) (unsigned char) (int) __fadd_rn(x0, 0.5f); xi1 = (unsigned) (unsigned char) (int) __fadd_rn(x1, 0.5f); }
            qs[j] = (unsigned char) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << j;
            qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
    } else if constexpr (T == PKV_TYPE_Q5_0) {
        float amax = 0.0f, mx = 0.0f;
#pragma unroll
        for (int j = 0; j < PKV_QK; ++j) if (amax < fabsf(v[j])) { amax = fabsf(v[j]); mx = v[j]; }
        const float d = __fdiv_rn(mx, -16.0f), id = d ? __fdiv_rn(1.0f, d) : 0.0f;
        put_h(0, d);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int a0 = (int) (signed char) (int) __fadd_rn(__fmul_rn(v[j], id), 16.5f), a1 = (int) (signed char) (int) __fadd_rn(__fmul_rn(v[j + 16], id), 16.5f);
            const unsigned xi0 = (unsigned) (unsigned char) min(31, a0), xi1 = (unsigned) (unsigned char) min(31, a1);
            qs[j] = (unsigned char) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << j;
            qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
    } else {                                                          // iq4_nl, ntry = -1: first-pass indices, the scale re-fitted with weights x^2
        float amax = 0.0f, mx = 0.0f;
#pragma unroll
        for (int j = 0; j < PKV_QK; ++j) { const float ax = fabsf(v[j]); if (ax > amax) { amax = ax; mx = v[j]; } }
        float dfit = 0.0f;
        unsigned char L[PKV_QK];
#pragma unroll
        for (int j = 0; j < PKV_QK; ++j) L[j] = 0;
        if (amax >= 1e-15f) {
            const float id = __fdiv_rn(1.0f, __fdiv_rn(mx, -127.0f));
            float sumqx = 0.0f, sumq2 = 0.0f;
#pragma unroll
            for (int j = 0; j < PKV_QK; ++j) {
                const float al = __fmul_rn(id, v[j]);
                int l;                                                // best_index_int8 (:4562-4571)
                if (al <= -127.0f) l = 0;
                else if (al >= 113.0f) l = 15;
                else {
                    int ml = 0, mu = 15;
                    while (mu - ml > 1) { const int mav = (ml + mu) / 2; if (al < (float) iq4nl_value(mav)) mu = mav; else ml = mav; }
                    l = __fsub_rn(al, (float) iq4nl_value(mu - 1)) < __fsub_rn((float) iq4nl_value(mu), al) ? mu - 1 : mu;
                }
                L[j] = (unsigned char) l;
                const float q = (float) iq4nl_value(l), w = __fmul_rn(v[j], v[j]);
                sumqx = __fadd_rn(sumqx, __fmul_rn(__fmul_rn(w, q), v[j]));
                sumq2 = __fadd_rn(sumq2, __fmul_rn(__fmul_rn(w, q), q));
            }
            dfit = __fdiv_rn(sumqx, sumq2);
        }
        put_h(0, dfit);
#pragma unroll
        for (int j = 0; j < 16; ++j) qs[j] = (unsigned char) (L[j] | (L[16 + j] << 4));
    }
    if constexpr (Blk<T>::QH >= 0) { *reinterpret_cast<unsigned short *>(y + Blk<T>::QH) = (unsigned short) (qh & 0xffffu); *reinterpret_cast<unsigned short *>(y + Blk<T>::QH + 2) = (unsigned short) (qh >> 16); }
#pragma unroll
    for (int j = 0; j < 16; j += 2) *reinterpret_cast<unsigned short *>(y + Blk<T>::QS + j) = (unsigned short) (qs[j] | (qs[j + 1] << 8));
}

}  // namespace

// [n_q][n_head][FLAT_MAX_SPLIT][D + 2] floats: what the split-KV mode of a launch of this shape can use (0: it never splits)
size_t flat_workspace_bytes(int n_q, int n_head, int D, int n_kv) {
    const long long pairs = (long long) n_q * n_head;
    if (pairs <= 0 || pairs >= 1024 || n_kv < 512) return 0;
    return (size_t) pairs * FLAT_MAX_SPLIT * (D + 2) * sizeof(float);
}

int launch_attn_flat(const pkv_flat_attn_args & a, hipStream_t st) {
    PKV_REQUIRE(a.q && a.k && a.v && a.out, "pkv_flash_attn_ext_flat: null pointer");
    PKV_REQUIRE(a.n_q > 0 && a.n_kv > 0 && a.n_head > 0 && a.n_head_kv > 0 && a.n_head % a.n_head_kv == 0, "pkv_flash_attn_ext_flat: bad shape");
    PKV_REQUIRE(a.n_head <= 65535, "pkv_flash_attn_ext_flat: more than 65535 heads");
    if (!flat_type_ok(a.type_k) || !flat_type_ok(a.type_v)) { set_error("pkv_flash_attn_ext_flat: KV types must be f16 / bf16 / q8_0 / q4_0 / q4_1 / q5_0 / q5_1 / iq4_nl"); return PKV_ERR_UNSUPPORTED; }
    const bool quant = flat_blocked(a.type_k) || flat_blocked(a.type_v);
    if (a.head_dim <= 0 || a.head_dim > FLAT_DMAX || a.head_dim % 2 != 0 || (quant && a.head_dim % PKV_QK != 0)) {
        set_error("pkv_flash_attn_ext_flat: head_dim %d (even, <= %d; a multiple of 32 with quantized K / V)", a.head_dim, FLAT_DMAX);
        return PKV_ERR_UNSUPPORTED;
    }
    PKV_REQUIRE(a.k_stride_tok >= (int64_t) flat_row_bytes(a.type_k, a.head_dim) && a.v_stride_tok >= (int64_t) flat_row_bytes(a.type_v, a.head_dim) &&
                a.k_stride_tok % 2 == 0 && a.k_stride_head % 2 == 0 && a.v_stride_tok % 2 == 0 && a.v_stride_head % 2 == 0 && ((uintptr_t) a.k % 2) == 0 && ((uintptr_t) a.v % 2) == 0,
                "pkv_flash_attn_ext_flat: K / V rows must be 2-byte aligned and at least one row apart");
    PKV_REQUIRE(a.max_bias == 0.0f || a.mask, "pkv_flash_attn_ext_flat: ALiBi needs a mask");
    PKV_REQUIRE(!a.mask || a.mask_stride >= a.n_kv, "pkv_flash_attn_ext_flat: mask rows shorter than n_kv");
    FlatArgs fa;
    fa.q = a.q; fa.q_s_tok = a.q_stride_tok; fa.q_s_head = a.q_stride_head;
    fa.k = (const char *) a.k; fa.v = (const char *) a.v;
    fa.k_s_tok = a.k_stride_tok; fa.k_s_head = a.k_stride_head; fa.v_s_tok = a.v_stride_tok; fa.v_s_head = a.v_stride_head;
    fa.mask = (const _Float16 *) a.mask; fa.mask_stride = a.mask_stride;
    fa.out = a.out;
    fa.n_q = a.n_q; fa.n_kv = a.n_kv; fa.n_head = a.n_head; fa.n_head_kv = a.n_head_kv; fa.D = a.head_dim;
    fa.softcap = a.logit_softcap;
    fa.scale = a.logit_softcap != 0.0f ? a.scale / a.logit_softcap : a.scale;                       // :12290-12292
    fa.max_bias = a.max_bias;
    fa.n_head_log2 = 1u << (unsigned) floor(log2((double) a.n_head));                                // :12295-12298
    fa.m0 = powf(2.0f, -(a.max_bias) / fa.n_head_log2);
    fa.m1 = powf(2.0f, -(a.max_bias / 2.0f) / fa.n_head_log2);
    fa.k_al16 = ((uintptr_t) a.k % 16) == 0 && a.k_stride_tok % 16 == 0 && a.k_stride_head % 16 == 0;
    fa.v_al16 = ((uintptr_t) a.v % 16) == 0 && a.v_stride_tok % 16 == 0 && a.v_stride_head % 16 == 0;
    // split-KV: a launch of few (query row, head) pairs over a long K / V (a decode step: 32 workgroups for 256 CUs) is cut along the cells into n_split
    // workgroups per pair that leave (m, l, O) partials in the caller's workspace, merged by a second small launch.  Whole 64-cell chunks per share, at least
    // 256 cells each; no workspace (or a short one): unsplit.
    fa.n_split = 1; fa.split_len = a.n_kv; fa.part = nullptr;
    {
        const long long pairs = (long long) a.n_q * a.n_head;
        int ns = (int) std::min<long long>(FLAT_MAX_SPLIT, (1024 + pairs - 1) / pairs);
        ns = std::min(ns, a.n_kv / 256);
        if (ns > 1 && a.workspace && a.workspace_bytes >= flat_workspace_bytes(a.n_q, a.n_head, a.head_dim, a.n_kv)) {
            const int len = (((a.n_kv + ns - 1) / ns) + 63) & ~63;
            fa.split_len = len; fa.n_split = (a.n_kv + len - 1) / len; fa.part = (float *) a.workspace;
        }
    }
    int rc = PKV_ERR_UNSUPPORTED;
    const bool k16 = a.type_k == PKV_TYPE_F16 || a.type_k == PKV_TYPE_BF16, v16 = a.type_v == PKV_TYPE_F16 || a.type_v == PKV_TYPE_BF16;
    static const bool no_rowcoop = getenv("PKV_FLAT_NO_ROWCOOP") != nullptr;   // (A/B against the lane-per-row form)
    if (k16 && v16 && fa.k_al16 && fa.v_al16 && a.head_dim % 8 == 0 && !no_rowcoop) {   // 16-bit rows, 16-byte aligned, whole 8-element chunks: the row-cooperative form
        if (a.type_k == PKV_TYPE_F16)  rc = a.type_v == PKV_TYPE_F16 ? launch_flat16<PKV_TYPE_F16, PKV_TYPE_F16>(fa, st) : launch_flat16<PKV_TYPE_F16, PKV_TYPE_BF16>(fa, st);
        else                           rc = a.type_v == PKV_TYPE_F16 ? launch_flat16<PKV_TYPE_BF16, PKV_TYPE_F16>(fa, st) : launch_flat16<PKV_TYPE_BF16, PKV_TYPE_BF16>(fa, st);
    } else {
        switch (a.type_k) {
            case PKV_TYPE_F16:  rc = launch_v<PKV_TYPE_F16>(fa, a.type_v, st); break;
            case PKV_TYPE_BF16: rc = launch_v<PKV_TYPE_BF16>(fa, a.type_v, st); break;
            case PKV_TYPE_Q8_0: rc = launch_v<PKV_TYPE_Q8_0>(fa, a.type_v, st); break;
            case PKV_TYPE_Q4_0: rc = launch_v<PKV_TYPE_Q4_0>(fa, a.type_v, st); break;
            case PKV_TYPE_Q4_1: rc = launch_v<PKV_TYPE_Q4_1>(fa, a.type_v, st); break;
            case PKV_TYPE_Q5_0: rc = launch_v<PKV_TYPE_Q5_0>(fa, a.type_v, st); break;
            case PKV_TYPE_Q5_1: rc = launch_v<PKV_TYPE_Q5_1>(fa, a.type_v, st); break;
            case PKV_TYPE_IQ4_NL: rc = launch_v<PKV_TYPE_IQ4_NL>(fa, a.type_v, st); break;
        }
    }
    if (rc == PKV_OK && fa.n_split > 1) {
        hipLaunchKernelGGL(flat_merge_kernel, dim3(fa.n_q, fa.n_head), dim3(FLAT_DMAX), 0, st, fa);
        PKV_HIP(hipGetLastError());
    }
    return rc;
}

// pkv_convert_f32 for the cache types that exist only as flat rows
int launch_quantize_blocks(const float * x, void * dst, int type, int64_t n_rows, int64_t ne0, hipStream_t st) {
    if (ne0 % PKV_QK != 0) { set_error("convert: row length %lld is not a multiple of 32", (long long) ne0); return PKV_ERR_UNSUPPORTED; }
    PKV_REQUIRE(((uintptr_t) x % 16) == 0 && ((uintptr_t) dst % 2) == 0, "convert: x must be 16-byte aligned, dst 2-byte aligned");
    const int64_t nb = n_rows * (ne0 / PKV_QK);
    if (nb == 0) return PKV_OK;
    const dim3 grid((unsigned) ((nb + 255) / 256)), block(256);
    switch (type) {
        case PKV_TYPE_Q4_1:   hipLaunchKernelGGL((quantize_blocks_kernel<PKV_TYPE_Q4_1>), grid, block, 0, st, x, (char *) dst, nb); break;
        case PKV_TYPE_Q5_0:   hipLaunchKernelGGL((quantize_blocks_kernel<PKV_TYPE_Q5_0>), grid, block, 0, st, x, (char *) dst, nb); break;
        case PKV_TYPE_Q5_1:   hipLaunchKernelGGL((quantize_blocks_kernel<PKV_TYPE_Q5_1>), grid, block, 0, st, x, (char *) dst, nb); break;
        case PKV_TYPE_IQ4_NL: hipLaunchKernelGGL((quantize_blocks_kernel<PKV_TYPE_IQ4_NL>), grid, block, 0, st, x, (char *) dst, nb); break;
        default: return PKV_ERR_UNSUPPORTED;
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

}  // namespace pkv
