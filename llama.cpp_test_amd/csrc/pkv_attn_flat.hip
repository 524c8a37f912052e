// pkv_attn_flat.hip — GGML_OP_FLASH_ATTN_EXT on the reference's OWN tensor layout (gfx950), for what the paged kernels do not take.
//
// The paged decode / prefill kernels (pkv_attn_decode.hip, pkv_attn_prefill.hip) are cut for the north-star geometries: head size 64 / 128, f16 / q8_0 / q4_0
// tiles of a block pool.  The operator they stand in for — ggml_compute_forward_flash_attn_ext_f16 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:12219-12432) — is
// wider: any head size, bf16 K / V (CLI-whitelisted cache type, /ref/common/arg.cpp:172-182), any gqa ratio, K / V as strided views of flat tensors
// (/ref/tests/test-backend-ops.cpp:4367-4390 tests head sizes 64 / 80 / 128 / 256 x f16 / bf16 / q8_0 / q4_0).  This file is that width: one kernel that reads
// the reference's rows where they lie (block_q8_0 34 B, block_q4_0 18 B, 2-byte aligned; 16-bit rows by 16-byte loads when aligned) — no pool, no
// conversion pass, bf16 stays bf16.
//
// Arithmetic follows the reference's operand roundings: Q goes through the K type's vec_dot_type first (:12343 — f16 for f16 K, bf16 for bf16 K
// (ggml_compute_fp32_to_bf16: nearest even), q8_0 with the AVX2 rounding for q8_0 / q4_0 K: id = 127 / amax, ties to even, d kept as f16); integer
// dots of the quantized types are exact in f32 (|sum| <= 32 * 127 * 127); scale, softcap (tanhf), mask * ALiBi slope as :12349-12365; V rows are
// accumulated in f32 (the reference: f32 for every V type but f16, where its f16 accumulator drifts from exact arithmetic by itself).
//
// Shape: one workgroup of 4 waves per (query row, head); wave w walks the 64-token chunks w, w + 4, ..: lane = token for QK^T (the Q operand sits in LDS,
// every lane reads the same address: a broadcast), online softmax per wave, then lane = column for PV (consecutive lanes read consecutive elements of one V
// row); one cross-wave merge through LDS.  A coverage kernel, not a roofline one: the hot geometries never come here (pkv_flash_attn_ext_flat's callers
// ask the paged kernels first).
#include "pkv_common.h"

namespace pkv {
namespace {

constexpr int FLAT_NW = 4, FLAT_DMAX = 256;

struct FlatArgs {
    const float * q; int64_t q_s_tok, q_s_head;
    const char * k, * v; int64_t k_s_tok, k_s_head, v_s_tok, v_s_head;
    const _Float16 * mask; int64_t mask_stride;
    float * out;
    int n_q, n_kv, n_head, n_head_kv, D;
    float scale, softcap, max_bias, m0, m1;
    unsigned n_head_log2;
    int k_al16, v_al16;
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
    } else {
        const char * blk = row + 18 * (d >> 5);
        const unsigned byte = *reinterpret_cast<const unsigned char *>(blk + 2 + (d & 15));
        const int nib = (d & 16) ? (int) (byte >> 4) : (int) (byte & 0xf);
        return (float) (nib - 8) * h2f(*reinterpret_cast<const unsigned short *>(blk));
    }
}

// grid (n_q, n_head) x 256 threads
template <int KT, int VT>
__global__ void __launch_bounds__(64 * FLAT_NW) attn_flat_kernel(const FlatArgs a) {
    const int iq = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int D = a.D, hk = h / (a.n_head / a.n_head_kv);
    __shared__ float sQ[FLAT_DMAX], sQd[FLAT_DMAX / PKV_QK], sAcc[FLAT_NW][FLAT_DMAX], sM[FLAT_NW], sL[FLAT_NW];

    // ---- the Q operand, rounded as the K type's vec_dot_type rounds it (:12343)
    {
        const float x = tid < D ? a.q[(size_t) iq * a.q_s_tok + (size_t) h * a.q_s_head + tid] : 0.0f;
        if constexpr (KT == PKV_TYPE_F16)       sQ[tid] = (float) (_Float16) x;
        else if constexpr (KT == PKV_TYPE_BF16) sQ[tid] = bf_round(x);
        else {                                                       // quantize_row_q8_0, AVX2 branch (/ref/ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862)
            float am = fabsf(x);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));   // the 32 lanes of one block
            const float id = am != 0.0f ? 127.0f / am : 0.0f;
            sQ[tid] = rintf(x * id);
            if ((tid & 31) == 0) sQd[tid >> 5] = (float) (_Float16) (am / 127.0f);
        }
    }
    __syncthreads();

    const float slope = a.max_bias > 0.0f ? ((unsigned) h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2 * (h - (int) a.n_head_log2) + 1))) : 1.0f;   // :12315-12316
    const _Float16 * mrow = a.mask ? a.mask + (size_t) iq * a.mask_stride : nullptr;
    const char * kb = a.k + (size_t) hk * a.k_s_head, * vb = a.v + (size_t) hk * a.v_s_head;
    const bool ksafe = a.k_al16 != 0;
    float m = -INFINITY, l = 0.0f, acc[FLAT_DMAX / 64] = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int c0 = wave * 64; c0 < a.n_kv; c0 += FLAT_NW * 64) {
        const int t = c0 + lane;
        float s = -INFINITY;
        if (t < a.n_kv) {
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
        const int nt = min(64, a.n_kv - c0);
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
#pragma unroll
        for (int w = 1; w < FLAT_NW; ++w) M = fmaxf(M, sM[w]);
        float L = 0.0f, o = 0.0f;
        if (M != -INFINITY) {
#pragma unroll
            for (int w = 0; w < FLAT_NW; ++w) {
                const float f = expf(sM[w] - M);
                L = fmaf(sL[w], f, L);
                o = fmaf(sAcc[w][tid], f, o);
            }
        }
        a.out[((size_t) iq * a.n_head + h) * D + tid] = L > 0.0f ? o * (1.0f / L) : 0.0f;   // (a row that sees no cell: zeros, as the paged kernels)
    }
}

template <int KT>
int launch_v(const FlatArgs & fa, int type_v, hipStream_t st) {
    const dim3 grid(fa.n_q, fa.n_head), block(64 * FLAT_NW);
    switch (type_v) {
        case PKV_TYPE_F16:  hipLaunchKernelGGL((attn_flat_kernel<KT, PKV_TYPE_F16>), grid, block, 0, st, fa); break;
        case PKV_TYPE_BF16: hipLaunchKernelGGL((attn_flat_kernel<KT, PKV_TYPE_BF16>), grid, block, 0, st, fa); break;
        case PKV_TYPE_Q8_0: hipLaunchKernelGGL((attn_flat_kernel<KT, PKV_TYPE_Q8_0>), grid, block, 0, st, fa); break;
        case PKV_TYPE_Q4_0: hipLaunchKernelGGL((attn_flat_kernel<KT, PKV_TYPE_Q4_0>), grid, block, 0, st, fa); break;
        default: return PKV_ERR_UNSUPPORTED;
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

bool flat_type_ok(int t) { return t == PKV_TYPE_F16 || t == PKV_TYPE_BF16 || t == PKV_TYPE_Q8_0 || t == PKV_TYPE_Q4_0; }
size_t flat_row_bytes(int t, int D) { return t == PKV_TYPE_Q8_0 ? (size_t) D / PKV_QK * 34 : t == PKV_TYPE_Q4_0 ? (size_t) D / PKV_QK * 18 : (size_t) D * 2; }

}  // namespace

int launch_attn_flat(const pkv_flat_attn_args & a, hipStream_t st) {
    PKV_REQUIRE(a.q && a.k && a.v && a.out, "pkv_flash_attn_ext_flat: null pointer");
    PKV_REQUIRE(a.n_q > 0 && a.n_kv > 0 && a.n_head > 0 && a.n_head_kv > 0 && a.n_head % a.n_head_kv == 0, "pkv_flash_attn_ext_flat: bad shape");
    PKV_REQUIRE(a.n_head <= 65535, "pkv_flash_attn_ext_flat: more than 65535 heads");
    if (!flat_type_ok(a.type_k) || !flat_type_ok(a.type_v)) { set_error("pkv_flash_attn_ext_flat: KV types must be f16 / bf16 / q8_0 / q4_0"); return PKV_ERR_UNSUPPORTED; }
    const bool quant = a.type_k == PKV_TYPE_Q8_0 || a.type_k == PKV_TYPE_Q4_0 || a.type_v == PKV_TYPE_Q8_0 || a.type_v == PKV_TYPE_Q4_0;
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
    fa.v_al16 = 0;
    switch (a.type_k) {
        case PKV_TYPE_F16:  return launch_v<PKV_TYPE_F16>(fa, a.type_v, st);
        case PKV_TYPE_BF16: return launch_v<PKV_TYPE_BF16>(fa, a.type_v, st);
        case PKV_TYPE_Q8_0: return launch_v<PKV_TYPE_Q8_0>(fa, a.type_v, st);
        case PKV_TYPE_Q4_0: return launch_v<PKV_TYPE_Q4_0>(fa, a.type_v, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

}  // namespace pkv
