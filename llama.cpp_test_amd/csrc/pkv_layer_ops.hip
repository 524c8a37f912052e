// pkv_layer_ops.hip — FIRST SLICE of SURVEY §8(f) row 1 ("rest of the decoder layer on-device"): RMSNorm, row-broadcast
// add / mul, SiLU and weight x activation products for f32 / f16 / q8_0 weights — enough for the reference runtime to keep a
// whole llama layer on the MI355X through the ggml-backend plugin.  Correct and parity-checked (the reference's
// test-backend-ops + an end-to-end llama_decode against the CPU backend); NOT yet tuned: the mat-vec streams a weight row per
// wave with 16-byte loads (f32/f16) or 2-byte-aligned block loads (q8_0) and re-reads the weights per activation column.
//
// Reference semantics: ggml_compute_forward_rms_norm_f32 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:8215-8260), ggml_silu_f32 (:2015),
// binary ops with row broadcast (:4560-4700), ggml_compute_forward_mul_mat (:7467-7640; dst[m][n] = sum_k src0[m][k] src1[n][k]).
#include "pkv_common.h"

#include <algorithm>

namespace pkv {
namespace {

template <int CTRL> __device__ __forceinline__ float dppl(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float q8k_wave_amax(float x) {  // all-reduce max over the wave without LDS (as pkv_gemv.hip's wave_amax)
    x = fmaxf(x, dppl<0xB1>(x)); x = fmaxf(x, dppl<0x4E>(x)); x = fmaxf(x, dppl<0x141>(x)); x = fmaxf(x, dppl<0x140>(x));
    { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    return x;
}
__device__ __forceinline__ int q8k_sum8(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);
    return x;
}
__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// one 256-thread workgroup per row; rows of up to 8192 elements stay in registers between the sum of squares and the scaling
// (one memory round trip instead of two: at decode there is ONE row, so the kernel is a latency chain, not a bandwidth problem).
// y_norm may be NULL; w (one row of ne0 weights) may be NULL (plain rms_norm: y = the normalised row).
// The weight row is requested together with x (it does not depend on the reduction), so its HBM latency is not a second trip.
constexpr int RMS_MAX_V4 = 8;   // float4 per thread: rows of up to 256 * 4 * 8 = 8192 elements
// res / x_out (both or neither): the residual add in front of the norm, x_out = x + res is what gets normalised (ADD -> RMS_NORM -> MUL
// of a llama layer, /ref/src/llama-model.cpp: ffn_inp = attn_out + inpL, then build_norm) — each thread adds, stores and keeps its own elements
// img (optional, in_regs rows with ne0 % 256 == 0, one row): the scaled row also leaves as the int8-per-32 activation image of pkv_gemv.hip
__global__ void __launch_bounds__(256) rms_norm_mul_kernel(const float * x, const float * res, float * x_out, const float * w, float * y_norm, float * y,
                                                           int64_t ne0, float eps, char * img = nullptr) {
    __shared__ double sred[4 * RMS_MAX_V4];                      // in_regs: one sum per 256-element block (pkv_common.h, canonical order)
    const size_t ro = (size_t) blockIdx.x * ne0;
    const float * xr = x + ro;
    const bool in_regs = (ne0 & 3) == 0 && ne0 <= 256 * 4 * RMS_MAX_V4;
    float4 v[RMS_MAX_V4], wv[RMS_MAX_V4];
    float s = 0.0f;
    if (in_regs) {
        const int64_t n4 = ne0 >> 2;
#pragma unroll
        for (int j = 0; j < RMS_MAX_V4; ++j) {
            const int64_t i = threadIdx.x + 256 * j;
            v[j]  = i < n4 ? reinterpret_cast<const float4 *>(xr)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[j] = (w && i < n4) ? reinterpret_cast<const float4 *>(w)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        }
        if (res) {
            float4 rv[RMS_MAX_V4];
#pragma unroll
            for (int j = 0; j < RMS_MAX_V4; ++j) {
                const int64_t i = threadIdx.x + 256 * j;
                rv[j] = i < n4 ? reinterpret_cast<const float4 *>(res + ro)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < RMS_MAX_V4; ++j) {
                const int64_t i = threadIdx.x + 256 * j;
                v[j] = make_float4(v[j].x + rv[j].x, v[j].y + rv[j].y, v[j].z + rv[j].z, v[j].w + rv[j].w);
                if (i < n4) reinterpret_cast<float4 *>(x_out + ro)[i] = v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < RMS_MAX_V4; ++j) {
            const int b = (int) (threadIdx.x >> 6) + 4 * j;          // the wave's 256-element block of pass j
            if ((int64_t) b * 64 < n4) {                             // wave-uniform
                const double bs = block_sumsq(v[j]);
                if ((threadIdx.x & 63) == 0) sred[b] = bs;
            }
        }
    } else {
        for (int64_t i = threadIdx.x; i < ne0; i += 256) {
            float t = xr[i];
            if (res) { t += res[ro + i]; x_out[ro + i] = t; }
            s = fmaf(t, t, s);
        }
    }
    float scale;
    if (in_regs) {
        __syncthreads();
        scale = rms_scale(sred, (int) (((ne0 >> 2) + 63) >> 6), ne0, eps);
    } else {
        s = wave_sum_f(s);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = (double) s;
        __syncthreads();
        const float mean = (float) (sred[0] + sred[1] + sred[2] + sred[3]) / (float) ne0;
        scale = 1.0f / sqrtf(mean + eps);
    }
    if (in_regs) {
        const int64_t n4 = ne0 >> 2;
#pragma unroll
        for (int j = 0; j < RMS_MAX_V4; ++j) {
            const int64_t i = threadIdx.x + 256 * j;
            if (i < n4) {
                const float4 n = make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale);
                if (y_norm) reinterpret_cast<float4 *>(y_norm + ro)[i] = n;
                if (y) reinterpret_cast<float4 *>(y + ro)[i] = w ? make_float4(n.x * wv[j].x, n.y * wv[j].y, n.z * wv[j].z, n.w * wv[j].w) : n;
            }
            if (img) {   // q8_K image (pkv_gemv.hip): the 64 lanes of a wave hold one 256-element block (ne0 % 256 == 0)
                const float4 o = w ? make_float4(v[j].x * scale * wv[j].x, v[j].y * scale * wv[j].y, v[j].z * scale * wv[j].z, v[j].w * scale * wv[j].w)
                                   : make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale);
                const float am = q8k_wave_amax(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
                const float iscale = am != 0.0f ? -127.0f / am : 0.0f;     // max = +amax (pkv_gemv.hip: the sign convention cancels in every product)
                const int q0 = (int) __builtin_rintf(iscale * o.x), q1 = (int) __builtin_rintf(iscale * o.y), q2 = (int) __builtin_rintf(iscale * o.z),
                          q3 = (int) __builtin_rintf(iscale * o.w);
                const int qs = q8k_sum8(q0 + q1 + q2 + q3);
                if (i < n4) {
                    reinterpret_cast<uint32_t *>(img)[i] = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
                    if ((i & 63) == 0) reinterpret_cast<float *>(img + ne0)[i >> 6] = am != 0.0f ? 1.0f / iscale : 0.0f;
                    if ((i & 7) == 0) reinterpret_cast<int *>(img + ne0 + (ne0 >> 8) * 4)[i >> 3] = qs;
                }
            }
        }
    } else {
        const float * xs = res ? x_out + ro : xr;                 // the thread's own stores of the first pass
        for (int64_t i = threadIdx.x; i < ne0; i += 256) {
            const float n = xs[i] * scale;
            if (y_norm) y_norm[ro + i] = n;
            if (y) y[ro + i] = w ? n * w[i] : n;
        }
    }
}
// ---- ggml_soft_max_ext (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10222-10318): the non-flash-attention path's softmax over the KQ logits
// (/ref/src/llama-graph.cpp:1201-1239): y[r][:] = softmax(x[r][:] * scale + slope(head) * mask[r % ne01][:]); mask f32 or f16 (or none),
// ALiBi slope per head (ne02 heads) when max_bias > 0.  One workgroup per row; three passes over the thread's own elements of the
// output row (w -> exp(w - max) -> / sum), the sum accumulated in f64 like ggml_vec_soft_max_f32's ggml_float.
template <typename MT>
__global__ void __launch_bounds__(256) soft_max_kernel(const float * x, const MT * mask, float * y, int64_t ne00, int64_t ne01, int64_t ne02, float scale, float max_bias,
                                                       float m0, float m1, uint32_t n_head_log2) {
    __shared__ float  sm[4];
    __shared__ double ss[4];
    const int64_t r = blockIdx.x;
    const uint32_t h = (uint32_t) ((r / ne01) % ne02);
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const float * xr = x + r * ne00;
    float * yr = y + r * ne00;
    const MT * mr = mask ? mask + (r % ne01) * ne00 : nullptr;
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < ne00; i += 256) {
        float w = xr[i] * scale;
        if (mr) w += slope * (float) mr[i];
        yr[i] = w;
        mx = fmaxf(mx, w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < ne00; i += 256) {
        const float e = expf(yr[i] - mx);
        yr[i] = e;
        sum += (double) e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = (float) (1.0 / (ss[0] + ss[1] + ss[2] + ss[3]));
    for (int64_t i = threadIdx.x; i < ne00; i += 256) yr[i] *= inv;
}
// ---- the two generic ops of the non-flash-attention graphs (`-fa 0`, /ref/src/llama-graph.cpp:1201-1239, :1387-1394): NOT a hot path — correct,
// coalesced where the layout allows, nothing more.
// ggml_cpy / ggml_cont between strided tensors of equal element count (the V store into the TRANSPOSED v cache: src = transpose(v_cur), dst = a
// 2-D strided f16 view; kqv_merged -> cont): element i of the source, in its own index order, becomes element i of the destination in ITS order
// (ggml_compute_forward_dup_f32, /ref/ggml/src/ggml-cpu/ggml-cpu.c:3550-3800)
struct Strided4 { int64_t ne[4], nb[4]; };
__device__ __forceinline__ int64_t offset_of(int64_t i, const Strided4 & t) {
    const int64_t i0 = i % t.ne[0]; i /= t.ne[0];
    const int64_t i1 = i % t.ne[1]; i /= t.ne[1];
    const int64_t i2 = i % t.ne[2], i3 = i / t.ne[2];
    return i0 * t.nb[0] + i1 * t.nb[1] + i2 * t.nb[2] + i3 * t.nb[3];
}
// ggml_bf16_t with the reference's conversions (ggml_compute_fp32_to_bf16 / _bf16_to_fp32, /ref/ggml/src/ggml-impl.h:519-551: nearest even, NaNs forced quiet)
struct bf16_t {
    unsigned short bits;
    bf16_t() = default;
    __device__ explicit bf16_t(float f) {
        const unsigned u = __float_as_uint(f);
        bits = (u & 0x7fffffffu) > 0x7f800000u ? (unsigned short) ((u >> 16) | 64u) : (unsigned short) ((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
    }
    __device__ explicit operator float() const { return __uint_as_float((unsigned) bits << 16); }
};
template <typename ST, typename DT>
__global__ void __launch_bounds__(256) cpy_strided_kernel(const char * src, char * dst, int64_t n, const Strided4 s, const Strided4 d) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256)
        *reinterpret_cast<DT *>(dst + offset_of(i, d)) = (DT) (float) *reinterpret_cast<const ST *>(src + offset_of(i, s));
}
// ggml_mul_mat on strided / batched operands (kq = k . q over permuted views of the cache and of Qcur; kqv = v^T . kq): y[b3][b2][n][m] =
// sum_k w[b3 / r3][b2 / r2][m][k] x[b3][b2][n][k], w f16 or f32 with unit stride in k, x f32 with unit stride in k, y contiguous
// (ggml_compute_forward_mul_mat, /ref/ggml/src/ggml-cpu/ggml-cpu.c:7467-7640; f32 accumulation).  One wave per output row m, 4 columns at a time.
template <typename WT>
__global__ void __launch_bounds__(256) mul_mat_strided_kernel(const char * w, const char * x, float * y, const Strided4 sw, const Strided4 sx, int64_t r2, int64_t r3) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.z, b2 = b % sx.ne[2], b3 = b / sx.ne[2];
    const int64_t K = sw.ne[0], M = sw.ne[1], N = sx.ne[1];
    if (m >= M) return;
    const char * wr = w + m * sw.nb[1] + (b2 / r2) * sw.nb[2] + (b3 / r3) * sw.nb[3];
    for (int64_t n0 = (int64_t) blockIdx.y * 4; n0 < N; n0 += (int64_t) gridDim.y * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const char * xr = x + b2 * sx.nb[2] + b3 * sx.nb[3];
        for (int64_t k = lane; k < K; k += 64) {
            const float wv = (float) reinterpret_cast<const WT *>(wr)[k];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (n0 + c < N) acc[c] = fmaf(wv, reinterpret_cast<const float *>(xr + (n0 + c) * sx.nb[1])[k], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v = wave_sum_f(acc[c]);
            if (lane == 0 && n0 + c < N) y[((b3 * sx.ne[2] + b2) * N + n0 + c) * M + m] = v;
        }
    }
}
// SiLU(gate) * up (the SwiGLU product of the FFN): one launch for the pair
__global__ void __launch_bounds__(256) silu_mul_kernel(const float * g, const float * u, float * y_silu, float * y, int64_t n) {
    for (int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t) gridDim.x * 256) {
        const float v = g[e], sv = v / (1.0f + expf(-v));
        if (y_silu) y_silu[e] = sv;
        y[e] = sv * u[e];
    }
}

// y[r][i] = a[r][i] (op) b[r % b_rows][i]
template <int OP>
__global__ void __launch_bounds__(256) binary_kernel(const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows) {
    const int64_t n = n_rows * ne0;
    for (int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t) gridDim.x * 256) {
        const int64_t r = e / ne0, i = e % ne0;
        const float bv = b[(r % b_rows) * ne0 + i];
        y[e] = OP == 0 ? a[e] + bv : a[e] * bv;
    }
}

__global__ void __launch_bounds__(256) silu_kernel(const float * x, float * y, int64_t n) {
    for (int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t) gridDim.x * 256) {
        const float v = x[e];
        y[e] = v / (1.0f + expf(-v));
    }
}

// y[c][m] = sum_k W[m][k] x[c][k].  One wave per weight row (4 rows per workgroup); activation columns go in tiles of CT held in
// registers, so the weights leave HBM once and later column tiles re-read them from cache.  f32 / f16 rows are read with 16-byte
// loads; q8_0 rows one 34-byte block per lane with 2-byte loads (the blocks are only 2-byte aligned).  Measured alternatives for
// q8_0 that lost (average per product in a TinyLlama-1.1B decode step, 8.8 us with direct activation loads; what helped afterwards is
// the ACTIVATION side: the chunk staged once per workgroup in LDS, transposed, see the q8_0 / q4_K branches): rows staged through LDS with coalesced dword
// loads 10.4 us; one row per workgroup split over its 4 waves 14.8 us; weights repacked to int8 qs[M][K] + half d[M][K/32] with q8_0-
// quantized activations and v_dot4: 7.4 us + 3.3 us for the quantize launch, 9.7 us with the quantization fused into the product.
// At these sizes (4-12 MB per product, 17-48 KB per CU) the products are one memory round trip plus the launch floor, not
// bandwidth-bound: the next step is fewer, larger launches (fused gate/up, persistent layer kernel), not a different inner loop.
constexpr int MM_CT = 8, Q4K_ROWS = 2;
// block_q4_K rows x one f32 activation column per pass; `wg` = index of the group of 4 Q4K_ROWS rows this workgroup owns
template <bool PK = false>   // PK: rows in the plane layout of pkv_wlayout.h
__device__ __forceinline__ void q4k_rows(const char * w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, size_t w_row_bytes,
                                         int64_t wg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

        // block_q4_K: half d, half dmin, 12 bytes of packed 6-bit (scale, min) pairs, 128 bytes of nibbles = 256 elements in 4 groups
        // of 64 (low nibbles = first 32 elements of the group, high nibbles = the next 32).  A lane owns one 64-element group of a
        // 4096-element chunk.  The chunk of the activation column is staged ONCE per workgroup in LDS, transposed (element-major,
        // row stride 65) so that both the coalesced staging writes and the per-lane reads are bank-conflict free; read straight
        // from global memory every lane's 64 floats are 256 bytes apart and each load instruction touches 64 cache lines.
        // ... and QUANTIZED while staged (int8 per 32-element sub-block, d = amax / 127 kept in f32 — finer than the q8_K blocks of 256
        // the reference's CPU backend quantizes activations to before a q4_K product, vec_dot_type of q4_K, ggml-cpu.c:7467-7640): the
        // nibbles then meet them in v_dot4_i32_i8, 4 elements per instruction instead of one convert + one FMA each.
        constexpr int KC = 4096, R = Q4K_ROWS;   // R weight rows per wave: the staged chunk serves 4 R rows, R x 3 loads in flight per lane
        __shared__ uint32_t sq[16 * 65];      // [dword of the group (4 int8)][group] : lane = group reads stride-1
        __shared__ float    sd[2 * 65], ss[2 * 65];   // per (half of the group, group): scale, scale * sum of the int8
        const int64_t m0 = (wg * 4 + wv) * R;
        for (int64_t c = 0; c < n_cols; ++c) {
            const float * xc = x + (size_t) c * K;
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.0f;
            for (int64_t k0 = 0; k0 < K; k0 += KC) {
                const int kc = (int) min((int64_t) KC, K - k0), ng = kc / 64;
                // this lane's slices of the weight rows are requested FIRST (they do not depend on the activations): their HBM latency
                // runs under the staging of the chunk.  (lanes past the end of a short chunk re-read the last group and drop it;
                // rows past M re-read the last row and are not stored)
                const int p = lane & 3;
                uint4 hdr[R], qa[R], qb[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned char * rowp = reinterpret_cast<const unsigned char *>(w) + (size_t) min(m0 + r, M - 1) * w_row_bytes;
                    if constexpr (PK) {
                        const size_t gpr = (size_t) (K >> 6), g = (size_t) (k0 >> 6) + (size_t) min(lane, ng - 1);
                        hdr[r] = *reinterpret_cast<const uint4 *>(rowp + 32 * gpr + 16 * (g >> 2));
                        qa[r]  = *reinterpret_cast<const uint4 *>(rowp + 16 * g);
                        qb[r]  = *reinterpret_cast<const uint4 *>(rowp + 16 * gpr + 16 * g);
                    } else {
                    const unsigned char * bp = rowp + (size_t) ((k0 >> 8) + (min(lane, ng - 1) >> 2)) * 144;
                    hdr[r] = *reinterpret_cast<const uint4 *>(bp);   // d, dmin and the 12 scale bytes in ONE 16-byte load
                    qa[r]  = *reinterpret_cast<const uint4 *>(bp + 16 + 32 * p);
                    qb[r]  = *reinterpret_cast<const uint4 *>(bp + 16 + 32 * p + 16);
                    }
                }
                // the chunk's activations: all of a thread's float4 are requested before any is used (a loop over them would pay
                // the L2 latency once per iteration: kc is not a compile-time constant)
                float4 xv[KC / 1024];
#pragma unroll
                for (int u = 0; u < KC / 1024; ++u) {
                    const int i4 = threadIdx.x + 256 * u;
                    xv[u] = i4 < kc / 4 ? *reinterpret_cast<const float4 *>(xc + k0 + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __syncthreads();                                   // the previous chunk is no longer read
#pragma unroll
                for (int u = 0; u < KC / 1024; ++u) {                  // 4 elements per thread, 8 threads per 32-element sub-block
                    const int i4 = threadIdx.x + 256 * u;
                    const float4 v = xv[u];
                    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                    amax = fmaxf(amax, __shfl_xor(amax, 1)); amax = fmaxf(amax, __shfl_xor(amax, 2)); amax = fmaxf(amax, __shfl_xor(amax, 4));
                    const float dq = amax / 127.0f, id = amax != 0.0f ? 127.0f / amax : 0.0f;
                    const int q0 = (int) __builtin_rintf(v.x * id), q1 = (int) __builtin_rintf(v.y * id), q2 = (int) __builtin_rintf(v.z * id),
                              q3 = (int) __builtin_rintf(v.w * id);
                    int qs = q0 + q1 + q2 + q3;
                    qs += __shfl_xor(qs, 1); qs += __shfl_xor(qs, 2); qs += __shfl_xor(qs, 4);
                    const int grp = i4 >> 4, wi = i4 & 15;         // group of 64 elements, dword inside it
                    if (i4 < kc / 4) {
                        sq[wi * 65 + grp] = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
                        if ((i4 & 7) == 0) { sd[(wi >> 3) * 65 + grp] = dq; ss[(wi >> 3) * 65 + grp] = dq * (float) qs; }
                    }
                }
                __syncthreads();
                if (lane < ng) {
                    uint32_t xq[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) xq[i] = sq[i * 65 + lane];
                    const float dx0 = sd[lane], dx1 = sd[65 + lane], sx0 = ss[lane], sx1 = ss[65 + lane];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint16_t dlo = (uint16_t) (hdr[r].x & 0xFFFF), dhi = (uint16_t) (hdr[r].x >> 16);
                        const float d = (float) *reinterpret_cast<const _Float16 *>(&dlo), dmin = (float) *reinterpret_cast<const _Float16 *>(&dhi);
                        // get_scale_min_k4 (/ref/ggml/src/ggml-quants.c:631-638) with shifts only: picking one of hdr.y / .z / .w by a runtime index
                        // sent the header through scratch memory (r01: 368 bytes of private segment per lane)
                        auto scale_min = [&](int j, int & s_, int & m_) {
                            const int sh = 8 * (j & 3);
                            const int b0 = (int) ((hdr[r].y >> sh) & 0xFF), b1 = (int) ((hdr[r].z >> sh) & 0xFF), b2 = (int) ((hdr[r].w >> sh) & 0xFF);
                            s_ = j < 4 ? (b0 & 63) : ((b2 & 0xF) | ((b0 >> 6) << 4));
                            m_ = j < 4 ? (b1 & 63) : ((b2 >> 4) | ((b1 >> 6) << 4));
                        };
                        int s1, m1, s2, m2;
                        scale_min(2 * p, s1, m1);
                        scale_min(2 * p + 1, s2, m2);
                        const uint32_t qw[8] = {qa[r].x, qa[r].y, qa[r].z, qa[r].w, qb[r].x, qb[r].y, qb[r].z, qb[r].w};
                        int ilo = 0, ihi = 0;
#pragma unroll
                        for (int w4 = 0; w4 < 8; ++w4) {           // byte b of dword w4: low nibble = element 4 w4 + b, high = 32 + 4 w4 + b
                            ilo = __builtin_amdgcn_sdot4((int) (qw[w4] & 0x0F0F0F0Fu), (int) xq[w4], ilo, false);
                            ihi = __builtin_amdgcn_sdot4((int) ((qw[w4] >> 4) & 0x0F0F0F0Fu), (int) xq[8 + w4], ihi, false);
                        }
                        // sum (d*sc*q - dmin*m) x  with x ~ dx * xq :  d*sc*dx*sum(q xq) - dmin*m*dx*sum(xq)
                        acc[r] += (d * (float) s1 * dx0 * (float) ilo - dmin * (float) m1 * sx0) + (d * (float) s2 * dx1 * (float) ihi - dmin * (float) m2 * sx1);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float v = wave_sum_f(acc[r]);
                if (lane == 0 && m0 + r < M) y[(size_t) c * M + m0 + r] = v;
            }
        }
        }

// block_q8_0 rows x f32 activation columns; `wg` = index of the group of 4 rows (one per wave) this workgroup owns
__device__ __forceinline__ void q8_rows(const char * w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, size_t w_row_bytes, int64_t wg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t m = wg * 4 + wv;
    const char * wr = w + (size_t) min(m, M - 1) * w_row_bytes;            // surplus waves redo the last row and do not store

        // same scheme for block_q8_0 rows: a lane owns one 32-element block of a 2048-element chunk, the activation chunk is staged
        // once per workgroup in LDS, transposed (row stride 65)
        constexpr int KC = 2048;
        __shared__ float sx[32 * 65];
        for (int64_t c = 0; c < n_cols; ++c) {
            const float * xc = x + (size_t) c * K;
            float acc = 0.0f;
            for (int64_t k0 = 0; k0 < K; k0 += KC) {
                const int kc = (int) min((int64_t) KC, K - k0), nb = kc / PKV_QK;
                __syncthreads();
                for (int i = threadIdx.x; i < kc; i += 256) sx[(i & 31) * 65 + (i >> 5)] = xc[k0 + i];
                __syncthreads();
                if (lane < nb) {
                    const uint16_t * bp = reinterpret_cast<const uint16_t *>(wr + (size_t) (k0 / PKV_QK + lane) * 34);
                    const uint16_t db = bp[0];
                    const float d = (float) *reinterpret_cast<const _Float16 *>(&db);
                    float sdot = 0.0f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint16_t two = bp[1 + j];
                        sdot = fmaf((float) (int) (signed char) (two & 0xFF), sx[(2 * j) * 65 + lane], sdot);
                        sdot = fmaf((float) (int) (signed char) (two >> 8), sx[(2 * j + 1) * 65 + lane], sdot);
                    }
                    acc = fmaf(d, sdot, acc);
                }
            }
            const float v = wave_sum_f(acc);
            if (lane == 0 && m < M) y[(size_t) c * M + m] = v;
        }
        }

// up to PKV_MUL_MAT_MAX products that share the activation column(s) and K (the q / k / v projections of a layer, gate / up of its
// FFN) in ONE launch: at decode each product is a ~5 us launch-and-latency floor plus its bytes, so three launches cost three floors
struct MulMatMulti { const char * w[PKV_MUL_MAT_MAX]; float * y[PKV_MUL_MAT_MAX]; int64_t M[PKV_MUL_MAT_MAX]; int wg_end[PKV_MUL_MAT_MAX]; int n; };
template <int WT>
__global__ void __launch_bounds__(256) mul_mat_multi_kernel(MulMatMulti a, const float * x, int64_t K, int64_t n_cols, size_t w_row_bytes) {
    int i = 0;
    while (i + 1 < a.n && (int) blockIdx.x >= a.wg_end[i]) ++i;
    const int64_t wg = (int64_t) blockIdx.x - (i ? a.wg_end[i - 1] : 0);
    if constexpr (WT == PKV_TYPE_Q4_K) q4k_rows<false>(a.w[i], x, a.y[i], a.M[i], K, n_cols, w_row_bytes, wg);
    else if constexpr (WT == (PKV_TYPE_Q4_K | PKV_TYPE_PACKED)) q4k_rows<true>(a.w[i], x, a.y[i], a.M[i], K, n_cols, w_row_bytes, wg);
    else                               q8_rows(a.w[i], x, a.y[i], a.M[i], K, n_cols, w_row_bytes, wg);
}

template <int WT>
__global__ void __launch_bounds__(256) mul_mat_kernel(const char * w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols,
                                                      size_t w_row_bytes) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t m = (int64_t) blockIdx.x * 4 + wv;
    const char * wr = w + (size_t) min(m, M - 1) * w_row_bytes;            // surplus waves redo the last row and do not store
    if constexpr (WT == PKV_TYPE_Q4_K) { q4k_rows<false>(w, x, y, M, K, n_cols, w_row_bytes, (int64_t) blockIdx.x); return; }
    if constexpr (WT == (PKV_TYPE_Q4_K | PKV_TYPE_PACKED)) { q4k_rows<true>(w, x, y, M, K, n_cols, w_row_bytes, (int64_t) blockIdx.x); return; }
    if constexpr (WT == PKV_TYPE_Q8_0) { q8_rows(w, x, y, M, K, n_cols, w_row_bytes, (int64_t) blockIdx.x); return; }
    for (int64_t c0 = 0; c0 < n_cols; c0 += MM_CT) {
        const int nc = (int) min((int64_t) MM_CT, n_cols - c0);
        float acc[MM_CT];
#pragma unroll
        for (int c = 0; c < MM_CT; ++c) acc[c] = 0.0f;
        if constexpr (WT == PKV_TYPE_F32 || WT == PKV_TYPE_F16) {
            constexpr int E = WT == PKV_TYPE_F32 ? 4 : 8;                  // elements per 16-byte load
            for (int64_t k = (int64_t) lane * E; k < K; k += 64 * E) {
                float wf[E];
                if constexpr (WT == PKV_TYPE_F32) {
                    const float4 a = *reinterpret_cast<const float4 *>(wr + k * 4);
                    wf[0] = a.x; wf[1] = a.y; wf[2] = a.z; wf[3] = a.w;
                } else {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(wr + k * 2);
                    const _Float16 * h = reinterpret_cast<const _Float16 *>(&raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) wf[j] = (float) h[j];
                }
#pragma unroll
                for (int c = 0; c < MM_CT; ++c) {
                    if (c < nc) {
                        const float * xc = x + (size_t) (c0 + c) * K + k;
#pragma unroll
                        for (int j = 0; j < E; j += 4) {
                            const float4 b = *reinterpret_cast<const float4 *>(xc + j);
                            acc[c] = fmaf(wf[j], b.x, acc[c]); acc[c] = fmaf(wf[j + 1], b.y, acc[c]);
                            acc[c] = fmaf(wf[j + 2], b.z, acc[c]); acc[c] = fmaf(wf[j + 3], b.w, acc[c]);
                        }
                    }
                }
            }
        } else if constexpr ((WT & ~PKV_TYPE_PACKED) == PKV_TYPE_Q4_K) {
            // handled below (activations staged through LDS): nothing in the generic column-tile loop
        } else {
            for (int64_t blk = lane; blk < K / PKV_QK; blk += 64) {
                const uint16_t * bp = reinterpret_cast<const uint16_t *>(wr + (size_t) blk * 34);
                const uint16_t db = bp[0];
                const float d = (float) *reinterpret_cast<const _Float16 *>(&db);
                float q[PKV_QK];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint16_t two = bp[1 + j];
                    q[2 * j]     = (float) (int) (signed char) (two & 0xFF);
                    q[2 * j + 1] = (float) (int) (signed char) (two >> 8);
                }
#pragma unroll
                for (int c = 0; c < MM_CT; ++c) {
                    if (c < nc) {
                        const float * xc = x + (size_t) (c0 + c) * K + blk * PKV_QK;
                        float s = 0.0f;
#pragma unroll
                        for (int j = 0; j < PKV_QK; j += 4) {
                            const float4 b = *reinterpret_cast<const float4 *>(xc + j);
                            s = fmaf(q[j], b.x, s); s = fmaf(q[j + 1], b.y, s); s = fmaf(q[j + 2], b.z, s); s = fmaf(q[j + 3], b.w, s);
                        }
                        acc[c] = fmaf(d, s, acc[c]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MM_CT; ++c) {
            const float v = wave_sum_f(acc[c]);
            if (lane == 0 && c < nc && m < M) y[(size_t) (c0 + c) * M + m] = v;
        }
    }
}

int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int) (g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace

int launch_rms_norm(const float * x, float * y, int64_t n_rows, int64_t ne0, float eps, hipStream_t st) {
    if (n_rows == 0) return PKV_OK;
    hipLaunchKernelGGL(rms_norm_mul_kernel, dim3((unsigned) n_rows), dim3(256), 0, st, x, (const float *) nullptr, (float *) nullptr, (const float *) nullptr, (float *) nullptr, y, ne0, eps, (char *) nullptr);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_rms_norm_mul(const float * x, const float * res, float * x_out, const float * w, float * y_norm, float * y, int64_t n_rows, int64_t ne0, float eps,
                        hipStream_t st) {
    if (n_rows == 0) return PKV_OK;
    hipLaunchKernelGGL(rms_norm_mul_kernel, dim3((unsigned) n_rows), dim3(256), 0, st, x, res, x_out, w, y_norm, y, ne0, eps, (char *) nullptr);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_rms_norm_mul_q(const float * x, const float * res, float * x_out, const float * w, float * y, void * img, int64_t ne0, float eps, hipStream_t st) {
    PKV_REQUIRE(ne0 > 0 && ne0 % 256 == 0 && ne0 <= 256 * 4 * RMS_MAX_V4, "pkv_rms_norm_mul_q: row length must be a multiple of 256 up to %d", 256 * 4 * RMS_MAX_V4);
    hipLaunchKernelGGL(rms_norm_mul_kernel, dim3(1), dim3(256), 0, st, x, res, x_out, w, (float *) nullptr, y, ne0, eps, (char *) img);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_soft_max(const float * x, const void * mask, int mask_f16, float * y, int64_t ne00, int64_t ne01, int64_t ne02, int64_t ne03, float scale, float max_bias,
                    hipStream_t st) {
    const int64_t nr = ne01 * ne02 * ne03;
    if (nr == 0 || ne00 == 0) return PKV_OK;
    const uint32_t n_head_log2 = 1u << (uint32_t) floorf(log2f((float) ne02));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    if (mask && mask_f16) hipLaunchKernelGGL(soft_max_kernel<_Float16>, dim3((unsigned) nr), dim3(256), 0, st, x, (const _Float16 *) mask, y, ne00, ne01, ne02, scale, max_bias, m0, m1, n_head_log2);
    else                  hipLaunchKernelGGL(soft_max_kernel<float>, dim3((unsigned) nr), dim3(256), 0, st, x, (const float *) mask, y, ne00, ne01, ne02, scale, max_bias, m0, m1, n_head_log2);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_cpy_strided(const void * src, int src_type, const int64_t * ne_s, const int64_t * nb_s, void * dst, int dst_type, const int64_t * ne_d, const int64_t * nb_d, hipStream_t st) {
    Strided4 s, d;
    int64_t n = 1, nd = 1;
    for (int i = 0; i < 4; ++i) { s.ne[i] = ne_s[i]; s.nb[i] = nb_s[i]; d.ne[i] = ne_d[i]; d.nb[i] = nb_d[i]; n *= ne_s[i]; nd *= ne_d[i]; }
    PKV_REQUIRE(n == nd, "pkv_cpy_strided: element counts differ");
    if (n == 0) return PKV_OK;
    const unsigned grid = (unsigned) std::min<int64_t>((n + 255) / 256, 65535);
#define PKV_CPY(ST, DT) hipLaunchKernelGGL((cpy_strided_kernel<ST, DT>), dim3(grid), dim3(256), 0, st, (const char *) src, (char *) dst, n, s, d)
    if (src_type == PKV_TYPE_F32 && dst_type == PKV_TYPE_F32)        PKV_CPY(float, float);
    else if (src_type == PKV_TYPE_F32 && dst_type == PKV_TYPE_F16)   PKV_CPY(float, _Float16);
    else if (src_type == PKV_TYPE_F16 && dst_type == PKV_TYPE_F16)   PKV_CPY(_Float16, _Float16);
    else if (src_type == PKV_TYPE_F16 && dst_type == PKV_TYPE_F32)   PKV_CPY(_Float16, float);
    else if (src_type == PKV_TYPE_F32 && dst_type == PKV_TYPE_BF16)  PKV_CPY(float, bf16_t);     // the KV store of a bf16 cache (-ctk / -ctv bf16)
    else if (src_type == PKV_TYPE_F16 && dst_type == PKV_TYPE_BF16)  PKV_CPY(_Float16, bf16_t);
    else if (src_type == PKV_TYPE_BF16 && dst_type == PKV_TYPE_BF16) PKV_CPY(bf16_t, bf16_t);
    else if (src_type == PKV_TYPE_BF16 && dst_type == PKV_TYPE_F32)  PKV_CPY(bf16_t, float);
    else if (src_type == PKV_TYPE_BF16 && dst_type == PKV_TYPE_F16)  PKV_CPY(bf16_t, _Float16);
#undef PKV_CPY
    else { set_error("pkv_cpy_strided: types %d -> %d not supported (f32 / f16 / bf16)", src_type, dst_type); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_mul_mat_strided(const void * w, int type_w, const int64_t * ne_w, const int64_t * nb_w, const float * x, const int64_t * ne_x, const int64_t * nb_x, float * y,
                           hipStream_t st) {
    Strided4 sw, sx;
    for (int i = 0; i < 4; ++i) { sw.ne[i] = ne_w[i]; sw.nb[i] = nb_w[i]; sx.ne[i] = ne_x[i]; sx.nb[i] = nb_x[i]; }
    PKV_REQUIRE(ne_w[0] == ne_x[0] && ne_w[2] > 0 && ne_w[3] > 0 && ne_x[2] % ne_w[2] == 0 && ne_x[3] % ne_w[3] == 0, "pkv_mul_mat_strided: shapes do not broadcast");
    PKV_REQUIRE(nb_x[0] == 4 && nb_w[0] == (type_w == PKV_TYPE_F16 ? 2 : 4), "pkv_mul_mat_strided: unit stride along k required");
    const int64_t M = ne_w[1], N = ne_x[1], nb = ne_x[2] * ne_x[3];
    if (M == 0 || N == 0 || nb == 0) return PKV_OK;
    PKV_REQUIRE(nb <= 65535, "pkv_mul_mat_strided: too many batches");
    const dim3 grid((unsigned) ((M + 3) / 4), (unsigned) std::min<int64_t>((N + 3) / 4, 64), (unsigned) nb);
    if (type_w == PKV_TYPE_F16)      hipLaunchKernelGGL(mul_mat_strided_kernel<_Float16>, grid, dim3(256), 0, st, (const char *) w, (const char *) x, y, sw, sx, ne_x[2] / ne_w[2], ne_x[3] / ne_w[3]);
    else if (type_w == PKV_TYPE_F32) hipLaunchKernelGGL(mul_mat_strided_kernel<float>, grid, dim3(256), 0, st, (const char *) w, (const char *) x, y, sw, sx, ne_x[2] / ne_w[2], ne_x[3] / ne_w[3]);
    else { set_error("pkv_mul_mat_strided: weight type %d not supported (f16, f32)", type_w); return PKV_ERR_UNSUPPORTED; }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_silu_mul(const float * g, const float * u, float * y_silu, float * y, int64_t n, hipStream_t st) {
    if (n == 0) return PKV_OK;
    hipLaunchKernelGGL(silu_mul_kernel, dim3(grid_for(n)), dim3(256), 0, st, g, u, y_silu, y, n);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_binary(int op, const float * a, const float * b, float * y, int64_t n_rows, int64_t ne0, int64_t b_rows, hipStream_t st) {
    if (n_rows * ne0 == 0) return PKV_OK;
    if (op == 0) hipLaunchKernelGGL(binary_kernel<0>, dim3(grid_for(n_rows * ne0)), dim3(256), 0, st, a, b, y, n_rows, ne0, b_rows);
    else         hipLaunchKernelGGL(binary_kernel<1>, dim3(grid_for(n_rows * ne0)), dim3(256), 0, st, a, b, y, n_rows, ne0, b_rows);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
int launch_silu(const float * x, float * y, int64_t n, hipStream_t st) {
    if (n == 0) return PKV_OK;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
// one column of K-quant weights: the decode-step product of pkv_gemv.hip (the column quantized to q8_K in its prologue)
static bool gemv_shape(int type_w, int64_t M, int64_t K) { type_w &= ~PKV_TYPE_PACKED; return (type_w == PKV_TYPE_Q4_K || type_w == PKV_TYPE_Q5_K || type_w == PKV_TYPE_Q6_K) && K % 256 == 0 && K <= 49152 && M % 2 == 0; }
constexpr int64_t GEMM_MIN_COLS = 16;   // from here on a weight tile read once per 128 columns beats a stream per column
int launch_mul_mat(const void * w, int type_w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, hipStream_t st) {
    if (M == 0 || n_cols == 0) return PKV_OK;
    if (n_cols == 1 && gemv_shape(type_w, M, K)) {
        pkv_gemv g = {};
        g.n = 1; g.type_w[0] = type_w; g.w[0] = w; g.y[0] = y; g.M[0] = M; g.K = K; g.x = x;
        return launch_gemv(g, st);
    }
    const int bt = type_w & ~PKV_TYPE_PACKED;                        // plane-layout rows (pkv_wlayout.h) keep the base type's row size
    if ((n_cols >= GEMM_MIN_COLS || bt == PKV_TYPE_Q6_K || bt == PKV_TYPE_Q5_K) && gemm_supported(type_w, M, K)) return launch_gemm(w, type_w, x, y, M, K, n_cols, st);
    const int rows_wg = bt == PKV_TYPE_Q4_K ? 4 * Q4K_ROWS : 4;
    const dim3 grid((unsigned) ((M + rows_wg - 1) / rows_wg));
    const size_t rb = row_size(bt, K);
    switch (type_w) {
        case PKV_TYPE_Q4_K | PKV_TYPE_PACKED: hipLaunchKernelGGL(mul_mat_kernel<(PKV_TYPE_Q4_K | PKV_TYPE_PACKED)>, grid, dim3(256), 0, st, (const char *) w, x, y, M, K, n_cols, rb); break;
        case PKV_TYPE_F32:  hipLaunchKernelGGL(mul_mat_kernel<PKV_TYPE_F32>, grid, dim3(256), 0, st, (const char *) w, x, y, M, K, n_cols, rb); break;
        case PKV_TYPE_F16:  hipLaunchKernelGGL(mul_mat_kernel<PKV_TYPE_F16>, grid, dim3(256), 0, st, (const char *) w, x, y, M, K, n_cols, rb); break;
        case PKV_TYPE_Q8_0: hipLaunchKernelGGL(mul_mat_kernel<PKV_TYPE_Q8_0>, grid, dim3(256), 0, st, (const char *) w, x, y, M, K, n_cols, rb); break;
        case PKV_TYPE_Q4_K: hipLaunchKernelGGL(mul_mat_kernel<PKV_TYPE_Q4_K>, grid, dim3(256), 0, st, (const char *) w, x, y, M, K, n_cols, rb); break;
        default: set_error("pkv_mul_mat: weight type %d not supported (f32, f16, q8_0, q4_K)", type_w); return PKV_ERR_UNSUPPORTED;
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_mul_mat_multi(int n, const void * const * w, int type_w, const float * x, float * const * y, const int64_t * M, int64_t K, int64_t n_cols,
                         hipStream_t st) {
    if (n_cols == 1 && n <= PKV_GEMV_MAX) {                           // decode: the products that share the column in ONE pkv_gemv.hip launch
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && gemv_shape(type_w, M[i], K) && M[i] > 0;
        if (ok) {
            pkv_gemv g = {};
            g.n = n; g.K = K; g.x = x;
            for (int i = 0; i < n; ++i) { g.type_w[i] = type_w; g.w[i] = w[i]; g.y[i] = y[i]; g.M[i] = M[i]; }
            return launch_gemv(g, st);
        }
    }
    if (n_cols >= GEMM_MIN_COLS && gemm_supported(type_w, 1, K) && n <= 4) return launch_gemm_multi(n, w, type_w, x, y, M, K, n_cols, st);   // prompt ubatch: ONE MFMA launch
    const int bt = type_w & ~PKV_TYPE_PACKED;
    if ((n_cols >= GEMM_MIN_COLS && gemm_supported(type_w, 1, K)) || (bt != PKV_TYPE_Q4_K && bt != PKV_TYPE_Q8_0)) {       // one launch each
        for (int i = 0; i < n; ++i) if (int rc = launch_mul_mat(w[i], type_w, x, y[i], M[i], K, n_cols, st)) return rc;
        return PKV_OK;
    }
    if (n_cols == 0) return PKV_OK;
    MulMatMulti a = {};
    int tot = 0;
    for (int i = 0; i < n; ++i) {
        if (M[i] == 0) continue;
        a.w[a.n] = (const char *) w[i]; a.y[a.n] = y[i]; a.M[a.n] = M[i];
        const int rows_wg = bt == PKV_TYPE_Q4_K ? 4 * Q4K_ROWS : 4;
        tot += (int) ((M[i] + rows_wg - 1) / rows_wg);
        a.wg_end[a.n++] = tot;
    }
    if (a.n == 0) return PKV_OK;
    if (type_w == PKV_TYPE_Q4_K) hipLaunchKernelGGL(mul_mat_multi_kernel<PKV_TYPE_Q4_K>, dim3((unsigned) tot), dim3(256), 0, st, a, x, K, n_cols, row_size(PKV_TYPE_Q4_K, K));
    else if (bt == PKV_TYPE_Q4_K) hipLaunchKernelGGL(mul_mat_multi_kernel<(PKV_TYPE_Q4_K | PKV_TYPE_PACKED)>, dim3((unsigned) tot), dim3(256), 0, st, a, x, K, n_cols, row_size(PKV_TYPE_Q4_K, K));
    else                         hipLaunchKernelGGL(mul_mat_multi_kernel<PKV_TYPE_Q8_0>, dim3((unsigned) tot), dim3(256), 0, st, a, x, K, n_cols, row_size(PKV_TYPE_Q8_0, K));
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

}  // namespace pkv
