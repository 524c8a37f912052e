// pkv_gemv.hip — decode-step weight products (one activation column) for K-quant weights, gfx950.
//
// the mat-vec half of a llama decoder layer (/ref/src/llama-model.cpp:4070-4160: wq / wk / wv, wo, ffn gate / up /
// down; lm_head :4186).  The reference's CPU backend quantizes the activation column to q8_K and runs integer dots
// (ggml_compute_forward_mul_mat, /ref/ggml/src/ggml-cpu/ggml-cpu.c:7467-7640, vec_dot_q4_K_q8_K / vec_dot_q6_K_q8_K); here the column is
// quantized ONCE per product group to the same q8_K integers by pkv_quantize_act — or by the kernel that produced it — and every
// workgroup copies the 1 byte / element image into LDS, then STREAMS its rows:
//   * block_q4_K (144 B / 256 elements, /ref/ggml/src/ggml-common.h:279-290) and block_q6_K (210 B, :309-319) rows, read exactly once;
//   * a lane owns 64 elements (q4_K: one 64-element group = 32 bytes of nibbles; q6_K: four runs of 16 with their own scales) of one of
//     the TWO rows a wave works on at a time, so the 64 lanes cover 2 x K / 64 groups in passes of 64 whatever K is;
//   * products that share the column run in ONE launch (wq + wk + wv; gate + up), the residual add of wo / ffn_down and the
//     SiLU(gate) * up of the FFN are epilogues — at decode every launch boundary is ~1.8 us against 2-20 us of streaming.
// r01's products (pkv_layer_ops.hip) re-staged and re-quantized the column in every 8-row workgroup: 16 KB of activations per 18 KB of
// weights, 0.18-0.39 of HBM peak.
#include "pkv_common.h"

namespace pkv {
namespace {

__device__ __forceinline__ float wsum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float h2f(uint16_t b) { return (float) *reinterpret_cast<const _Float16 *>(&b); }

// ---- activation column -> q8_K, the reference's own operand of a q4_K / q6_K product ---------------------------------------------
// quantize_row_q8_K_ref (/ref/ggml/src/ggml-quants.c:2479-2516; the x86 CPU backend runs exactly this, ggml-cpu-quants.c:1746): per 256
// elements iscale = -127 / max (max = the signed value of largest magnitude), q = nearest_int(iscale * x), d = 1 / iscale, plus the sums of
// the quants.  Image: [K int8 q][K/256 f32 d][K/32 i32 sum of q per 32].  The integer dots of pkv_gemv_q then see the reference's integers:
// a product differs from the CPU backend's only by the order of a handful of f32 additions.  (Which of two elements of equal magnitude
// and opposite sign supplies `max` only flips the sign of d and of every q together: the products q * d are the same bits.)
// One wave per 256-element block, 4 elements per lane.
__device__ __forceinline__ void quant_block_q8k(const float4 v, int lane, uint32_t & packed, float & d, int & bsum32) {
    float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    float mx = fabsf(v.x) == am ? v.x : (fabsf(v.y) == am ? v.y : (fabsf(v.z) == am ? v.z : v.w));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float oa = __shfl_xor(am, o), om = __shfl_xor(mx, o);
        if (oa > am || (oa == am && (lane & o))) { am = oa; mx = om; }     // equal magnitudes: every lane settles on the lower lane's value
    }
    const float iscale = am != 0.0f ? -127.0f / mx : 0.0f;
    const int q0 = (int) __builtin_rintf(iscale * v.x), q1 = (int) __builtin_rintf(iscale * v.y), q2 = (int) __builtin_rintf(iscale * v.z),
              q3 = (int) __builtin_rintf(iscale * v.w);
    int qs = q0 + q1 + q2 + q3;
    qs += __shfl_xor(qs, 1); qs += __shfl_xor(qs, 2); qs += __shfl_xor(qs, 4);
    packed = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
    d = am != 0.0f ? 1.0f / iscale : 0.0f;
    bsum32 = qs;
}
__global__ void __launch_bounds__(256) quant_act_kernel(const float * x, char * img, int64_t K) {
    const int64_t i4 = (int64_t) blockIdx.x * 256 + threadIdx.x;            // float4 index; K % 256 == 0: a wave never straddles the end
    if (i4 * 4 >= K) return;
    uint32_t packed; float d; int bs;
    quant_block_q8k(reinterpret_cast<const float4 *>(x)[i4], threadIdx.x & 63, packed, d, bs);
    reinterpret_cast<uint32_t *>(img)[i4] = packed;
    if ((i4 & 63) == 0) reinterpret_cast<float *>(img + K)[i4 >> 6] = d;
    if ((i4 & 7) == 0) reinterpret_cast<int *>(img + K + (K >> 8) * 4)[i4 >> 3] = bs;
}

struct GemvArgs {
    const char * w[PKV_GEMV_MAX];
    float *      y[PKV_GEMV_MAX];
    int64_t      M[PKV_GEMV_MAX];
    int          type[PKV_GEMV_MAX];
    int          rp_end[PKV_GEMV_MAX];   // cumulative row pairs
    int          n, silu_mul;
    const float * residual;
    const char * img;
    int64_t      K;
    int          gpr, gs;                // 64-element groups per row, LDS row stride (odd)
};

// dynamic LDS: sq[16][gs] dwords (dword wi of group g at wi * gs + g), sd[2][gs] (scale of the group's first / second 32), ss[2][gs]
extern __shared__ __attribute__((aligned(16))) uint32_t g_lds[];

// one 64-element group of a block_q4_K row against the staged column; lane-private
__device__ __forceinline__ float dot_q4k(const uint4 & hdr, const uint4 & qa, const uint4 & qb, int p, const uint32_t * sq, const float * sd, const float * ss, int gs, int g) {
    uint32_t xq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xq[i] = sq[i * gs + g];
    const float dx = sd[g];                                                     // the column's scale for this 256-block
    const int   b0 = reinterpret_cast<const int *>(ss)[g], b1 = reinterpret_cast<const int *>(ss)[gs + g];   // sums of its quants, per 32
    const float d = h2f((uint16_t) (hdr.x & 0xFFFF)) * dx, dmin = h2f((uint16_t) (hdr.x >> 16)) * dx;       // as ggml_vec_dot_q4_K_q8_K forms them
    auto sc = [&](int k) -> int { const uint32_t wd_ = k < 4 ? hdr.y : (k < 8 ? hdr.z : hdr.w); return (int) ((wd_ >> (8 * (k & 3))) & 0xFF); };
    auto scale_min = [&](int j, int & s_, int & m_) {   // get_scale_min_k4 (/ref/ggml/src/ggml-quants.c:631-638)
        if (j < 4) { s_ = sc(j) & 63; m_ = sc(j + 4) & 63; }
        else       { s_ = (sc(j + 4) & 0xF) | ((sc(j - 4) >> 6) << 4); m_ = (sc(j + 4) >> 4) | ((sc(j) >> 6) << 4); }
    };
    int s1, m1, s2, m2;
    scale_min(2 * p, s1, m1);
    scale_min(2 * p + 1, s2, m2);
    const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    int ilo = 0, ihi = 0;
#pragma unroll
    for (int w4 = 0; w4 < 8; ++w4) {           // byte b of dword w4: low nibble = element 4 w4 + b, high = 32 + 4 w4 + b
        ilo = __builtin_amdgcn_sdot4((int) (qw[w4] & 0x0F0F0F0Fu), (int) xq[w4], ilo, false);
        ihi = __builtin_amdgcn_sdot4((int) ((qw[w4] >> 4) & 0x0F0F0F0Fu), (int) xq[8 + w4], ihi, false);
    }
    // the reference's integers (ggml-cpu-quants.c:6511: sum_j scale_j * (q4 . q8)_j and sum_j min_j * bsum_j), then ONE pair of f32 products
    return d * (float) (s1 * ilo + s2 * ihi) - dmin * (float) (m1 * b0 + m2 * b1);
}

// block_q6_K (210 bytes, 2-byte aligned): quarter `qt` = (n, is) of block -> elements n*128 + 32 j + 16 is + [0, 16), j = 0..3, scale index
// 8 n + is + 2 j (dequantize_row_q6_K, /ref/ggml/src/ggml-quants.c:1690-1719).  Loads are 2-byte pieces assembled in registers (odd blocks
// sit 2 bytes off a dword boundary).
__device__ __forceinline__ uint32_t ld_u32_b2(const unsigned char * p) {
    const uint16_t * h = reinterpret_cast<const uint16_t *>(p);
    return (uint32_t) h[0] | ((uint32_t) h[1] << 16);
}
__device__ __forceinline__ float dot_q6k(const unsigned char * blk, int qt, const uint32_t * sq, const float * sd, int gs, int g64) {
    const int n = qt >> 1, is = qt & 1;
    const unsigned char * ql = blk + 64 * n + 16 * is;          // ql[l], l = 16 is .. 16 is + 15; ql[l + 32] 32 bytes on
    const unsigned char * qh = blk + 128 + 32 * n + 16 * is;
    const signed char *   sc = reinterpret_cast<const signed char *>(blk + 192) + 8 * n + is;
    const float d = h2f(*reinterpret_cast<const uint16_t *>(blk + 208));
    uint32_t la[4], lb[4], hh[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { la[w] = ld_u32_b2(ql + 4 * w); lb[w] = ld_u32_b2(ql + 32 + 4 * w); hh[w] = ld_u32_b2(qh + 4 * w); }
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // the run's activations: elements n*128 + 32 j + 16 is + [0, 16) = 64-group (2 n + j / 2), dwords 8 (j & 1) + 4 is + [0, 4)
        const int g = g64 + (j >> 1), wi0 = 8 * (j & 1) + 4 * is;
        int idot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t lo = ((j & 1) ? lb[w] : la[w]) >> (4 * (j >> 1));
            const uint32_t u  = (lo & 0x0F0F0F0Fu) | (((hh[w] >> (2 * j)) & 0x03030303u) << 4);      // 6-bit quants 0..63
            const uint32_t v  = ((u | 0x80808080u) - 0x20202020u) ^ 0x80808080u;                     // per byte: q - 32 as int8
            idot = __builtin_amdgcn_sdot4((int) v, (int) sq[(wi0 + w) * gs + g], idot, false);
        }
        acc += (int) sc[2 * j] * idot;
    }
    return d * sd[g64] * (float) acc;                           // ggml_vec_dot_q6_K_q8_K: d = x.d * y.d, sum_is scale * (q6 . q8)
}

__global__ void __launch_bounds__(256) gemv_kernel(const GemvArgs a) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int gs = a.gs, gpr = a.gpr;
    uint32_t * sq = g_lds;
    float *    sd = reinterpret_cast<float *>(g_lds + 16 * gs);
    float *    ss = sd + 2 * gs;
    // stage the column image: K / 4 dwords of int8, K / 32 scales and scaled sums
    for (int i = threadIdx.x; i < (int) (a.K >> 2); i += 256) sq[(i & 15) * gs + (i >> 4)] = reinterpret_cast<const uint32_t *>(a.img)[i];
    const float * dd = reinterpret_cast<const float *>(a.img + a.K);
    const int *   bb = reinterpret_cast<const int *>(a.img + a.K + (a.K >> 8) * 4);
    for (int g = threadIdx.x; g < gpr; g += 256) sd[g] = dd[g >> 2];                                  // the 256-block's scale, once per 64-group
    for (int b = threadIdx.x; b < (int) (a.K >> 5); b += 256) reinterpret_cast<int *>(ss)[(b & 1) * gs + (b >> 1)] = bb[b];
    __syncthreads();

    const int total_rp = a.rp_end[a.n - 1];
    for (int rp = (int) blockIdx.x * 4 + wv; rp < total_rp; rp += (int) gridDim.x * 4) {
        int mi = 0;
        while (mi + 1 < a.n && rp >= a.rp_end[mi]) ++mi;
        const int rpl = rp - (mi ? a.rp_end[mi - 1] : 0);
        // the two rows of the pair: consecutive rows of one matrix, or (gate row m, up row m) for the SwiGLU epilogue
        const char * w0, * w1;
        int type0, type1;
        int64_t m0;
        if (a.silu_mul) { m0 = rp; w0 = a.w[0]; w1 = a.w[1]; type0 = a.type[0]; type1 = a.type[1]; }
        else            { m0 = 2 * (int64_t) rpl; w0 = w1 = a.w[mi]; type0 = type1 = a.type[mi]; }
        const size_t rb0 = row_size(type0, a.K), rb1 = row_size(type1, a.K);
        const char * r0 = w0 + (size_t) m0 * rb0;
        const char * r1 = w1 + (size_t) (a.silu_mul ? m0 : m0 + 1) * rb1;
        float acc0 = 0.0f, acc1 = 0.0f;
        for (int f0 = 0; f0 < 2 * gpr; f0 += 128) {             // two passes of 64 (row, group) items per iteration: 6 loads in flight (q4_K)
            float part[2];
            int   rowsel[2];
            // q4_K: every load of both passes goes out before the first dot
            uint4 hdr[2], qa[2], qb[2];
            const unsigned char * bp[2];
            int   g[2], p[2], ty[2];
            bool  ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = f0 + 64 * u + lane;
                ok[u] = f < 2 * gpr;
                const int fc = ok[u] ? f : 2 * gpr - 1;
                rowsel[u] = fc >= gpr;
                g[u] = fc - rowsel[u] * gpr;
                p[u] = g[u] & 3;
                ty[u] = rowsel[u] ? type1 : type0;
                const char * rr = rowsel[u] ? r1 : r0;
                bp[u] = reinterpret_cast<const unsigned char *>(rr) + (size_t) (g[u] >> 2) * (ty[u] == PKV_TYPE_Q4_K ? 144 : 210);
                if (ty[u] == PKV_TYPE_Q4_K) {
                    hdr[u] = *reinterpret_cast<const uint4 *>(bp[u]);
                    qa[u]  = *reinterpret_cast<const uint4 *>(bp[u] + 16 + 32 * p[u]);
                    qb[u]  = *reinterpret_cast<const uint4 *>(bp[u] + 32 + 32 * p[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (ty[u] == PKV_TYPE_Q4_K) part[u] = dot_q4k(hdr[u], qa[u], qb[u], p[u], sq, sd, ss, gs, g[u]);
                else                        part[u] = dot_q6k(bp[u], p[u], sq, sd, gs, (g[u] & ~3) + 2 * (p[u] >> 1));
                if (!ok[u]) part[u] = 0.0f;
                acc0 += rowsel[u] ? 0.0f : part[u];
                acc1 += rowsel[u] ? part[u] : 0.0f;
            }
        }
        acc0 = wsum(acc0);
        acc1 = wsum(acc1);
        if (lane == 0) {
            if (a.silu_mul) {
                a.y[0][m0] = acc0 / (1.0f + expf(-acc0)) * acc1;
            } else {
                float * y = a.y[mi];
                if (a.residual) { y[m0] = a.residual[m0] + acc0; y[m0 + 1] = a.residual[m0 + 1] + acc1; }
                else            { y[m0] = acc0; y[m0 + 1] = acc1; }
            }
        }
    }
}

// ---- ggml_get_rows (/ref/src/llama-model.cpp:4047 inp_embd = get_rows(tok_embd, tokens); CPU kernel ggml_compute_forward_get_rows,
//      /ref/ggml/src/ggml-cpu/ggml-cpu.c:8940-9090): out[i][:] = dequantize(row ids[i]) ----------------------------------------------
template <int T>
__global__ void __launch_bounds__(256) get_rows_kernel(const char * w, const int32_t * ids, float * out, int64_t K, int64_t n_rows_w) {
    const int64_t r = ids[blockIdx.x];
    float * o = out + (size_t) blockIdx.x * K;
    if (r < 0 || r >= n_rows_w) { for (int64_t k = threadIdx.x; k < K; k += 256) o[k] = 0.0f; return; }
    const unsigned char * row = reinterpret_cast<const unsigned char *>(w) + (size_t) r * row_size(T, K);
    if constexpr (T == PKV_TYPE_F32) {
        for (int64_t k = threadIdx.x; k < K; k += 256) o[k] = reinterpret_cast<const float *>(row)[k];
    } else if constexpr (T == PKV_TYPE_F16) {
        for (int64_t k = threadIdx.x; k < K; k += 256) o[k] = (float) reinterpret_cast<const _Float16 *>(row)[k];
    } else if constexpr (T == PKV_TYPE_Q8_0) {
        for (int64_t k = threadIdx.x; k < K; k += 256) {
            const unsigned char * b = row + (k >> 5) * 34;
            o[k] = h2f(*reinterpret_cast<const uint16_t *>(b)) * (float) (int) (signed char) b[2 + (k & 31)];
        }
    } else if constexpr (T == PKV_TYPE_Q4_K) {      // dequantize_row_q4_K (/ref/ggml/src/ggml-quants.c:1280-1302)
        for (int64_t k = threadIdx.x; k < K; k += 256) {
            const unsigned char * b = row + (k >> 8) * 144;
            const int e = (int) (k & 255), j = e >> 5, l = e & 31;           // sub-block j, element l
            const float d = h2f(*reinterpret_cast<const uint16_t *>(b)), dmin = h2f(*reinterpret_cast<const uint16_t *>(b + 2));
            const unsigned char * s = b + 4;
            int sc, mn;
            if (j < 4) { sc = s[j] & 63; mn = s[j + 4] & 63; }
            else       { sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4); mn = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
            const unsigned char qb = b[16 + 32 * (j >> 1) + l];
            const int q = (j & 1) ? (qb >> 4) : (qb & 0xF);
            const float d1 = d * (float) sc, m1 = dmin * (float) mn;         // f32 products, as the reference forms them
            o[k] = d1 * (float) q - m1;
        }
    } else {                                         // q6_K: dequantize_row_q6_K (:1690-1719)
        for (int64_t k = threadIdx.x; k < K; k += 256) {
            const unsigned char * b = row + (k >> 8) * 210;
            const int e = (int) (k & 255), n = e >> 7, r128 = e & 127, j = r128 >> 5, l = r128 & 31;
            const unsigned char lq = b[64 * n + l + 32 * (j & 1)], hq = b[128 + 32 * n + l];
            const int q = (int) (((j >> 1) ? (lq >> 4) : (lq & 0xF)) | (((hq >> (2 * j)) & 3) << 4)) - 32;
            const float d = h2f(*reinterpret_cast<const uint16_t *>(b + 208));
            const int sc = (int) reinterpret_cast<const signed char *>(b + 192)[8 * n + (l >> 4) + 2 * j];
            o[k] = d * (float) sc * (float) q;
        }
    }
}

}  // namespace

size_t act_image_bytes(int64_t K) { return (size_t) K + (size_t) (K / 256) * 4 + (size_t) (K / 32) * 4; }

int launch_quantize_act(const float * x, void * img, int64_t K, hipStream_t st) {
    PKV_REQUIRE(K > 0 && K % 256 == 0, "pkv_quantize_act: K must be a positive multiple of 256");
    hipLaunchKernelGGL(quant_act_kernel, dim3((unsigned) ((K / 4 + 255) / 256)), dim3(256), 0, st, x, (char *) img, K);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_gemv(const pkv_gemv & g, hipStream_t st) {
    PKV_REQUIRE(g.n >= 1 && g.n <= PKV_GEMV_MAX && g.xq && g.K > 0 && g.K % 256 == 0 && g.K <= 49152, "pkv_gemv_q: bad arguments");
    PKV_REQUIRE(!g.silu_mul || (g.n == 2 && g.M[0] == g.M[1] && !g.residual), "pkv_gemv_q: silu_mul needs two products of equal M and no residual");
    PKV_REQUIRE(!g.residual || g.n == 1, "pkv_gemv_q: the residual epilogue takes one product");
    GemvArgs a = {};
    int tot = 0;
    for (int i = 0; i < g.n; ++i) {
        PKV_REQUIRE(g.w[i] && g.y[i] && g.M[i] > 0 && g.M[i] % 2 == 0, "pkv_gemv_q: matrix %d: null pointer or odd row count", i);
        if (g.type_w[i] != PKV_TYPE_Q4_K && g.type_w[i] != PKV_TYPE_Q6_K) { set_error("pkv_gemv_q: weight type %d not supported (q4_K, q6_K)", g.type_w[i]); return PKV_ERR_UNSUPPORTED; }
        a.w[i] = (const char *) g.w[i]; a.y[i] = g.y[i]; a.M[i] = g.M[i]; a.type[i] = g.type_w[i];
        tot += g.silu_mul ? (i == 0 ? (int) g.M[0] : 0) : (int) (g.M[i] / 2);
        a.rp_end[i] = tot;
    }
    a.n = g.n; a.silu_mul = g.silu_mul; a.residual = g.residual; a.img = (const char *) g.xq; a.K = g.K;
    a.gpr = (int) (g.K / 64); a.gs = a.gpr | 1;
    const size_t lds = (size_t) 20 * a.gs * 4;
    // every workgroup pays one copy of the column image: a few row pairs per wave at least, 4 workgroups per CU at most
    int grid = (tot + 3) / 4;
    const int cap = 256 * 4;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned) grid), dim3(256), lds, st, a);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_get_rows(const void * w, int type_w, const int32_t * ids, float * out, int64_t n_ids, int64_t K, int64_t n_rows_w, hipStream_t st) {
    if (n_ids == 0) return PKV_OK;
    const dim3 grid((unsigned) n_ids);
    switch (type_w) {
        case PKV_TYPE_F32:  hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_F32>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_F16:  hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_F16>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q8_0: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q8_0>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q4_K: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q4_K>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q6_K: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q6_K>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        default: set_error("pkv_get_rows: type %d not supported", type_w); return PKV_ERR_UNSUPPORTED;
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

}  // namespace pkv
