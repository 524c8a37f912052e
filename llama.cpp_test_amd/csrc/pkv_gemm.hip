// pkv_gemm.hip — batched weight products (prompt processing: many activation columns) on the matrix cores of gfx950.
//
// SURVEY §8(f) row 1, "MFMA batched products": ggml_mul_mat of a llama layer at n_tokens > 1 (/ref/src/llama-model.cpp:4070-4160; harness sweep
// /ref/tests/test-backend-ops.cpp:4436-4442; CPU kernel ggml_compute_forward_mul_mat, /ref/ggml/src/ggml-cpu/ggml-cpu.c:7467-7640):
//     y[n][m] = sum_k W[m][k] x[n][k],   W = M rows of K elements in f16 / q8_0 / q4_K / q6_K (the reference's row formats), x = N rows of K f32.
// The decode-step kernels (pkv_gemv.hip, pkv_layer_ops.hip) stream W once per activation column; here every weight tile is read ONCE per 128 columns,
// dequantized to f16 on its way into LDS, and multiplied by v_mfma_f32_32x32x16_f16 with f32 accumulation (the activations are rounded to f16: the
// reference's CPU backend rounds them to q8_K / q8_0 integers or f16 — both are ~1e-3 relative per element; the harness bar NMSE 5e-4 holds).
//
//   workgroup = 128 (m) x 128 (n) outputs, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers); 64 x 128 and 64 x 64 tiles
//   where the launch would otherwise leave CUs without a second workgroup (launch_gemm);
//   K in steps of 64: thread t stages row t >> 1, 32-element half t & 1 of BOTH operands — one q4_K / q6_K sub-block (or q8_0 block, or 32 halfs)
//   of W, 32 floats of x — raw loads issued one step ahead of the conversion (issue early / write late) into the OTHER of two LDS images per operand:
//   ONE barrier per step, the stores run under the step's MFMAs (r02: q8_0 / q6_K +25 ... +45 % at 4096 rows, tall matrices 32000 x 2048 +57 ... +68 %);
//   LDS rows of 64 f16 padded to 144 B: the 16-byte fragment reads (lane = row r, k = 8 h + ..) are conflict-free;
//   fragments per cdna_hip_programming.md §3: A[row r][k = 8 h + j], B[k = 8 h + j][col r]; C col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h,
//   so a lane owns 4 consecutive m of one token n per register group: float4 stores into y[n][m].
#include "pkv_common.h"

#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <unordered_map>

namespace pkv {
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 64, LDR = BK + 8;   // LDS row = 72 halfs = 144 B; tile = BM weight rows x BN columns: 128 x 128, or 64 x 128 / 64 x 64 where that would leave CUs idle

__device__ __forceinline__ float hf(uint16_t b) { return (float) *reinterpret_cast<const _Float16 *>(&b); }
__device__ __forceinline__ uint4 ldu(const unsigned char * p) { return *reinterpret_cast<const uint4 *>(p); }

// raw bytes of 32 consecutive weights of one row (k0 a multiple of 32), and their conversion to 32 halfs
// T may carry PKV_TYPE_PACKED: the row is in the plane layout of pkv_wlayout.h (same bytes, same conversion; only load() differs)
template <int T> struct RawW;
template <> struct RawW<PKV_TYPE_F16> {
    static constexpr int NLOADS = 4;   // vector-memory instructions of load() (gemm2_kernel's counted waits)
    uint4 q[4];
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0) { const unsigned char * p = row + k0 * 2;
        q[0] = ldu(p); q[1] = ldu(p + 16); q[2] = ldu(p + 32); q[3] = ldu(p + 48); }
    __device__ __forceinline__ void to_half(_Float16 * o, int64_t) const { uint4 * d = reinterpret_cast<uint4 *>(o); d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3]; }
};
template <> struct RawW<PKV_TYPE_Q8_0> {   // block_q8_0: f16 d, int8 qs[32] (/ref/ggml/src/ggml-common.h:197-203)
    static constexpr int NLOADS = 3;
    uint4 q[2]; uint16_t d;
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0) { const unsigned char * p = row + (k0 >> 5) * 34;
        d = *reinterpret_cast<const uint16_t *>(p); q[0] = ldu(p + 2); q[1] = ldu(p + 18); }
    __device__ __forceinline__ void to_half(_Float16 * o, int64_t) const {
        const float df = hf(d);
        const signed char * b = reinterpret_cast<const signed char *>(q);
#pragma unroll
        for (int i = 0; i < 32; ++i) o[i] = (_Float16) (df * (float) b[i]);
    }
};
template <> struct RawW<PKV_TYPE_Q4_K> {   // block_q4_K (:279-290): sub-block j = (k0 % 256) / 32: qs[32 (j / 2) + l] nibble j & 1
    static constexpr int NLOADS = 3;
    uint4 h, q[2];
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0) { const unsigned char * p = row + (k0 >> 8) * 144; const int j = (int) (k0 & 255) >> 5;
        h = ldu(p); q[0] = ldu(p + 16 + 32 * (j >> 1)); q[1] = ldu(p + 32 + 32 * (j >> 1)); }
    __device__ __forceinline__ void to_half(_Float16 * o, int64_t k0) const {
        const int j = (int) (k0 & 255) >> 5;
        const float d = hf((uint16_t) (h.x & 0xFFFF)), dmin = hf((uint16_t) (h.x >> 16));
        const int sh = 8 * (j & 3);
        const int b0 = (int) ((h.y >> sh) & 0xFF), b1 = (int) ((h.z >> sh) & 0xFF), b2 = (int) ((h.w >> sh) & 0xFF);   // get_scale_min_k4
        const int sc = j < 4 ? (b0 & 63) : ((b2 & 0xF) | ((b0 >> 6) << 4)), mn = j < 4 ? (b1 & 63) : ((b2 >> 4) | ((b1 >> 6) << 4));
        const float d1 = d * (float) sc, m1 = dmin * (float) mn;
        const unsigned char * b = reinterpret_cast<const unsigned char *>(q);
        const int s4 = 4 * (j & 1);
#pragma unroll
        for (int i = 0; i < 32; ++i) o[i] = (_Float16) (d1 * (float) ((b[i] >> s4) & 0xF) - m1);
    }
};
template <> struct RawW<(PKV_TYPE_Q4_K | PKV_TYPE_PACKED)> : RawW<PKV_TYPE_Q4_K> {
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0, int64_t K) { const size_t gpr = (size_t) (K >> 6), g = (size_t) (k0 >> 6);   // group g = 4 b + (j >> 1)
        h = ldu(row + 32 * gpr + 16 * (g >> 2)); q[0] = ldu(row + 16 * g); q[1] = ldu(row + 16 * gpr + 16 * g); }
};
template <> struct RawW<PKV_TYPE_Q5_K> {   // block_q5_K (:291-308): header {d, dmin, scales[12]}, qh[32] (bit j of byte l = the fifth bit of element 32 j + l), qs[128] as q4_K
    static constexpr int NLOADS = 5;
    uint4 h, qh[2], q[2];
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0) { const unsigned char * p = row + (k0 >> 8) * 176; const int j = (int) (k0 & 255) >> 5;
        h = ldu(p); qh[0] = ldu(p + 16); qh[1] = ldu(p + 32); q[0] = ldu(p + 48 + 32 * (j >> 1)); q[1] = ldu(p + 64 + 32 * (j >> 1)); }
    __device__ __forceinline__ void to_half(_Float16 * o, int64_t k0) const {   // dequantize_row_q5_K (/ref/ggml/src/ggml-quants.c:1482-1507)
        const int j = (int) (k0 & 255) >> 5;
        const float d = hf((uint16_t) (h.x & 0xFFFF)), dmin = hf((uint16_t) (h.x >> 16));
        const int sh = 8 * (j & 3);
        const int b0 = (int) ((h.y >> sh) & 0xFF), b1 = (int) ((h.z >> sh) & 0xFF), b2 = (int) ((h.w >> sh) & 0xFF);   // get_scale_min_k4
        const int sc = j < 4 ? (b0 & 63) : ((b2 & 0xF) | ((b0 >> 6) << 4)), mn = j < 4 ? (b1 & 63) : ((b2 >> 4) | ((b1 >> 6) << 4));
        const float d1 = d * (float) sc, m1 = dmin * (float) mn;
        const unsigned char * b = reinterpret_cast<const unsigned char *>(q), * hb = reinterpret_cast<const unsigned char *>(qh);
        const int s4 = 4 * (j & 1);
#pragma unroll
        for (int i = 0; i < 32; ++i) o[i] = (_Float16) (d1 * (float) (((b[i] >> s4) & 0xF) | (((hb[i] >> j) & 1) << 4)) - m1);
    }
};
template <> struct RawW<(PKV_TYPE_Q5_K | PKV_TYPE_PACKED)> : RawW<PKV_TYPE_Q5_K> {   // planes of pkv_wlayout.h: qa, qb per group; ha, hb, header per block
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0, int64_t K) { const size_t gpr = (size_t) (K >> 6), g = (size_t) (k0 >> 6);
        h = ldu(row + 40 * gpr + 16 * (g >> 2)); qh[0] = ldu(row + 32 * gpr + 16 * (g >> 2)); qh[1] = ldu(row + 36 * gpr + 16 * (g >> 2));
        q[0] = ldu(row + 16 * g); q[1] = ldu(row + 16 * gpr + 16 * g); }
};
template <> struct RawW<PKV_TYPE_Q6_K> {   // block_q6_K (:309-319): e0 = k0 % 256, n = e0 / 128, j = (e0 % 128) / 32: ql[64 n + 32 (j & 1) + l] nibble j >> 1, qh[32 n + l] bits 2 j
    static constexpr int NLOADS = 6;
    uint4 ql[2], qh[2]; uint16_t sc2, d;
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0) { const unsigned char * p = row + (k0 >> 8) * 210; const int e0 = (int) (k0 & 255), n = e0 >> 7, j = (e0 & 127) >> 5;
        ql[0] = ldu(p + 64 * n + 32 * (j & 1)); ql[1] = ldu(p + 64 * n + 32 * (j & 1) + 16); qh[0] = ldu(p + 128 + 32 * n); qh[1] = ldu(p + 128 + 32 * n + 16);
        sc2 = *reinterpret_cast<const uint16_t *>(p + 192 + 8 * n + 2 * j); d = *reinterpret_cast<const uint16_t *>(p + 208); }
    __device__ __forceinline__ void to_half(_Float16 * o, int64_t k0) const {
        const int j = ((int) (k0 & 127)) >> 5;
        const float df = hf(d), s0 = df * (float) (signed char) (sc2 & 0xFF), s1 = df * (float) (signed char) (sc2 >> 8);
        const unsigned char * l = reinterpret_cast<const unsigned char *>(ql), * hh = reinterpret_cast<const unsigned char *>(qh);
        const int s4 = 4 * (j >> 1), s2 = 2 * j;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int qv = (int) (((l[i] >> s4) & 0xF) | (((hh[i] >> s2) & 3) << 4)) - 32;
            o[i] = (_Float16) ((i < 16 ? s0 : s1) * (float) qv);
        }
    }
};

template <> struct RawW<(PKV_TYPE_Q6_K | PKV_TYPE_PACKED)> : RawW<PKV_TYPE_Q6_K> {
    __device__ __forceinline__ void load(const unsigned char * row, int64_t k0, int64_t K) {
        const size_t gpr = (size_t) (K >> 6), b = (size_t) (k0 >> 8); const int e0 = (int) (k0 & 255), n = e0 >> 7, j = (e0 & 127) >> 5;
        const size_t it = 4 * b + 2 * n;                                                        // items (b, n, is = 0 / 1): 32 contiguous bytes of a plane
        const unsigned char * pl = row + (size_t) (j & 1) * 16 * gpr + 16 * it, * ph = row + 32 * gpr + 16 * it;
        ql[0] = ldu(pl); ql[1] = ldu(pl + 16); qh[0] = ldu(ph); qh[1] = ldu(ph + 16);
        sc2 = *reinterpret_cast<const uint16_t *>(row + 48 * gpr + 16 * b + 8 * n + 2 * j); d = *reinterpret_cast<const uint16_t *>(row + 52 * gpr + 2 * b); }
};

// up to GEMM_MAX matrices of one weight type against the same activations in ONE launch (q + k, gate + up of a llama layer: launch_mul_mat_multi):
// blockIdx.x runs over the row tiles of all of them.  A product of a few hundred rows (k, v: 1024 rows at 8B, 256 at TinyLlama) is a handful of
// workgroups walking all of K in 64-deep steps at one load latency per step — 50 us whatever its size — so it rides along with a large one instead.
constexpr int GEMM_MAX = 4;
struct GemmArgs { const unsigned char * w[GEMM_MAX]; float * y[GEMM_MAX]; int64_t M[GEMM_MAX]; int tile_end[GEMM_MAX]; int n; const float * x; int64_t N, K, row_bytes; int ksplit; };   // ksplit 2: blockIdx.z halves K, results added into zeroed y

template <int T, int BM, int BN>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs a) {
    // waves along m / n and MFMA tiles per wave along m / n: 128 x 128: 2 x 2 waves of 2 x 2 tiles; 64 x 128: 1 x 4 of 2 x 1; 64 x 64: 2 x 2 of 1 x 1
    constexpr int WM = BN == 64 ? 2 : BM / 64, WN = 4 / WM, TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WM * WN == 4, "tile shape");
    __shared__ __attribute__((aligned(16))) _Float16 sAb[2][BM * LDR], sBb[2][BN * LDR];   // two images per operand: ONE barrier per K-step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave % WM, wn = wave / WM;
    const int r = lane & 31, h = lane >> 5;
    int mi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX; ++i) if (i < a.n && (int) blockIdx.x >= a.tile_end[i - 1]) mi = i;
    const int64_t aM = a.M[mi];
    const unsigned char * const aw = a.w[mi];
    float * const ay = a.y[mi];
    const int64_t m0 = (int64_t) ((int) blockIdx.x - (mi ? a.tile_end[mi - 1] : 0)) * BM, n0 = (int64_t) blockIdx.y * BN;
    // staging role: row srow, 32-element half shalf of both tiles (rows past the end re-read the last row; their outputs are never stored)
    const int srow = tid >> 1, shalf = tid & 1;
    const bool stage_a = srow < BM, stage_b = srow < BN;
    const unsigned char * wrow = aw + (size_t) min(m0 + (stage_a ? srow : 0), aM - 1) * a.row_bytes;
    const float * xrow = a.x + (size_t) min(n0 + (stage_b ? srow : 0), a.N - 1) * a.K;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    RawW<T> rw;
    float4 rx[8];
    auto load_raw = [&](int64_t k0) {
        if (stage_a) { if constexpr ((T & PKV_TYPE_PACKED) != 0) rw.load(wrow, k0 + 32 * shalf, a.K); else rw.load(wrow, k0 + 32 * shalf); }
        if (stage_b) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rx[i] = reinterpret_cast<const float4 *>(xrow + k0 + 32 * shalf)[i];
        }
    };
    // split-K (launch_gemm_multi): this workgroup's half of K; with TWO partial sums per element the atomic adds commute exactly (a + b == b + a)
    const int64_t kbeg = a.ksplit > 1 ? (int64_t) blockIdx.z * (a.K / a.ksplit) : 0, kend = a.ksplit > 1 ? kbeg + a.K / a.ksplit : a.K;
    auto stage = [&](int64_t k0, _Float16 * sA, _Float16 * sB) {   // conversion of the loaded step into an LDS image pair
        if (stage_a) {
            __attribute__((aligned(16))) _Float16 tmp[32];
            rw.to_half(tmp, k0 + 32 * shalf);
            uint4 * da = reinterpret_cast<uint4 *>(sA + srow * LDR + 32 * shalf);
#pragma unroll
            for (int i = 0; i < 4; ++i) da[i] = reinterpret_cast<const uint4 *>(tmp)[i];
        }
        if (stage_b) {
            __attribute__((aligned(16))) _Float16 tb[32];
#pragma unroll
            for (int i = 0; i < 8; ++i) { tb[4 * i] = (_Float16) rx[i].x; tb[4 * i + 1] = (_Float16) rx[i].y; tb[4 * i + 2] = (_Float16) rx[i].z; tb[4 * i + 3] = (_Float16) rx[i].w; }
            uint4 * db = reinterpret_cast<uint4 *>(sB + srow * LDR + 32 * shalf);
#pragma unroll
            for (int i = 0; i < 4; ++i) db[i] = reinterpret_cast<const uint4 *>(tb)[i];
        }
    };
    load_raw(kbeg);
    stage(kbeg, sAb[0], sBb[0]);
    if (kbeg + BK < kend) load_raw(kbeg + BK);
    int it = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK, ++it) {
        const _Float16 * sA = sAb[it & 1], * sB = sBb[it & 1];
        __syncthreads();                                  // image `it` is complete; everyone is done reading image it - 1 (the other one)
        if (k0 + BK < kend) {
            stage(k0 + BK, sAb[(it + 1) & 1], sBb[(it + 1) & 1]);   // the step loaded a whole step ago -> the other image, under this step's MFMAs
            if (k0 + 2 * BK < kend) load_raw(k0 + 2 * BK);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8_t *>(sA + (wm * 32 * TM + i * 32 + r) * LDR + 16 * ks + 8 * h);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8_t *>(sB + (wn * 32 * TN + j * 32 + r) * LDR + 16 * ks + 8 * h);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // C: col (token) = lane & 31, row (m) = (reg & 3) + 8 (reg >> 2) + 4 h
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn * 32 * TN + j * 32 + r;
        if (n >= a.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t m = m0 + wm * 32 * TM + i * 32 + 8 * g + 4 * h;
                if (a.ksplit > 1) {
                    for (int e = 0; e < 4; ++e) if (m + e < aM) unsafeAtomicAdd(ay + (size_t) n * aM + m + e, acc[i][j][4 * g + e]);
                } else if ((aM & 3) == 0 && m + 3 < aM) *reinterpret_cast<float4 *>(ay + (size_t) n * aM + m) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                else
                    for (int e = 0; e < 4; ++e) if (m + e < aM) ay[(size_t) n * aM + m + e] = acc[i][j][4 * g + e];
            }
    }
}


// ======================================================================================================================================================
// r03: the second form of the product (gemm2_kernel).  What bounded the kernel above (profiles/r02_gemm_bench.txt: 0.08-0.17 of the MFMA peak) was its staging:
// every workgroup converted its 128 x 64 activation tile from f32 on every K-step (the same tile in every row-tile's workgroup), every operand element went
// HBM -> VGPR -> VALU -> ds_write, and a 128 x 128 tile of 4 waves re-read each LDS byte for only 64 x 64 of MFMA work.  Here:
//   * the activations are rounded to f16 ONCE per product (convert pass into a per-stream scratch, 3 % of the product's time at 2048 columns) and their tile
//     goes HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPRs, no VALU, no ds_write; f16 weights travel the same way;
//   * quantized weight tiles are dequantized with packed f16 arithmetic (the 0x6400 byte-permute of the attention kernels: two elements per instruction,
//     the sub-block scale rounded to f16 once) and written as 16-byte chunks;
//   * LDS rows are 64 halfs = 128 B, unpadded (LDS-DMA writes lane-linear), 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): fragment reads
//     (16 consecutive rows x one chunk) and the staging writes touch every bank once;
//   * tiles up to 256 x 256 with 8 waves (2 x 4, wave tile 128 x 64: 128 accumulator registers), two LDS images, one barrier per 64-deep K-step;
//   * workgroup order: column tiles fastest, so the workgroups that run together share one band of weight rows and each XCD's L2 keeps its own activation tile.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2v u2h(uint32_t u) { return *reinterpret_cast<const h2v *>(&u); }
__device__ __forceinline__ uint32_t h2u(h2v h) { return *reinterpret_cast<const uint32_t *>(&h); }
__device__ __forceinline__ h2v splat(float f) { const _Float16 x = (_Float16) f; return h2v{x, x}; }
// 4 bytes u (0..255 each) -> 4 halfs (u - bias) * mul + add, in two dwords (elements 0, 1 | 2, 3)
__device__ __forceinline__ void cvt4(uint32_t b, h2v nbias, h2v mul, h2v add, uint32_t & o01, uint32_t & o23) {
    o01 = h2u((u2h(__builtin_amdgcn_perm(0x64646464u, b, 0x04010400u)) + nbias) * mul + add);
    o23 = h2u((u2h(__builtin_amdgcn_perm(0x64646464u, b, 0x04030402u)) + nbias) * mul + add);
}
// 32 consecutive weights of a row as 32 halfs in 4 x 16 bytes (k order)
template <int T> struct Deq;
template <> struct Deq<PKV_TYPE_Q8_0> { static __device__ __forceinline__ void run(const RawW<PKV_TYPE_Q8_0> & w, int64_t, uint4 * o) {
    const h2v d = u2h((uint32_t) w.d * 0x10001u), nb = splat(-1152.0f), z = splat(0.0f);   // int8 ^ 0x80 = value + 128 under the 1024 of the magic
    const uint32_t * q = reinterpret_cast<const uint32_t *>(w.q);
    uint32_t * ow = reinterpret_cast<uint32_t *>(o);
#pragma unroll
    for (int i = 0; i < 8; ++i) cvt4(q[i] ^ 0x80808080u, nb, d, z, ow[2 * i], ow[2 * i + 1]);
} };
template <> struct Deq<PKV_TYPE_Q4_K> { static __device__ __forceinline__ void run(const RawW<PKV_TYPE_Q4_K> & w, int64_t k0, uint4 * o) {
    const int j = (int) (k0 & 255) >> 5;
    const float d = hf((uint16_t) (w.h.x & 0xFFFF)), dmin = hf((uint16_t) (w.h.x >> 16));
    const int sh = 8 * (j & 3);
    const int b0 = (int) ((w.h.y >> sh) & 0xFF), b1 = (int) ((w.h.z >> sh) & 0xFF), b2 = (int) ((w.h.w >> sh) & 0xFF);   // get_scale_min_k4
    const int sc = j < 4 ? (b0 & 63) : ((b2 & 0xF) | ((b0 >> 6) << 4)), mn = j < 4 ? (b1 & 63) : ((b2 >> 4) | ((b1 >> 6) << 4));
    const h2v d1 = splat(d * (float) sc), m1 = splat(-(dmin * (float) mn)), nb = splat(-1024.0f);
    const uint32_t * q = reinterpret_cast<const uint32_t *>(w.q);
    uint32_t * ow = reinterpret_cast<uint32_t *>(o);
    const int s4 = 4 * (j & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) cvt4((q[i] >> s4) & 0x0F0F0F0Fu, nb, d1, m1, ow[2 * i], ow[2 * i + 1]);
} };
template <> struct Deq<PKV_TYPE_Q5_K> { static __device__ __forceinline__ void run(const RawW<PKV_TYPE_Q5_K> & w, int64_t k0, uint4 * o) {
    const int j = (int) (k0 & 255) >> 5;
    const float d = hf((uint16_t) (w.h.x & 0xFFFF)), dmin = hf((uint16_t) (w.h.x >> 16));
    const int sh = 8 * (j & 3);
    const int b0 = (int) ((w.h.y >> sh) & 0xFF), b1 = (int) ((w.h.z >> sh) & 0xFF), b2 = (int) ((w.h.w >> sh) & 0xFF);   // get_scale_min_k4
    const int sc = j < 4 ? (b0 & 63) : ((b2 & 0xF) | ((b0 >> 6) << 4)), mn = j < 4 ? (b1 & 63) : ((b2 >> 4) | ((b1 >> 6) << 4));
    const h2v d1 = splat(d * (float) sc), m1 = splat(-(dmin * (float) mn)), nb = splat(-1024.0f);
    const uint32_t * q = reinterpret_cast<const uint32_t *>(w.q), * hh = reinterpret_cast<const uint32_t *>(w.qh);
    uint32_t * ow = reinterpret_cast<uint32_t *>(o);
    const int s4 = 4 * (j & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) cvt4(((q[i] >> s4) & 0x0F0F0F0Fu) | (((hh[i] >> j) & 0x01010101u) << 4), nb, d1, m1, ow[2 * i], ow[2 * i + 1]);   // 5-bit quants 0..31
} };
template <> struct Deq<(PKV_TYPE_Q5_K | PKV_TYPE_PACKED)> { static __device__ __forceinline__ void run(const RawW<(PKV_TYPE_Q5_K | PKV_TYPE_PACKED)> & w, int64_t k0, uint4 * o) {
    Deq<PKV_TYPE_Q5_K>::run(w, k0, o); } };
template <> struct Deq<(PKV_TYPE_Q4_K | PKV_TYPE_PACKED)> { static __device__ __forceinline__ void run(const RawW<(PKV_TYPE_Q4_K | PKV_TYPE_PACKED)> & w, int64_t k0, uint4 * o) {
    Deq<PKV_TYPE_Q4_K>::run(w, k0, o); } };
template <> struct Deq<PKV_TYPE_Q6_K> { static __device__ __forceinline__ void run(const RawW<PKV_TYPE_Q6_K> & w, int64_t k0, uint4 * o) {
    const int j = ((int) (k0 & 127)) >> 5;
    const float df = hf(w.d);
    const h2v s0 = splat(df * (float) (signed char) (w.sc2 & 0xFF)), s1 = splat(df * (float) (signed char) (w.sc2 >> 8)), nb = splat(-1056.0f), z = splat(0.0f);   // 1024 + the 32 of q6_K
    const uint32_t * l = reinterpret_cast<const uint32_t *>(w.ql), * hh = reinterpret_cast<const uint32_t *>(w.qh);
    uint32_t * ow = reinterpret_cast<uint32_t *>(o);
    const int s4 = 4 * (j >> 1), s2 = 2 * j;
#pragma unroll
    for (int i = 0; i < 8; ++i) cvt4(((l[i] >> s4) & 0x0F0F0F0Fu) | (((hh[i] >> s2) & 0x03030303u) << 4), nb, i < 4 ? s0 : s1, z, ow[2 * i], ow[2 * i + 1]);
} };
template <> struct Deq<(PKV_TYPE_Q6_K | PKV_TYPE_PACKED)> { static __device__ __forceinline__ void run(const RawW<(PKV_TYPE_Q6_K | PKV_TYPE_PACKED)> & w, int64_t k0, uint4 * o) {
    Deq<PKV_TYPE_Q6_K>::run(w, k0, o); } };

typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int ROWB2 = BK * 2;                                 // LDS row of the second form: 64 halfs, unpadded
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) gemm2_kernel(const GemmArgs a, const _Float16 * __restrict__ xh) {
    constexpr int NT = 64 * WM * WN, NWAVE = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr bool ADMA = (T & ~PKV_TYPE_PACKED) == PKV_TYPE_F16;          // f16 weights: no conversion, the tile goes by LDS-DMA like the activations
    constexpr int IMG = (BM + BN) * ROWB2;
    static_assert(TM >= 1 && TN >= 1 && BM % (8 * NWAVE) == 0 && BN % (8 * NWAVE) == 0 && (2 * BM) % 64 == 0, "tile shape");
    static_assert(2 * BM <= NT || ADMA, "one (row, half) item per thread in the register-staged weight tile");
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG];          // [image][A rows | B rows]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave % WM, wn = wave / WM;
    const int r = lane & 31, h = lane >> 5;
    // workgroup -> (row tile of one of the matrices, column tile): blockIdx.x = column tile (fastest), blockIdx.y = row tile over all matrices
    int mi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX; ++i) if (i < a.n && (int) blockIdx.y >= a.tile_end[i - 1]) mi = i;
    const int64_t aM = a.M[mi];
    const unsigned char * const aw = a.w[mi];
    float * const ay = a.y[mi];
    const int64_t m0 = (int64_t) ((int) blockIdx.y - (mi ? a.tile_end[mi - 1] : 0)) * BM, n0 = (int64_t) blockIdx.x * BN;

    // ---- LDS-DMA roles: instruction i of a wave fills rows (wave * NI + i) * 8 .. + 7 of a tile: lane l -> row + l / 8, chunk POSITION l % 8, which holds
    //      chunk (l % 8) ^ swz(row) of the source row (rows past the end of the operand re-read its last row; their outputs are never stored)
    constexpr int NIB = BN / 8 / NWAVE, NIA = BM / 8 / NWAVE;
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(xh) + n0 * a.K, 0, (int) (min((int64_t) BN, a.N - n0) * a.K * 2), 0x00020000);
    int vo_b[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int row = (wave * NIB + i) * 8 + (lane >> 3);
        vo_b[i] = (int) (min((int64_t) row, a.N - 1 - n0) * a.K * 2) + (((lane & 7) ^ swz(row)) << 4);
    }
    [[maybe_unused]] rsrc_t rs_w;
    [[maybe_unused]] int vo_a[ADMA ? NIA : 1];
    if constexpr (ADMA) {
        rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(aw) + m0 * a.row_bytes, 0, (int) (min((int64_t) BM, aM - m0) * a.row_bytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int row = (wave * NIA + i) * 8 + (lane >> 3);
            vo_a[i] = (int) (min((int64_t) row, aM - 1 - m0) * a.row_bytes) + (((lane & 7) ^ swz(row)) << 4);
        }
    }
    auto dma = [&](int64_t k0, char * img) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (ADMA) {
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *) (img + (wave * NIA + i) * 1024), 16, vo_a[i], (int) (k0 * 2), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *) (img + BM * ROWB2 + (wave * NIB + i) * 1024), 16, vo_b[i], (int) (k0 * 2), 0, 0);
#endif
    };
    // ---- register-staged weight tile (quantized types): thread -> row tid >> 1, 32-element half tid & 1
    const int srow = tid >> 1, shalf = tid & 1;
    const bool stage_a = !ADMA && srow < BM;
    const unsigned char * wrow = aw + (size_t) min(m0 + (srow < BM ? srow : 0), aM - 1) * a.row_bytes;
    RawW<T> rw;
    auto load_raw = [&](int64_t k0) {
        if constexpr (!ADMA) {
            if (stage_a) { if constexpr ((T & PKV_TYPE_PACKED) != 0) rw.load(wrow, k0 + 32 * shalf, a.K); else rw.load(wrow, k0 + 32 * shalf); }
        }
    };
    auto write_a = [&](int64_t k0, char * img) {
        if constexpr (!ADMA) {
            if (stage_a) {
                uint4 o[4];
                Deq<T>::run(rw, k0 + 32 * shalf, o);
                char * dst = img + srow * ROWB2;
                const int key = swz(srow);
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(dst + (((4 * shalf + i) ^ key) << 4)) = o[i];
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment addresses: row = wave's first row + 32 i + r; chunk 2 ks + h at position (2 ks + h) ^ swz(row) — swz(row) is the same for every i (32 i is a multiple of 16)
    const int a_ro = (wm * 32 * TM + r) * ROWB2, b_ro = BM * ROWB2 + (wn * 32 * TN + r) * ROWB2, fkey = swz(r);

    // vector-memory order per step: [raw weight loads of step k+2] ... [LDS-DMA of step k+1] ...; the wait before the barrier leaves the NR youngest
    // operations — the raw loads just issued — in flight (memory returns in order: everything older, i.e. the LDS-DMA tiles of the next image, has landed)
    constexpr int NR = ADMA ? 0 : RawW<T>::NLOADS;
    const bool wave_stages = !ADMA && (wave * 64) / 2 < BM;       // wave-uniform: this wave's threads carry weight rows
    // (a raw s_barrier behind explicit waits: __syncthreads() would drain the raw loads too — with LDS-DMA in the loop hipcc puts vmcnt(0) in front of it)
    auto wait_dma_and_barrier = [&]() {
        if (wave_stages) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NR) : "memory");
        else             asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    load_raw(0);
    dma(0, smem);
    write_a(0, smem);
    load_raw(BK < a.K ? BK : 0);
    wait_dma_and_barrier();
    int it = 0;
    for (int64_t k0 = 0; k0 < a.K; k0 += BK, ++it) {
        const char * cur = smem + (it & 1) * IMG;
        char * nxt = smem + ((it + 1) & 1) * IMG;
        const bool more = k0 + BK < a.K;
        if (more) dma(k0 + BK, nxt);                       // the next step's LDS-DMA tiles: in flight under this step's MFMAs (the other image: last read one barrier ago)
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8_t af[TM], bf[TN];
            const int po = (((2 * ks + h) ^ fkey) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8_t *>(cur + a_ro + i * 32 * ROWB2 + po);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8_t *>(cur + b_ro + j * 32 * ROWB2 + po);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) write_a(k0 + BK, nxt);                   // the step loaded a whole step ago: converted and stored behind this step's MFMAs
        if (k0 + 2 * BK < a.K) load_raw(k0 + 2 * BK);
        else if (more) load_raw(k0 + BK);                  // (keeps the count of the wait below: a re-load of the last step, never used)
        if (more) wait_dma_and_barrier();                  // this wave's share of the next image has landed (LDS-DMA is counted in vmcnt) ... and everyone's; everyone is
    }                                                      // done reading the current image
    // C: col (token) = lane & 31, row (m) = (reg & 3) + 8 (reg >> 2) + 4 h
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn * 32 * TN + j * 32 + r;
        if (n >= a.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t m = m0 + wm * 32 * TM + i * 32 + 8 * g + 4 * h;
                if ((aM & 3) == 0 && m + 3 < aM) *reinterpret_cast<float4 *>(ay + (size_t) n * aM + m) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                else
                    for (int e = 0; e < 4; ++e) if (m + e < aM) ay[(size_t) n * aM + m + e] = acc[i][j][4 * g + e];
            }
    }
}

// ======================================================================================================================================================
// Third form (gemm3_kernel): the second form for the shapes whose K-step is too short to hide a memory round trip.  With 64 x 128 / 128 x 128 tiles a workgroup
// multiplies for 250-500 cycles per 64-deep step and then stands at the barrier until the next step's tiles — requested at the top of the same step — have
// landed: one memory latency per step, ~0.9 us (4096 x 4096 x 512 at 0.12 of the MFMA peak).  Here every global -> LDS transfer is requested TWO steps before
// its first use, and nothing in the loop is a VGPR load (hipcc waits vmcnt(0) for those whenever LDS-DMA is in flight: it cannot count the two kinds together):
//   * activations (f16 scratch) and f16 weights: LDS-DMA into a ring of THREE tile images, requested at step k for step k + 2;
//   * packed q4_K weights (pkv_wlayout.h planes): the step's RAW bytes — 16 B of qa, 16 B of qb and the block's 16-B header per row — go by LDS-DMA into a ring
//     of three raw buffers, requested at step k for step k + 3; at step k + 2 they are read back from LDS, dequantized (packed f16 arithmetic, as the second
//     form) and written into one of TWO f16 weight images for step k + 3;
//   * one barrier per step; the wait in front of it leaves exactly the transfers requested in this step in flight (s_waitcnt vmcnt(N), memory returns in order).
template <int T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) gemm3_kernel(const GemmArgs a, const _Float16 * __restrict__ xh) {
    constexpr int NT = 64 * WM * WN, NWAVE = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr bool ADMA = (T & ~PKV_TYPE_PACKED) == PKV_TYPE_F16;
    constexpr bool Q6 = T == (PKV_TYPE_Q6_K | PKV_TYPE_PACKED);
    static_assert(ADMA || Q6 || T == (PKV_TYPE_Q4_K | PKV_TYPE_PACKED), "third form: f16, packed q4_K and packed q6_K weights");
    static_assert(TM >= 1 && TN >= 1 && BM % 64 == 0 && BN % (8 * NWAVE) == 0 && (!ADMA || BM % (8 * NWAVE) == 0) && 2 * BM <= NT, "tile shape");   // (2 * BM <= NT: every row-half has a thread)
    constexpr int AIMG = BM * ROWB2, BIMG = BN * ROWB2, NA = ADMA ? 3 : 2;           // f16 tile images: weights (ring of 3 by DMA, or 2 by conversion), activations (3)
    // one raw buffer: q4_K planes qa | qb | header, 16 B per row each; q6_K planes la0 | la1 | lb0 | lb1 | hh0 | hh1 | scales (16 B per row each) and a 4-byte plane with the
    // dword that holds the block's f16 d (item (b, n, is) of pkv_wlayout.h: a 64-deep step = half n of block b needs both `is` items of la, lb and hh)
    constexpr int NPL = ADMA ? 0 : (Q6 ? 7 : 3);                                      // 16-byte granule planes
    constexpr int RAWB = ADMA ? 0 : NPL * BM * 16 + (Q6 ? BM * 4 : 0);
    constexpr int OFF_B = NA * AIMG, OFF_R = OFF_B + 3 * BIMG;
    __shared__ __attribute__((aligned(16))) char smem[OFF_R + 3 * RAWB];             // ONE LDS object (a second one can make hipcc wait vmcnt(0) before every ds_read)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave % WM, wn = wave / WM;
    const int r = lane & 31, h = lane >> 5;
    int mi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX; ++i) if (i < a.n && (int) blockIdx.y >= a.tile_end[i - 1]) mi = i;
    const int64_t aM = a.M[mi];
    const unsigned char * const aw = a.w[mi];
    float * const ay = a.y[mi];
    const int64_t m0 = (int64_t) ((int) blockIdx.y - (mi ? a.tile_end[mi - 1] : 0)) * BM, n0 = (int64_t) blockIdx.x * BN;
    const int nsteps = (int) (a.K / BK);

    // ---- LDS-DMA roles (as the second form): instruction i of a wave fills 8 rows of a tile image, lane l -> row + l / 8, chunk position l % 8 = source chunk (l % 8) ^ swz(row)
    constexpr int NIB = BN / 8 / NWAVE, NIA = ADMA ? BM / 8 / NWAVE : 0;
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(xh) + n0 * a.K, 0, (int) (min((int64_t) BN, a.N - n0) * a.K * 2), 0x00020000);
    int vo_b[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int row = (wave * NIB + i) * 8 + (lane >> 3);
        vo_b[i] = (int) (min((int64_t) row, a.N - 1 - n0) * a.K * 2) + (((lane & 7) ^ swz(row)) << 4);
    }
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(aw) + m0 * a.row_bytes, 0, (int) (min((int64_t) BM, aM - m0) * a.row_bytes), 0x00020000);
    [[maybe_unused]] int vo_a[ADMA ? NIA : 1];
    if constexpr (ADMA) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int row = (wave * NIA + i) * 8 + (lane >> 3);
            vo_a[i] = (int) (min((int64_t) row, aM - 1 - m0) * a.row_bytes) + (((lane & 7) ^ swz(row)) << 4);
        }
    }
    // raw q4_K granules of a step: instruction j (0 .. 3 BM / 64) = plane j / (BM / 64), rows 64 (j % (BM / 64)) + lane.  Wave w issues instructions w, w + NWAVE, ...
    constexpr int NRG = NPL * BM / 64;                                               // 16-byte granule instructions per step
    constexpr int NRI = ADMA ? 0 : NRG + (Q6 ? BM / 64 : 0), NRW = (NRI + NWAVE - 1) / NWAVE;     // + the 4-byte d plane; per step, per wave (the last round may be short)
    const int gpr = (int) (a.K >> 6);
    [[maybe_unused]] int vo_r[NRW > 0 ? NRW : 1];
    if constexpr (!ADMA) {
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int j = wave + i * NWAVE, plane = j / (BM / 64), row = 64 * (j % (BM / 64)) + lane;
            const int rb = (int) (min((int64_t) row, aM - 1 - m0) * a.row_bytes);
            if constexpr (Q6) vo_r[i] = rb + (plane < 6 ? (plane >> 1) * 16 * gpr + (plane & 1) * 16 : (plane == 6 ? 48 * gpr : 52 * gpr));   // la / lb / hh (is = plane & 1), scales, d
            else              vo_r[i] = rb + plane * 16 * gpr;       // (+ 16 g for qa / qb, 16 (g >> 2) for the header: the step's soffset)
        }
    }
    const int n_raw = ADMA ? 0 : (NRI - wave + NWAVE - 1) / NWAVE;                     // raw instructions this wave issues per step (wave-uniform): NRW or NRW - 1
    auto dma_tiles = [&](int step, int slot) {            // B (and f16 A) of `step` into image ring slot `slot`
#if defined(__HIP_DEVICE_COMPILE__)
        const int so = step * (BK * 2);
        if constexpr (ADMA) {
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *) (smem + slot * AIMG + (wave * NIA + i) * 1024), 16, vo_a[i], so, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *) (smem + OFF_B + slot * BIMG + (wave * NIB + i) * 1024), 16, vo_b[i], so, 0, 0);
#endif
    };
    auto dma_raw = [&](int step, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (!ADMA) {
#pragma unroll
            for (int i = 0; i < NRW; ++i) {
                const int j = wave + i * NWAVE;            // wave-uniform
                if (j < NRI) {
                    const int plane = j / (BM / 64);
                    if constexpr (Q6) {
                        const int b = step >> 2, it0 = 4 * b + 2 * ((step >> 1) & 1);                     // block, first item of its half n
                        if (j < NRG) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *) (smem + OFF_R + slot * RAWB + j * 1024), 16, vo_r[i],
                                                                              plane < 6 ? it0 * 16 : b * 16, 0, 0);
                        else         __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *) (smem + OFF_R + slot * RAWB + NRG * 1024 + (j - NRG) * 256), 4, vo_r[i],
                                                                              (2 * b) & ~3, 0, 0);        // the aligned dword around the block's d
                    } else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *) (smem + OFF_R + slot * RAWB + j * 1024), 16, vo_r[i],
                                                             plane < 2 ? step * 16 : (step >> 2) * 16, 0, 0);
                }
            }
        }
#endif
    };
    // conversion of a landed raw buffer into a weight image: IPR threads per row (2: a 32-element half each; 4, where the tile has fewer rows than half the
    // threads: a 16-element quarter each — qa or qb of the row, low or high nibbles).  The raw bytes come back through ds_reads hipcc does not see (loads and
    // their wait in ONE asm statement, early-clobber outputs: cdna_hip_programming.md §5.7 form (i)): for a ds_read it can see while LDS-DMA is in flight it
    // waits vmcnt(0) first — it cannot tell that the bytes being read landed a step ago — and that drains the transfers this step has just requested (seen in
    // the ISA of the first build of this kernel); the converted chunks leave through asm ds_writes for the same reason.
    constexpr int IPR = (ADMA || Q6) ? 2 : (4 * BM <= NT ? 4 : 2);
    const int srow = tid / IPR, sq = tid % IPR;
    auto convert = [&](int step, int rslot, int aslot) {
        if constexpr (!ADMA) {
            if (srow < BM) {
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned ra = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char *) (smem + OFF_R + rslot * RAWB + srow * 16);   // LDS byte addresses
                const unsigned wa = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char *) (smem + aslot * AIMG + srow * ROWB2);
#else
                const unsigned ra = 0, wa = 0;
#endif
                const int key = swz(srow);
                if constexpr (Q6) {   // half sq of the step = sub-block j = 2 m + sq of half n: la (sq 0) or lb (sq 1), both `is` items; hh; scales 8 n + 2 j, + 1; d
                    const int m = step & 1, n = (step >> 1) & 1, j = 2 * m + sq;
                    u4v l0, l1, h0, h1, sc;
                    unsigned dd;
                    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:%9\n\tds_read_b128 %2, %7 offset:%10\n\tds_read_b128 %3, %7 offset:%11\n\t"
                                 "ds_read_b128 %4, %7 offset:%12\n\tds_read_b32 %5, %8 offset:%13\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(l0), "=&v"(l1), "=&v"(h0), "=&v"(h1), "=&v"(sc), "=&v"(dd)
                                 : "v"(ra + (unsigned) (sq * 2 * BM * 16)), "v"(ra), "v"(ra - (unsigned) (srow * 12)),
                                   "n"(BM * 16), "n"(4 * BM * 16), "n"(5 * BM * 16), "n"(6 * BM * 16), "n"(7 * BM * 16) : "memory");
                    RawW<PKV_TYPE_Q6_K> rw;
                    rw.ql[0] = make_uint4(l0.x, l0.y, l0.z, l0.w); rw.ql[1] = make_uint4(l1.x, l1.y, l1.z, l1.w);
                    rw.qh[0] = make_uint4(h0.x, h0.y, h0.z, h0.w); rw.qh[1] = make_uint4(h1.x, h1.y, h1.z, h1.w);
                    const unsigned scw[4] = {sc.x, sc.y, sc.z, sc.w};
                    const int si = 8 * n + 2 * j;                                                          // even: both scales sit in one dword
                    rw.sc2 = (uint16_t) ((scw[si >> 2] >> (8 * (si & 3))) & 0xFFFF);
                    rw.d   = (uint16_t) (((step >> 2) & 1) ? (dd >> 16) : (dd & 0xFFFF));
                    uint4 o[4];
                    Deq<PKV_TYPE_Q6_K>::run(rw, (int64_t) step * BK + 32 * sq, o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u4v x = {o[i].x, o[i].y, o[i].z, o[i].w};
                        asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" :: "v"(wa + (unsigned) (((4 * sq + i) ^ key) << 4)), "v"(x) : "memory");
                    }
                } else
                if constexpr (IPR == 2) {
                    u4v r0, r1, r2;
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%4\n\tds_read_b128 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(ra), "n"(BM * 16), "n"(2 * BM * 16) : "memory");
                    RawW<PKV_TYPE_Q4_K> rw;
                    rw.q[0] = make_uint4(r0.x, r0.y, r0.z, r0.w); rw.q[1] = make_uint4(r1.x, r1.y, r1.z, r1.w); rw.h = make_uint4(r2.x, r2.y, r2.z, r2.w);
                    uint4 o[4];
                    Deq<PKV_TYPE_Q4_K>::run(rw, (int64_t) step * BK + 32 * sq, o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u4v x = {o[i].x, o[i].y, o[i].z, o[i].w};
                        asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" :: "v"(wa + (unsigned) (((4 * sq + i) ^ key) << 4)), "v"(x) : "memory");
                    }
                } else {   // quarter sq = 2 hsel + pl: the 16 bytes of plane pl (qa / qb), nibble hsel -> elements 32 hsel + 16 pl + [0, 16) of the step
                    const int hsel = sq >> 1, pl = sq & 1;
                    u4v qv, hv;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(qv), "=&v"(hv) : "v"(ra + (unsigned) (pl * BM * 16)), "v"(ra), "n"(2 * BM * 16) : "memory");
                    const int j = 2 * (step & 3) + hsel;                                                    // sub-block of the 256-block
                    const float d = hf((uint16_t) (hv.x & 0xFFFF)), dmin = hf((uint16_t) (hv.x >> 16));
                    const int sh = 8 * (j & 3);
                    const int b0 = (int) ((hv.y >> sh) & 0xFF), b1 = (int) ((hv.z >> sh) & 0xFF), b2 = (int) ((hv.w >> sh) & 0xFF);   // get_scale_min_k4
                    const int sc = j < 4 ? (b0 & 63) : ((b2 & 0xF) | ((b0 >> 6) << 4)), mn = j < 4 ? (b1 & 63) : ((b2 >> 4) | ((b1 >> 6) << 4));
                    const h2v d1 = splat(d * (float) sc), m1 = splat(-(dmin * (float) mn)), nb = splat(-1024.0f);
                    const int s4 = 4 * hsel;
                    uint32_t e[8];
                    cvt4((qv.x >> s4) & 0x0F0F0F0Fu, nb, d1, m1, e[0], e[1]); cvt4((qv.y >> s4) & 0x0F0F0F0Fu, nb, d1, m1, e[2], e[3]);
                    cvt4((qv.z >> s4) & 0x0F0F0F0Fu, nb, d1, m1, e[4], e[5]); cvt4((qv.w >> s4) & 0x0F0F0F0Fu, nb, d1, m1, e[6], e[7]);
                    const u4v x0 = {e[0], e[1], e[2], e[3]}, x1 = {e[4], e[5], e[6], e[7]};
                    const int c0 = 4 * hsel + 2 * pl;
                    asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" :: "v"(wa + (unsigned) (((c0) ^ key) << 4)), "v"(x0) : "memory");
                    asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" :: "v"(wa + (unsigned) (((c0 + 1) ^ key) << 4)), "v"(x1) : "memory");
                }
            }
        }
    };
    // wait until only this step's requests are in flight, then the barrier (raw s_barrier: __syncthreads() would drain the DMA)
    auto wait_and_barrier = [&]() {
        constexpr int N0 = NIB + NIA;
        if constexpr (ADMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N0) : "memory");
        else {
            if (n_raw == NRW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N0 + NRW) : "memory");
            else              asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N0 + NRW - 1) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto clampst = [&](int st) { return st < nsteps ? st : nsteps - 1; };   // past the last step: a valid re-load into a slot nobody reads again (keeps the counts)

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const int a_ro = (wm * 32 * TM + r) * ROWB2, b_ro = OFF_B + (wn * 32 * TN + r) * ROWB2, fkey = swz(r);

    // prologue: raw 0, 1, 2 and tiles 0, 1 requested; everything landed; weight image 0 converted
    dma_raw(0, 0); dma_raw(clampst(1), 1); dma_raw(clampst(2), 2);
    dma_tiles(0, 0); dma_tiles(clampst(1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    convert(0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int s3 = 0, s2 = 0;                                    // k % 3, k % 2
    for (int k = 0; k < nsteps; ++k) {
        const int s3n = s3 == 2 ? 0 : s3 + 1, s3nn = s3n == 2 ? 0 : s3n + 1;
        dma_raw(clampst(k + 3), s3);                       // raw(k + 3) -> raw slot (k + 3) % 3 = k % 3 (raw(k) was converted at step k - 1)
        dma_tiles(clampst(k + 2), s3nn);                   // tiles(k + 2) -> slot (k + 2) % 3 (last read at step k - 1)
        if (k + 1 < nsteps) convert(k + 1, s3n, s2 ^ 1);   // raw(k + 1) (landed before the last barrier) -> weight image (k + 1) % 2
        const char * sA = smem + (ADMA ? s3 : s2) * AIMG;
        const char * sB = smem + s3 * BIMG;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8_t af[TM], bf[TN];
            const int po = (((2 * ks + h) ^ fkey) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8_t *>(sA + a_ro + i * 32 * ROWB2 + po);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8_t *>(sB + b_ro + j * 32 * ROWB2 + po);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        wait_and_barrier();
        s3 = s3n; s2 ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tail's re-loads
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn * 32 * TN + j * 32 + r;
        if (n >= a.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t m = m0 + wm * 32 * TM + i * 32 + 8 * g + 4 * h;
                if ((aM & 3) == 0 && m + 3 < aM) *reinterpret_cast<float4 *>(ay + (size_t) n * aM + m) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                else
                    for (int e = 0; e < 4; ++e) if (m + e < aM) ay[(size_t) n * aM + m + e] = acc[i][j][4 * g + e];
            }
    }
}

// activations f32 -> f16, once per product
__global__ void __launch_bounds__(256) cvt_f16_kernel(const float4 * __restrict__ x, uint2 * __restrict__ o, int64_t n4) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t) gridDim.x * 256) {
        const float4 v = x[i];
        const _Float16 hv[4] = {(_Float16) v.x, (_Float16) v.y, (_Float16) v.z, (_Float16) v.w};
        o[i] = *reinterpret_cast<const uint2 *>(hv);
    }
}

}  // namespace

bool gemm_supported(int type_w, int64_t M, int64_t K) {
    type_w &= ~PKV_TYPE_PACKED;
    if (type_w != PKV_TYPE_F16 && type_w != PKV_TYPE_Q8_0 && type_w != PKV_TYPE_Q4_K && type_w != PKV_TYPE_Q5_K && type_w != PKV_TYPE_Q6_K) return false;
    return M > 0 && K > 0 && K % BK == 0 && (type_w == PKV_TYPE_Q4_K || type_w == PKV_TYPE_Q5_K || type_w == PKV_TYPE_Q6_K ? K % 256 == 0 : true);
}

// ---- per-stream scratch of the f16 activations (grown on demand; hipFree of the old buffer synchronises) ----
namespace {
struct XhScratch { void * p = nullptr; size_t cap = 0; };
std::mutex g_xh_mu;
std::unordered_map<hipStream_t, XhScratch> g_xh;
int xh_scratch(hipStream_t st, size_t bytes, _Float16 ** out) {
    std::lock_guard<std::mutex> lk(g_xh_mu);
    XhScratch & s = g_xh[st];
    if (s.cap < bytes) {
        if (s.p) PKV_HIP(hipFree(s.p));
        s.p = nullptr; s.cap = 0;
        const size_t want = (bytes + (size_t) (8 << 20)) & ~(size_t) ((1 << 20) - 1);
        PKV_HIP(hipMalloc(&s.p, want));
        s.cap = want;
    }
    *out = (_Float16 *) s.p;
    return PKV_OK;
}

// second form (gemm2_kernel): PKV_OK and *done = true when it took the product
int launch_gemm2(GemmArgs & a, int type_w, int64_t n_cols, hipStream_t st, bool * done) {
    *done = false;
    if (getenv("PKV_GEMM_V1")) return PKV_OK;                  // A/B switch: the first form for everything
    if (n_cols < 64 || (a.K % BK) != 0 || a.K * 2 * 256 >= ((int64_t) 1 << 31)) return PKV_OK;
    auto tiles = [&](int bm, int bn) { int64_t t = 0; for (int i = 0; i < a.n; ++i) t += (a.M[i] + bm - 1) / bm; return t * ((n_cols + bn - 1) / bn); };
    // the largest tile that still fills the chip: 256 x 256 / 256 x 128 run one 8-wave workgroup per CU, 128 x 128 / 64 x 128 two 4-wave workgroups
    int shape;
    if      (tiles(256, 256) >= 200) shape = 0;
    else if (tiles(256, 128) >= 200) shape = 1;
    else if (tiles(128, 128) >= 384) shape = 2;
    else                             shape = 3;
    if (const char * e = getenv("PKV_GEMM2_SHAPE")) shape = atoi(e) & 3;   // experiments
    const int bm = shape <= 1 ? 256 : (shape == 2 ? 128 : 64), bn = shape == 0 ? 256 : 128;
    int tot = 0;
    for (int i = 0; i < a.n; ++i) { tot += (int) ((a.M[i] + bm - 1) / bm); a.tile_end[i] = tot; }
    _Float16 * xh = nullptr;
    const int rc = xh_scratch(st, (size_t) n_cols * a.K * 2, &xh);
    if (rc != PKV_OK) return rc;
    const int64_t n4 = n_cols * a.K / 4;
    hipLaunchKernelGGL(cvt_f16_kernel, dim3((unsigned) std::min<int64_t>((n4 + 255) / 256, 2048)), dim3(256), 0, st, (const float4 *) a.x, (uint2 *) xh, n4);
    a.ksplit = 1;
    const dim3 grid((unsigned) ((n_cols + bn - 1) / bn), (unsigned) tot);
    // third form (transfers two steps ahead) for the shapes with short K-steps; the 256 x 256 tile stays on the second form (three of its images do not fit LDS)
    // packed q6_K: its raw ring (7 planes) does not fit beside a 256-row tile and leaves room for ONE workgroup per CU — it pays where the launch is a single round of
    // workgroups anyway (4096 x 4096 x 512: 199 -> 243 TFLOP/s, 4096 x 14336 x 512: 202 -> 268; 6144 rows = 384 workgroups: 280 -> 193, stays on the second form);
    // rows 16-byte aligned: K % 2048 == 0 (pkv_wlayout.h)
    const bool q6p = type_w == (PKV_TYPE_Q6_K | PKV_TYPE_PACKED);
    const bool q6deep = q6p && shape >= 2 && a.K % 2048 == 0 && (int64_t) tot * ((n_cols + bn - 1) / bn) <= 256;
    const bool deep = shape != 0 && (type_w == PKV_TYPE_F16 || type_w == (PKV_TYPE_Q4_K | PKV_TYPE_PACKED) || q6deep) && a.K / BK >= 4 && !getenv("PKV_GEMM_NO_DEEP");
    if (deep) {
#define PKV_GEMM3(T_) do { if (shape == 1) hipLaunchKernelGGL((gemm3_kernel<T_, 256, 128, 4, 2>), grid, dim3(512), 0, st, a, xh); \
                           else if (shape == 2) hipLaunchKernelGGL((gemm3_kernel<T_, 128, 128, 2, 2>), grid, dim3(256), 0, st, a, xh); \
                           else hipLaunchKernelGGL((gemm3_kernel<T_, 64, 128, 1, 4>), grid, dim3(256), 0, st, a, xh); } while (0)
        if (type_w == PKV_TYPE_F16) PKV_GEMM3(PKV_TYPE_F16);
        else if (q6p) {
            if (shape == 2) hipLaunchKernelGGL((gemm3_kernel<(PKV_TYPE_Q6_K | PKV_TYPE_PACKED), 128, 128, 2, 2>), grid, dim3(256), 0, st, a, xh);
            else            hipLaunchKernelGGL((gemm3_kernel<(PKV_TYPE_Q6_K | PKV_TYPE_PACKED), 64, 128, 1, 4>), grid, dim3(256), 0, st, a, xh);
        } else PKV_GEMM3((PKV_TYPE_Q4_K | PKV_TYPE_PACKED));
#undef PKV_GEMM3
        PKV_HIP(hipGetLastError());
        *done = true;
        return PKV_OK;
    }
#define PKV_GEMM2(T_) do { if (shape == 0) hipLaunchKernelGGL((gemm2_kernel<T_, 256, 256, 2, 4>), grid, dim3(512), 0, st, a, xh); \
                           else if (shape == 1) hipLaunchKernelGGL((gemm2_kernel<T_, 256, 128, 4, 2>), grid, dim3(512), 0, st, a, xh); \
                           else if (shape == 2) hipLaunchKernelGGL((gemm2_kernel<T_, 128, 128, 2, 2>), grid, dim3(256), 0, st, a, xh); \
                           else hipLaunchKernelGGL((gemm2_kernel<T_, 64, 128, 1, 4>), grid, dim3(256), 0, st, a, xh); } while (0)
    switch (type_w) {
        case PKV_TYPE_F16:  PKV_GEMM2(PKV_TYPE_F16); break;
        case PKV_TYPE_Q8_0: PKV_GEMM2(PKV_TYPE_Q8_0); break;
        case PKV_TYPE_Q4_K: PKV_GEMM2(PKV_TYPE_Q4_K); break;
        case PKV_TYPE_Q5_K: PKV_GEMM2(PKV_TYPE_Q5_K); break;
        case PKV_TYPE_Q6_K: PKV_GEMM2(PKV_TYPE_Q6_K); break;
        case PKV_TYPE_Q4_K | PKV_TYPE_PACKED: PKV_GEMM2((PKV_TYPE_Q4_K | PKV_TYPE_PACKED)); break;
        case PKV_TYPE_Q5_K | PKV_TYPE_PACKED: PKV_GEMM2((PKV_TYPE_Q5_K | PKV_TYPE_PACKED)); break;
        case PKV_TYPE_Q6_K | PKV_TYPE_PACKED: PKV_GEMM2((PKV_TYPE_Q6_K | PKV_TYPE_PACKED)); break;
        default: return PKV_OK;
    }
#undef PKV_GEMM2
    PKV_HIP(hipGetLastError());
    *done = true;
    return PKV_OK;
}
}  // namespace

int launch_gemm_multi(int n, const void * const * w, int type_w, const float * x, float * const * y, const int64_t * M, int64_t K, int64_t n_cols, hipStream_t st) {
    if (n_cols == 0) return PKV_OK;
    PKV_REQUIRE(n >= 1 && n <= GEMM_MAX, "pkv_mul_mat (batched): 1 to %d matrices per launch", GEMM_MAX);
    if (!gemm_supported(type_w, 1, K)) { set_error("pkv_mul_mat_batched: weight type %d / K %lld not supported (f16, q8_0, q4_K, q5_K, q6_K; K %% 64 == 0)", type_w, (long long) K); return PKV_ERR_UNSUPPORTED; }
    PKV_REQUIRE(((uintptr_t) x % 16) == 0, "pkv_mul_mat (batched): x must be 16-byte aligned");
    GemmArgs a = {};
    a.x = x; a.N = n_cols; a.K = K; a.row_bytes = (int64_t) row_size(type_w & ~PKV_TYPE_PACKED, K);
    int64_t r128 = 0, r64 = 0;
    for (int i = 0; i < n; ++i) {
        if (M[i] == 0) continue;
        PKV_REQUIRE(((uintptr_t) y[i] % 16) == 0, "pkv_mul_mat (batched): y must be 16-byte aligned");
        a.w[a.n] = (const unsigned char *) w[i]; a.y[a.n] = y[i]; a.M[a.n] = M[i]; ++a.n;
        r128 += (M[i] + 127) / 128; r64 += (M[i] + 63) / 64;
    }
    if (a.n == 0) return PKV_OK;
    {
        bool done = false;
        const int rc = launch_gemm2(a, type_w, n_cols, st, &done);
        if (rc != PKV_OK || done) return rc;
    }
    // 128 x 128 tiles; 64 x 128, then 64 x 64, where the larger shape would leave CUs without a tile (or with one: a single 4-wave workgroup per CU
    // runs at about half the rate of two)
    const int64_t t128 = r128 * ((n_cols + 127) / 128), t64 = r64 * ((n_cols + 127) / 128);
    // (64 x 64 doubles the dequantization per MFMA: it pays for f16 and q4_K weights, +21 ... +29 % at 4096 rows x 512 columns, and loses 10-14 % for
    // q8_0 / q6_K, whose conversion is a float multiply per element)
    const bool cheap = type_w == PKV_TYPE_F16 || (type_w & ~PKV_TYPE_PACKED) == PKV_TYPE_Q4_K;
    // (... unless even 64-row tiles leave the chip short of one workgroup per CU: TinyLlama's 2048-row matrices, q8_0 71 -> 87, q6_K 54 -> 70 TFLOP/s)
    int shape = t128 >= 256 ? 0 : (t64 >= 256 ? 1 : 2);   // (with two LDS images per operand 64 x 128 is as fast as 64 x 64 was for f16 / q4_K at 4096 rows, and 25-45 % faster for q8_0 / q6_K)
    (void) cheap;
    if (const char * e = getenv("PKV_GEMM_SHAPE")) shape = atoi(e) % 3;   // experiments
    const int bm = shape == 0 ? 128 : 64, bn = shape == 2 ? 64 : 128;
    int tot = 0;
    for (int i = 0; i < a.n; ++i) { tot += (int) ((a.M[i] + bm - 1) / bm); a.tile_end[i] = tot; }
    // a long K over very few tiles (16 - 64 columns: 64 workgroups walking all of K at one load latency per step): two workgroups per tile, each
    // half of K, added into the zeroed output (two addends per element: the result does not depend on their order): 4096^2 x 32, 65 -> 40 us.
    // Not at hundreds of columns: there the 4 M atomic adds cost more than the idle CUs did (4096^2 x 512: 84 -> 175 us).
    a.ksplit = 1;
    if (shape == 2 && n_cols <= 64 && (int64_t) tot * ((n_cols + bn - 1) / bn) <= 768 && K >= 4096 && K % 512 == 0 && !getenv("PKV_GEMM_NO_SPLITK")) {
        a.ksplit = 2;
        for (int i = 0; i < a.n; ++i) PKV_HIP(hipMemsetAsync(a.y[i], 0, (size_t) a.M[i] * n_cols * sizeof(float), st));
    }
    const dim3 grid((unsigned) tot, (unsigned) ((n_cols + bn - 1) / bn), (unsigned) a.ksplit);
#define PKV_GEMM(T_) do { if (shape == 0) hipLaunchKernelGGL((gemm_kernel<T_, 128, 128>), grid, dim3(256), 0, st, a); \
                          else if (shape == 1) hipLaunchKernelGGL((gemm_kernel<T_, 64, 128>), grid, dim3(256), 0, st, a); \
                          else hipLaunchKernelGGL((gemm_kernel<T_, 64, 64>), grid, dim3(256), 0, st, a); } while (0)
    switch (type_w) {
        case PKV_TYPE_F16:  PKV_GEMM(PKV_TYPE_F16); break;
        case PKV_TYPE_Q8_0: PKV_GEMM(PKV_TYPE_Q8_0); break;
        case PKV_TYPE_Q4_K: PKV_GEMM(PKV_TYPE_Q4_K); break;
        case PKV_TYPE_Q5_K: PKV_GEMM(PKV_TYPE_Q5_K); break;
        case PKV_TYPE_Q6_K: PKV_GEMM(PKV_TYPE_Q6_K); break;
        case PKV_TYPE_Q4_K | PKV_TYPE_PACKED: PKV_GEMM((PKV_TYPE_Q4_K | PKV_TYPE_PACKED)); break;
        case PKV_TYPE_Q5_K | PKV_TYPE_PACKED: PKV_GEMM((PKV_TYPE_Q5_K | PKV_TYPE_PACKED)); break;
        case PKV_TYPE_Q6_K | PKV_TYPE_PACKED: PKV_GEMM((PKV_TYPE_Q6_K | PKV_TYPE_PACKED)); break;
    }
#undef PKV_GEMM
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_gemm(const void * w, int type_w, const float * x, float * y, int64_t M, int64_t K, int64_t n_cols, hipStream_t st) {
    if (M == 0 || n_cols == 0) return PKV_OK;
    return launch_gemm_multi(1, &w, type_w, x, &y, &M, K, n_cols, st);
}

}  // namespace pkv
