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
#include "pkv_wlayout.h"

#include <algorithm>
#include <cstdlib>

// every instantiation of the product kernel must round a given row identically (one launch for wq + wk + wv == three launches, bit for bit):
// no implicit FMA contraction anywhere in this file
#pragma clang fp contract(off)

namespace pkv {
namespace {

__device__ __forceinline__ float h2f(uint16_t b) { return (float) *reinterpret_cast<const _Float16 *>(&b); }

// ---- activation column -> q8_K, the reference's own operand of a q4_K / q6_K product ---------------------------------------------
// quantize_row_q8_K_ref (/ref/ggml/src/ggml-quants.c:2479-2516; the x86 CPU backend runs exactly this, ggml-cpu-quants.c:1746): per 256
// elements iscale = -127 / max (max = the signed value of largest magnitude), q = nearest_int(iscale * x), d = 1 / iscale, plus the sums of
// the quants.  Image: [K int8 q][K/256 f32 d][K/32 i32 sum of q per 32].  The integer dots of pkv_gemv_q then see the reference's integers:
// a product differs from the CPU backend's only by the order of a handful of f32 additions.  (Which of two elements of equal magnitude
// and opposite sign supplies `max` only flips the sign of d and of every q together: the products q * d are the same bits.)
// One wave per 256-element block, 4 elements per lane.
template <int CTRL> __device__ __forceinline__ float dppq(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_amax(float x) {     // all-reduce max over the wave: DPP inside the rows of 16, permlane swaps across them (no LDS)
    x = fmaxf(x, dppq<0xB1>(x)); x = fmaxf(x, dppq<0x4E>(x)); x = fmaxf(x, dppq<0x141>(x)); x = fmaxf(x, dppq<0x140>(x));
    { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    return x;
}
__device__ __forceinline__ int sum8(int x) {               // sum over aligned groups of 8 lanes, every lane gets it
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);   // row_half_mirror: lane i <-> 7 - i, the other quad
    return x;
}
// The reference takes `max` = the signed value of largest magnitude, so the sign of (d, q) follows that element's sign; here max = +amax
// always: where the reference's max is negative, d and every q of the block come out negated TOGETHER — the products q * d, d * bsum and
// with them every dot are the same bits (the GEMV only ever multiplies d with sums of q).
__device__ __forceinline__ void quant_block_q8k(const float4 v, int lane, uint32_t & packed, float & d, int & bsum32, int & bsum16) {
    const float am = wave_amax(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    const float iscale = am != 0.0f ? -127.0f / am : 0.0f;
    const int q0 = (int) __builtin_rintf(iscale * v.x), q1 = (int) __builtin_rintf(iscale * v.y), q2 = (int) __builtin_rintf(iscale * v.z),
              q3 = (int) __builtin_rintf(iscale * v.w);
    packed = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
    d = am != 0.0f ? 1.0f / iscale : 0.0f;
    int x = q0 + q1 + q2 + q3;
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
    bsum16 = x;                                                          // the quad's 16 elements
    bsum32 = x + __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);   // row_half_mirror: the other quad of the 8 lanes
}
__global__ void __launch_bounds__(256) quant_act_kernel(const float * x, char * img, int64_t K) {
    const int64_t i4 = (int64_t) blockIdx.x * 256 + threadIdx.x;            // float4 index; K % 256 == 0: a wave never straddles the end
    if (i4 * 4 >= K) return;
    uint32_t packed; float d; int bs, bs16;
    quant_block_q8k(reinterpret_cast<const float4 *>(x)[i4], threadIdx.x & 63, packed, d, bs, bs16);
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
    const float * x;                     // img == NULL: quantize this column in the kernel (after rms_norm(x) * norm_w when norm_w != NULL)
    const float * norm_w;
    float        eps;
    float *      norm_out, * col_out;     // optional: the normalised column (x * scale, and * norm_w), written by workgroup 0
    unsigned long long * stamps;          // diagnostic builds only (-DPKV_GEMV_STAMPS): [workgroup][wave][16] clock stamps, else NULL
    int64_t      K;
    int          gpr, gs;                // 64-element groups per row, LDS row stride (odd)
};

// dynamic LDS: one 80-byte RECORD per 64-element group of the column: 16 dwords of int8 quants, then a 4-dword tail
//   TAIL 0 (q4_K rows only):  { sum of the first 32 quants, sum of the second 32, f32 scale d of the group's 256-block, - }
//   TAIL 1 (q6_K rows about): { sums of the four runs of 16 as int16 x 4 in two dwords,            f32 scale d, - }
// A lane reads its group with five 16-byte LDS loads at immediate offsets from ONE address (r02's first layout, dword wi of group g at
// wi * gs + g, cost an address add and a 4-byte ds_read per dword: 38 of the product's ~210 vector instructions per 32 weight bytes); the
// 20-dword stride keeps ds_read_b128 conflict-free over each of its 16-lane groups (MI355X_MICROARCH.md, LDS table).
constexpr int REC = 20;
extern __shared__ __attribute__((aligned(16))) uint32_t g_lds[];

// one 64-element group of a block_q4_K row against the staged column; lane-private
template <int TAIL>
__device__ __forceinline__ float dot_q4k(const uint4 & hdr, const uint4 & qa, const uint4 & qb, int p, const uint32_t * rec) {
    const uint4 x0 = *reinterpret_cast<const uint4 *>(rec), x1 = *reinterpret_cast<const uint4 *>(rec + 4), x2 = *reinterpret_cast<const uint4 *>(rec + 8),
                x3 = *reinterpret_cast<const uint4 *>(rec + 12), tl = *reinterpret_cast<const uint4 *>(rec + 16);
    const uint32_t xq[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
    const float dx = __uint_as_float(tl.z);                                     // the column's scale for this 256-block
    int b0, b1;                                                                 // sums of its quants, per 32
    if constexpr (TAIL == 0) { b0 = (int) tl.x; b1 = (int) tl.y; }
    else { b0 = ((int) (tl.x << 16) >> 16) + ((int) tl.x >> 16); b1 = ((int) (tl.y << 16) >> 16) + ((int) tl.y >> 16); }
    const float d = h2f((uint16_t) (hdr.x & 0xFFFF)) * dx, dmin = h2f((uint16_t) (hdr.x >> 16)) * dx;       // as ggml_vec_dot_q4_K_q8_K forms them
    // get_scale_min_k4 (/ref/ggml/src/ggml-quants.c:631-638) for sub-blocks j = 2 p, 2 p + 1, both at once in the two low bytes of a dword and
    // without a branch: j < 4: sc = q[j] & 63, m = q[j + 4] & 63; j >= 4: sc = (q[j + 4] & 15) | (q[j - 4] >> 6 << 4), m = (q[j + 4] >> 4) | (q[j] >> 6 << 4)
    // (q[0..3] = hdr.y, q[4..7] = hdr.z, q[8..11] = hdr.w; the pair sits at bit 16 (p & 1) of each)
    const uint32_t sh = 16u * (uint32_t) (p & 1);
    const uint32_t ty = hdr.y >> sh, tz = hdr.z >> sh, tw = hdr.w >> sh;
    const uint32_t sc_lo = ty & 0x3F3Fu, mn_lo = tz & 0x3F3Fu;
    const uint32_t sc_hi = (tw & 0x0F0Fu) | ((ty >> 2) & 0x3030u), mn_hi = ((tw >> 4) & 0x0F0Fu) | ((tz >> 2) & 0x3030u);
    const uint32_t sc = p < 2 ? sc_lo : sc_hi, mn = p < 2 ? mn_lo : mn_hi;
    const int s1 = (int) (sc & 0xFF), s2 = (int) (sc >> 8), m1 = (int) (mn & 0xFF), m2 = (int) (mn >> 8);
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

// one 64-element group of a block_q5_K row (176 B: {d, dmin, scales[12]}, qh[32], qs[128], /ref/ggml/src/ggml-common.h:291-308): the q4_K group
// plus its fifth bits — bit 2 p of qh[l] belongs to element 64 p + l, bit 2 p + 1 to element 64 p + 32 + l (dequantize_row_q5_K,
// /ref/ggml/src/ggml-quants.c:1482-1507: u1 = 1 << 2 p, u2 = 2 << 2 p).  ha / hb = the block's 32 bytes of qh (shared by its four groups); the 5-bit
// quants 0..31 enter the integer dots as ggml_vec_dot_q5_K_q8_K forms them (ggml-cpu-quants.c: sum_j scale_j (q5 . q8)_j, sum_j min_j bsum_j).
template <int TAIL>
__device__ __forceinline__ float dot_q5k(const uint4 & hdr, const uint4 & qa, const uint4 & qb, const uint4 & ha, const uint4 & hb, int p, const uint32_t * rec) {
    const uint4 x0 = *reinterpret_cast<const uint4 *>(rec), x1 = *reinterpret_cast<const uint4 *>(rec + 4), x2 = *reinterpret_cast<const uint4 *>(rec + 8),
                x3 = *reinterpret_cast<const uint4 *>(rec + 12), tl = *reinterpret_cast<const uint4 *>(rec + 16);
    const uint32_t xq[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
    const float dx = __uint_as_float(tl.z);
    int b0, b1;
    if constexpr (TAIL == 0) { b0 = (int) tl.x; b1 = (int) tl.y; }
    else { b0 = ((int) (tl.x << 16) >> 16) + ((int) tl.x >> 16); b1 = ((int) (tl.y << 16) >> 16) + ((int) tl.y >> 16); }
    const float d = h2f((uint16_t) (hdr.x & 0xFFFF)) * dx, dmin = h2f((uint16_t) (hdr.x >> 16)) * dx;
    const uint32_t sh = 16u * (uint32_t) (p & 1);                               // get_scale_min_k4, as in dot_q4k
    const uint32_t ty = hdr.y >> sh, tz = hdr.z >> sh, tw = hdr.w >> sh;
    const uint32_t sc_lo = ty & 0x3F3Fu, mn_lo = tz & 0x3F3Fu;
    const uint32_t sc_hi = (tw & 0x0F0Fu) | ((ty >> 2) & 0x3030u), mn_hi = ((tw >> 4) & 0x0F0Fu) | ((tz >> 2) & 0x3030u);
    const uint32_t sc = p < 2 ? sc_lo : sc_hi, mn = p < 2 ? mn_lo : mn_hi;
    const int s1 = (int) (sc & 0xFF), s2 = (int) (sc >> 8), m1 = (int) (mn & 0xFF), m2 = (int) (mn >> 8);
    const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w}, qh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
    int ilo = 0, ihi = 0;
#pragma unroll
    for (int w4 = 0; w4 < 8; ++w4) {           // byte b of dword w4: qs / qh byte 4 w4 + b of the group / block
        const uint32_t h = qh[w4] >> (2 * p);  // 2 p <= 6: bits 0 / 1 of every byte = the group's two fifth bits of that byte
        ilo = __builtin_amdgcn_sdot4((int) ((qw[w4] & 0x0F0F0F0Fu) | ((h & 0x01010101u) << 4)), (int) xq[w4], ilo, false);
        ihi = __builtin_amdgcn_sdot4((int) (((qw[w4] >> 4) & 0x0F0F0F0Fu) | ((h & 0x02020202u) << 3)), (int) xq[8 + w4], ihi, false);
    }
    return d * (float) (s1 * ilo + s2 * ihi) - dmin * (float) (m1 * b0 + m2 * b1);
}

// block_q6_K (210 bytes: every other block sits 2 bytes off a dword boundary — the 16-byte loads below are misaligned, which gfx950 global
// loads serve correctly, probes/unaligned_probe.hip): quarter `qt` = (n, is) of the block -> elements n*128 + 32 j + 16 is + [0, 16), j = 0..3,
// scale index 8 n + is + 2 j (dequantize_row_q6_K, /ref/ggml/src/ggml-quants.c:1690-1719).  la = ql[16 is ..], lb = ql[32 + 16 is ..],
// hh = qh[16 is ..] of half n; sc16 = the block's 16 scales; dbits = its f16 d.  rec = the record of 64-group 4 b + 2 n (TAIL 1).  The 6-bit
// quants enter the dots as they are (0..63); the -32 of the format leaves through the column's sums of 16: sum (u - 32) x = u . x - 32 sum x.
__device__ __forceinline__ float dot_q6k(const uint4 & la4, const uint4 & lb4, const uint4 & hh4, const uint4 & sc16, uint16_t dbits, int qt, const uint32_t * rec) {
    const int n = qt >> 1, is = qt & 1;
    const uint32_t la[4] = {la4.x, la4.y, la4.z, la4.w}, lb[4] = {lb4.x, lb4.y, lb4.z, lb4.w}, hh[4] = {hh4.x, hh4.y, hh4.z, hh4.w};
    const uint32_t scw[4] = {sc16.x, sc16.y, sc16.z, sc16.w};
    int acc = 0;
    float dx = 0.0f;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {           // the two 64-groups of this half of the block: j = 2 gg, 2 gg + 1
        const uint32_t * r = rec + gg * REC;
        const uint4 tl = *reinterpret_cast<const uint4 *>(r + 16);
        if (gg == 0) dx = __uint_as_float(tl.z);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * gg + jj;
            // the run's activations: elements n*128 + 32 j + 16 is + [0, 16) = dwords 8 jj + 4 is + [0, 4) of the group
            const uint4 xv = *reinterpret_cast<const uint4 *>(r + 8 * jj + 4 * is);
            const uint32_t x[4] = {xv.x, xv.y, xv.z, xv.w};
            int idot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t lo = (jj ? lb[w] : la[w]) >> (4 * gg);
                const uint32_t u  = (lo & 0x0F0F0F0Fu) | (((hh[w] >> (2 * j)) & 0x03030303u) << 4);      // 6-bit quants 0..63
                idot = __builtin_amdgcn_sdot4((int) u, (int) x[w], idot, false);
            }
            const uint32_t sw = jj ? tl.y : tl.x;                                                        // sums of the runs 2 jj, 2 jj + 1 of this group
            idot -= 32 * ((int) (sw << (16 * (1 - is))) >> 16);
            const int si = 8 * n + is + 2 * j;                                                           // scale of this run of 16
            const int scv = (int) (signed char) ((scw[si >> 2] >> (8 * (si & 3))) & 0xFF);
            acc += scv * idot;
        }
    }
    return h2f(dbits) * dx * (float) acc;                      // ggml_vec_dot_q6_K_q8_K: d = x.d * y.d, sum_is scale * (q6 . q8)
}

// wave-wide sum without LDS: DPP inside the rows of 16, permlane swaps across them
template <int CTRL> __device__ __forceinline__ float dppf(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wsum_dpp(float x) {
    x += dppf<0xB1>(x); x += dppf<0x4E>(x); x += dppf<0x141>(x); x += dppf<0x140>(x);        // quad_perm, quad_perm, row_half_mirror, row_mirror
    { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return x;
}

// One unit of a wave's work = 128 (row, 64-element group) items of a row pair: the lane's items u = 0, 1.  Its loads (MODE 12: q4_K rows
// only, 3 x 16 bytes per item; MODE 13: q5_K only, 5 x 16; MODE 14: q6_K only, 4 x 16 + 2; MODE 1: any mix that holds q5_K rows (5 x 16 + 2 for every lane); MODE 0: any mix of q4_K / q6_K, the q6_K shape for every lane so that the number of loads
// in flight does not depend on the lane's type — a q4_K lane re-reads its header) are issued TWO units ahead of the dots that consume them.
// Plain named registers per pipeline stage (a struct handed to lambdas by reference ended up in scratch memory: 368 bytes per lane and
// every weight load doubled by a scratch store — rocprofv3 Scratch_Size column).
// Weight rows are read once by one CU: non-temporal loads (MI355X_MICROARCH.md price list, row nt-weights).
#ifndef PKV_GEMV_STAMPS
#define PKV_GEMV_STAMPS 0   // diagnostic build (probes/gemv_stamps.py): per-wave s_memtime stamps kept in registers, written once at the end; never in the product library
#endif
#if PKV_GEMV_STAMPS
#define PKV_STAMP(i) { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PKV_STAMP(i)
#endif
#ifndef PKV_GEMV_ABLATE
#define PKV_GEMV_ABLATE 0   // timing-only builds (results wrong): 1 = no dots (loads + reduce), 2 = no column prologue
#endif
typedef unsigned gv_u4 __attribute__((ext_vector_type(4)));
#ifndef PKV_GEMV_NT
#define PKV_GEMV_NT 1       // packed rows only: every 128-byte line is consumed by ONE wave instruction (GGUF-order rows: 2-3 instructions per line, nt lost 40 %)
#endif
template <bool NT> __device__ __forceinline__ uint4 ld16w(const unsigned char * p) {
    if constexpr (NT) {
        const gv_u4 v = __builtin_nontemporal_load(reinterpret_cast<const gv_u4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4 *>(p);
    }
}
#define PKV_UNIT_DECL(S) uint4 S##r0[UI], S##r1[UI], S##r2[UI], S##r3[UI], S##r4[UI]; uint16_t S##dh[UI]; int S##g[UI], S##p[UI], S##ty[UI], S##rs[UI]; bool S##ok[UI]; int S##mi; int64_t S##m0;
#define PKV_UNIT_LOAD(S, RPI, FI, PAST)                                                                                                           \
    {                                                                                                                                          \
        const int rp_ = first + __builtin_amdgcn_readfirstlane(RPI) * stride, f0_ = __builtin_amdgcn_readfirstlane(FI) * (64 * UI);            \
        const char * q0_, * q1_; int t0_, t1_, mi_; int64_t m0_;                                                                               \
        rows_of(rp_, q0_, q1_, t0_, t1_, mi_, m0_);                                                                                            \
        S##mi = mi_; S##m0 = m0_;                                                                                                              \
        _Pragma("unroll") for (int u = 0; u < UI; ++u) {                                                                                        \
            const int f = f_lo + f0_ + 64 * u + lane;                                                                                          \
            S##ok[u] = f < f_hi;                                                                                                               \
            const int fc = S##ok[u] ? f : f_hi - 1;                                                                                            \
            S##rs[u] = fc >= gpr;                                                                                                              \
            S##g[u] = fc - S##rs[u] * gpr;                                                                                                     \
            S##p[u] = S##g[u] & 3;                                                                                                             \
            S##ty[u] = MODE > 1 ? MODE : (S##rs[u] ? t1_ : t0_);                                                                               \
            const unsigned char * rw_ = reinterpret_cast<const unsigned char *>(S##rs[u] ? q1_ : q0_);                                         \
            const int ga_ = (PAST) ? 0 : S##g[u], pa_ = ga_ & 3;   /* a request past the wave's last unit: every lane the row's first 16 bytes (one line) */ \
            if constexpr (PK) {   /* plane layout (pkv_wlayout.h): lane-consecutive items = consecutive 16 bytes of each plane */              \
                const bool k4 = MODE == PKV_TYPE_Q4_K || (MODE == 0 && S##ty[u] == PKV_TYPE_Q4_K);                                             \
                const unsigned char * it_ = rw_ + 16 * ga_, * bl_ = rw_ + 16 * (ga_ >> 2);                                                           \
                if (MODE == 1) {   /* any mix with q5_K rows: five loads + the 2-byte one for every lane (a q4_K / q6_K lane re-reads its header / scales) */ \
                    const bool k4_ = S##ty[u] == PKV_TYPE_Q4_K, k5_ = S##ty[u] == PKV_TYPE_Q5_K, k45_ = k4_ || k5_;                             \
                    const unsigned char * hd_ = bl_ + (k4_ ? 32 : (k5_ ? 40 : 48)) * gpr;   /* header plane (q6_K: the scale plane) */          \
                    S##r0[u] = ld16w<NTL>(k45_ ? hd_ : it_);                                                                                   \
                    S##r1[u] = ld16w<NTL>(k45_ ? it_ : it_ + 16 * gpr);                                                                        \
                    S##r2[u] = ld16w<NTL>(k45_ ? it_ + 16 * gpr : it_ + 32 * gpr);                                                             \
                    S##r3[u] = ld16w<NTL>(k5_ ? bl_ + 32 * gpr : hd_);                                                                         \
                    S##r4[u] = ld16w<NTL>(k5_ ? bl_ + 36 * gpr : hd_);                                                                         \
                    S##dh[u] = *reinterpret_cast<const uint16_t *>(k45_ ? hd_ : rw_ + 52 * gpr + 2 * (ga_ >> 2));                              \
                } else                                                                                                                         \
                if (MODE == PKV_TYPE_Q5_K) {   /* planes: qa, qb per item; ha, hb, header per block */                                          \
                    S##r0[u] = ld16w<NTL>(bl_ + 40 * gpr);                                                                                     \
                    S##r1[u] = ld16w<NTL>(it_);                                                                                                \
                    S##r2[u] = ld16w<NTL>(it_ + 16 * gpr);                                                                                     \
                    S##r3[u] = ld16w<NTL>(bl_ + 32 * gpr);                                                                                     \
                    S##r4[u] = ld16w<NTL>(bl_ + 36 * gpr); S##dh[u] = 0;                                                                       \
                } else                                                                                                                         \
                if (MODE == PKV_TYPE_Q4_K) {                                                                                                   \
                    S##r0[u] = ld16w<NTL>(bl_ + 32 * gpr);                                                                                     \
                    S##r1[u] = ld16w<NTL>(it_);                                                                                                \
                    S##r2[u] = ld16w<NTL>(it_ + 16 * gpr);                                                                                     \
                    S##r3[u] = make_uint4(0, 0, 0, 0); S##dh[u] = 0;                                                                           \
                } else {                                                                                                                       \
                    S##r0[u] = ld16w<NTL>(k4 ? bl_ + 32 * gpr : it_);                                                                          \
                    S##r1[u] = ld16w<NTL>(k4 ? it_ : it_ + 16 * gpr);                                                                          \
                    S##r2[u] = ld16w<NTL>(k4 ? it_ + 16 * gpr : it_ + 32 * gpr);                                                               \
                    S##r3[u] = ld16w<NTL>(k4 ? bl_ + 32 * gpr : bl_ + 48 * gpr);                                                               \
                    S##dh[u] = *reinterpret_cast<const uint16_t *>(k4 ? bl_ + 32 * gpr : rw_ + 52 * gpr + 2 * (ga_ >> 2)    );                 \
                }                                                                                                                              \
            } else {                                                                                                                           \
            const unsigned char * bp = rw_ + (size_t) (ga_ >> 2) * (S##ty[u] == PKV_TYPE_Q4_K ? 144 : (S##ty[u] == PKV_TYPE_Q5_K ? 176 : 210));           \
            const int n_ = pa_ >> 1, is_ = pa_ & 1;                                                                                            \
            if (MODE == 1) {                                                                                                                   \
                const bool k4_ = S##ty[u] == PKV_TYPE_Q4_K, k5_ = S##ty[u] == PKV_TYPE_Q5_K, k45_ = k4_ || k5_;                                 \
                S##r0[u] = ld16w<false>(bp + (k45_ ? 0 : 64 * n_ + 16 * is_));                                                                 \
                S##r1[u] = ld16w<false>(bp + (k4_ ? 16 + 32 * pa_ : (k5_ ? 48 + 32 * pa_ : 64 * n_ + 16 * is_ + 32)));                         \
                S##r2[u] = ld16w<false>(bp + (k4_ ? 32 + 32 * pa_ : (k5_ ? 64 + 32 * pa_ : 128 + 32 * n_ + 16 * is_)));                        \
                S##r3[u] = ld16w<false>(bp + (k5_ ? 16 : (k4_ ? 0 : 192)));                                                                    \
                S##r4[u] = ld16w<false>(bp + (k5_ ? 32 : (k4_ ? 0 : 192)));                                                                    \
                S##dh[u] = *reinterpret_cast<const uint16_t *>(bp + (k45_ ? 0 : 208));                                                         \
            } else                                                                                                                             \
            if (MODE == PKV_TYPE_Q5_K) {   /* header, the group's 32 bytes of nibbles, the block's 32 bytes of fifth bits (176 = 11 x 16: all aligned) */ \
                S##r0[u] = ld16w<false>(bp);                                                                                                   \
                S##r1[u] = ld16w<false>(bp + 48 + 32 * pa_    );                                                                               \
                S##r2[u] = ld16w<false>(bp + 64 + 32 * pa_    );                                                                               \
                S##r3[u] = ld16w<false>(bp + 16);                                                                                              \
                S##r4[u] = ld16w<false>(bp + 32); S##dh[u] = 0;                                                                                \
            } else                                                                                                                             \
            if (MODE == PKV_TYPE_Q4_K) {                                                                                                       \
                S##r0[u] = ld16w<false>(bp);                                                                                                   \
                S##r1[u] = ld16w<false>(bp + 16 + 32 * pa_    );                                                                               \
                S##r2[u] = ld16w<false>(bp + 32 + 32 * pa_    );                                                                               \
                S##r3[u] = make_uint4(0, 0, 0, 0); S##dh[u] = 0;                                                                               \
            } else {                                                                                                                           \
                const bool k4 = MODE == 0 && S##ty[u] == PKV_TYPE_Q4_K;                                                                        \
                S##r0[u] = ld16w<false>(bp + (k4 ? 0 : 64 * n_ + 16 * is_));                                                                   \
                S##r1[u] = ld16w<false>(bp + (k4 ? 16 + 32 * pa_     : 64 * n_ + 16 * is_ + 32));                                              \
                S##r2[u] = ld16w<false>(bp + (k4 ? 32 + 32 * pa_     : 128 + 32 * n_ + 16 * is_));                                             \
                S##r3[u] = ld16w<false>(bp + (k4 ? 0 : 192));                                                                                  \
                S##dh[u] = *reinterpret_cast<const uint16_t *>(bp + (k4 ? 0 : 208));                                                           \
            }                                                                                                                                  \
            }                                                                                                                                  \
        }                                                                                                                                      \
    }
#define PKV_UNIT_COMPUTE(S, RPI, FI)                                                                                                           \
    {                                                                                                                                          \
        _Pragma("unroll") for (int u = 0; u < UI; ++u) {                                                                                        \
            float part;                                                                                                                        \
            if (PKV_GEMV_ABLATE == 1) { part = __uint_as_float((S##r0[u].x ^ S##r1[u].y ^ S##r2[u].z ^ S##r3[u].w ^ S##r1[u].w ^ S##r2[u].x) & 0x3fffffffu); } else \
            if (MODE == PKV_TYPE_Q5_K || (MODE == 1 && S##ty[u] == PKV_TYPE_Q5_K)) part = dot_q5k<TAIL>(S##r0[u], S##r1[u], S##r2[u], S##r3[u], S##r4[u], S##p[u], sq + S##g[u] * REC); \
            else if (S##ty[u] == PKV_TYPE_Q4_K) part = dot_q4k<TAIL>(S##r0[u], S##r1[u], S##r2[u], S##p[u], sq + S##g[u] * REC);                        \
            else part = dot_q6k(S##r0[u], S##r1[u], S##r2[u], S##r3[u], S##dh[u], S##p[u], sq + ((S##g[u] & ~3) + 2 * (S##p[u] >> 1)) * REC);            \
            if (!S##ok[u]) part = 0.0f;                                                                                                        \
            acc0 += S##rs[u] ? 0.0f : part;                                                                                                    \
            acc1 += S##rs[u] ? part : 0.0f;                                                                                                    \
        }                                                                                                                                      \
        if (__builtin_amdgcn_readfirstlane(FI) == n_f0 - 1) { /* the row pair is complete (its matrix and first row: noted when the unit was requested) */ \
            const int mi_ = S##mi; const int64_t m0_ = S##m0;                                                                                  \
            float s0 = wsum_dpp(acc0), s1 = wsum_dpp(acc1);                                                                                    \
            if constexpr (KS == 2) { /* the two waves of a row pair: the second hands its sums over, the first adds (a + b: one order) */      \
                if (kpart == 1 && lane == 0) sPart[wv >> 1] = make_float2(s0, s1);                                                             \
                __syncthreads();                                                                                                               \
                const float2 o_ = sPart[wv >> 1];                                                                                              \
                s0 += o_.x; s1 += o_.y;                                                                                                        \
            }                                                                                                                                  \
            if (lane == 0 && live && kpart == 0) {                                                                                             \
                if (a_silu) {                                                                                                                  \
                    a.y[0][m0_] = s0 / (1.0f + expf(-s0)) * s1;                                                                                \
                } else {                                                                                                                       \
                    float * y = mi_ == 0 ? a.y[0] : (mi_ == 1 ? a.y[1] : a.y[2]);                                                              \
                    if (a.residual) *reinterpret_cast<float2 *>(y + m0_) = make_float2(a.residual[m0_] + s0, a.residual[m0_ + 1] + s1);        \
                    else            *reinterpret_cast<float2 *>(y + m0_) = make_float2(s0, s1);                                                \
                }                                                                                                                              \
            }                                                                                                                                  \
            acc0 = acc1 = 0.0f;                                                                                                                \
        }                                                                                                                                      \
    }

// ONE workgroup of NWV waves per CU (grid <= 256): the activation column is normalised / quantized / copied into LDS once per CU (r02's first
// version paid that prologue in each of up to 1024 four-wave workgroups: for ffn_down 57 KB of f32 column per 64 KB of weights).  Wave wv of
// workgroup b owns row pairs b + grid * (wv + NWV k), k = 0, 1, ...: at any moment the chip streams one contiguous band of grid * NWV row pairs and
// every CU has the same number of busy waves.  SHORT (at most two units per wave: wq / wk / wv, wo at Llama-3-8B sizes): no loop, everything
// the wave will ever read is requested before the column is touched.
// KS = 2 (ffn_down at Llama-3-8B sizes: 2048 row pairs of 3.5 units): the two halves of a row pair's groups go to two waves of the workgroup, so that
// all 16 waves of every CU stream (and quantize the 14336-element column) and every wave is SHORT; their sums meet in LDS behind one barrier.
// The five leading arguments (8 dwords) are preloaded into SGPRs at dispatch (-amdgpu-kernarg-preload-count=8): the column's loads go out without
// waiting for a kernel-argument fetch; r02 read everything through the by-value struct — three dependent scalar-load round trips (~1 us, stamps in
// profiles/r03_gemv_stamps.txt) before the first vector load.  The rest of GemvArgs arrives in ONE batch of scalar loads behind them.
template <int MODE, int NWV, bool SHORT, int KS = 1, bool PK = false>
__global__ void __launch_bounds__(64 * NWV) gemv_kernel(const float * __restrict__ ax, const float * __restrict__ anw, const char * __restrict__ aimg, const int aK, const float aeps,
                                                        const GemvArgs a) {
    constexpr bool NTL = PK && PKV_GEMV_NT;
    // A unit = ONE (row, group) item per lane = 3-4 wave loads of 1 KiB.  r02 ran two items per lane and two units ahead: 12 KB requested per wave,
    // 192 KB per CU — eight times what a CU keeps in flight (~24 KiB), so every wave sat blocked at load ISSUE for microseconds before it
    // reached the column prologue (timing ablations profiles/r03_gemv_ablation.txt: neither the dots nor the prologue explained 18 us against an 11 us
    // bare read).  Now one small unit goes out ahead of the prologue, the second behind it, and two stay in flight per wave in the loop.
    constexpr int UI = 1;
    constexpr int NT = 64 * NWV;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    // The by-value struct spans four 64-byte lines of the kernel-argument segment, and left to itself hipcc fetched its fields where they are used:
    // rp_end[] (line 1), then — behind the division of the row-pair count — silu_mul (line 2), then w[] (line 0) and type[]: three DEPENDENT scalar-
    // cache misses between the launch and the first weight request (ISA audit of the r03 build).  One field of each line is requested up here in one
    // batch and pinned behind the column's vector loads; the later scalar loads of their neighbours hit the scalar cache.  (Pinning all thirteen
    // dwords instead overflowed the SGPR file: spills to scratch at kernel entry.)
    // Measured (profiles/r03_gemv_args_early_ab.txt): the products' times do not move (7.13 vs 7.10, 5.30 vs 5.35, 14.61 vs 14.60, 14.21 vs 14.47 us); end to end 606-607
    // against 601 tok/s in one box — the scalar loads were not what the ramp waits for.  Kept: it costs nothing.
#ifndef PKV_GEMV_ARGS_EARLY
#define PKV_GEMV_ARGS_EARLY 1
#endif
#if PKV_GEMV_ARGS_EARLY
    const int e0 = a.rp_end[0], e1 = a.rp_end[1], e2 = a.rp_end[2], a_silu = a.silu_mul;
    const char * const a_w0 = a.w[0];
    const int grid = (int) gridDim.x;      // (an implicit argument behind the struct: the fourth line)
#define a_w1 a.w[1]
#define a_w2 a.w[2]
#define a_t0 a.type[0]
#define a_t1 a.type[1]
#define a_t2 a.type[2]
    __builtin_amdgcn_sched_barrier(0);     // (the requests stay up here; the wait for them comes behind the column's vector loads, at the pin below)
#else
#define e2 a.rp_end[2]
#define a_silu a.silu_mul
#define a_w0 a.w[0]
#define a_w1 a.w[1]
#define a_w2 a.w[2]
#define a_t0 a.type[0]
#define a_t1 a.type[1]
#define a_t2 a.type[2]
    const int e0 = a.rp_end[0], e1 = a.rp_end[1];
    const int grid = (int) gridDim.x;
#endif
    // The column's own loads go out FIRST.  Vector-memory results return in issue order (one vmcnt counter per wave) and a CU serves its
    // requests first come, first served: r02 requested two units of weights per wave (192 KB per CU at 16 waves) ahead of the column, so the
    // column (16-57 KB, L2-resident) arrived behind ALL of them — the norm / quantize prologue started 5-8 us into the launch and every dot
    // waited for it (ffn_down: the whole matrix had landed before the first dot ran; profiles/r03_gemv_read_pattern.txt: a bare read of the same
    // bytes takes 6.5 us where the product took 12.0).  Now: column loads, then the weight loads, then a counted wait for the column alone.
#if PKV_GEMV_STAMPS
    unsigned long long st_[20];
    for (int i = 0; i < 20; ++i) st_[i] = 0;
    st_[0] = __builtin_amdgcn_s_memrealtime();
#endif
    PKV_STAMP(1)
    const int n4 = (int) (aK >> 2);
    const bool colk = PKV_GEMV_ABLATE != 2 && aimg == nullptr;      // the column is normalised / quantized here (no image from a launch of its own)
    constexpr int NV = 2048 / NT;                                    // float4 per thread of the norm path (K <= 8192, checked on the host)
    constexpr int NVP = NT >= 512 ? 4096 / NT : 8;                   // ... preloaded without a norm: K <= 16384 (8192 at 4 waves) in one round trip, the rest in batches
    float4 v[NVP], w4[NV];
    if (colk) {
#pragma unroll
        for (int j = 0; j < NVP; ++j) {                              // float4 index i = tid + NT j: the 64 lanes of a wave hold one 256-block (n4 % 64 == 0)
            const int i = threadIdx.x + NT * j;
            v[j] = i - lane < n4 ? reinterpret_cast<const float4 *>(ax)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (anw) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = threadIdx.x + NT * j;
                w4[j] = i - lane < n4 ? reinterpret_cast<const float4 *>(anw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // r03: an image made by the producer of the column (the attention merge launch: pkv_attn_args::out_image) goes out ahead of the weights as well
    constexpr int NIQ = 4;                                           // image dwords per thread held in registers (K <= 16 NT), else the loop further down
    const bool imgk = PKV_GEMV_ABLATE != 2 && aimg != nullptr && n4 <= NIQ * NT && (aK >> 5) <= NT;
    uint32_t iq[NIQ];
    float dpre = 0.0f;
    int   bpre = 0;
    if (imgk) {
        const uint32_t * qi = reinterpret_cast<const uint32_t *>(aimg);
#pragma unroll
        for (int j = 0; j < NIQ; ++j) { const int i = threadIdx.x + NT * j; iq[j] = i < n4 ? qi[i] : 0u; }
        if ((int) threadIdx.x < (aK >> 8)) dpre = reinterpret_cast<const float *>(aimg + aK)[threadIdx.x];
        if ((int) threadIdx.x < (aK >> 5)) bpre = reinterpret_cast<const int *>(aimg + aK + (aK >> 8) * 4)[threadIdx.x];
    }
    __builtin_amdgcn_sched_barrier(0);
#if PKV_GEMV_ARGS_EARLY
    asm volatile("" :: "s"(e0), "s"(e1), "s"(e2), "s"(a_silu), "s"(a_w0), "s"(grid));
#endif
    constexpr int TAIL = (MODE == PKV_TYPE_Q4_K || MODE == PKV_TYPE_Q5_K) ? 0 : 1;                // record tail format (per-32 sums | per-16 sums as int16)
    const int gpr = aK >> 6;
    uint32_t * sq = g_lds;                                           // gpr records of REC dwords
    __shared__ double sred[32];                                       // sums of squares per 256-element block (norm prologue)
    const int total_rp = e0 > e1 ? (e0 > e2 ? e0 : e2) : (e1 > e2 ? e1 : e2);
    __shared__ float2 sPart[NWV / 2];
    const int stride = grid * (NWV / KS);
    const int kpart = KS == 2 ? (wv & 1) : 0;
    const int mine = (int) blockIdx.x + grid * (KS == 2 ? (wv >> 1) : wv);
    const bool live = mine < total_rp;                                // a surplus wave shadows the last row pair up to the stores: no early exit before the barriers
    const int first = min(mine, total_rp - 1);
    const int n_rp = (total_rp - first + stride - 1) / stride;       // row pairs of this wave
    // (row, group) items of a row pair this wave owns: all 2 gpr of them, or one half (rounded to whole 64-lane strides)
    const int f_half = KS == 2 ? ((gpr + 63) / 64) * 64 : 2 * gpr;
    const int f_lo = kpart * f_half, f_hi = (KS == 2 && kpart == 0) ? f_half : 2 * gpr;
    const int n_f0 = (f_half + 64 * UI - 1) / (64 * UI);              // units per row pair (per half)
    const int n_it = n_rp * n_f0;
    // kernel-argument arrays are only ever indexed with constants (a dynamic index would move the struct to scratch)
    auto rows_of = [&](int rp, const char *& r0, const char *& r1, int & type0, int & type1, int & mi, int64_t & m0) {
        rp = __builtin_amdgcn_readfirstlane(rp);                      // wave-uniform by construction: the pointer arithmetic stays on the scalar unit
        mi = rp < e0 ? 0 : (rp < e1 ? 1 : 2);
        const int rpl = rp - (mi == 0 ? 0 : (mi == 1 ? e0 : e1));
        const char * w0, * w1;
        if (a_silu) { m0 = rp; w0 = a_w0; w1 = a_w1; type0 = a_t0; type1 = a_t1; mi = 0; }
        else {
            m0 = 2 * (int64_t) rpl;
            w0 = w1 = mi == 0 ? a_w0 : (mi == 1 ? a_w1 : a_w2);
            type0 = type1 = mi == 0 ? a_t0 : (mi == 1 ? a_t1 : a_t2);
        }
        r0 = w0 + (size_t) m0 * row_size(type0, aK);
        r1 = w1 + (size_t) (a_silu ? m0 : m0 + 1) * row_size(type1, aK);
    };
    float acc0 = 0.0f, acc1 = 0.0f;
    PKV_UNIT_DECL(A) PKV_UNIT_DECL(B)
    // unit cursors (row pair of this wave, unit inside the pair) without divisions: l* = the next unit to request, c* = the next to consume; past
    // the end a request re-reads the wave's last unit (the load count stays static: a load behind a branch drains the queue at the join)
    int lr = 0, lf = 0, cr = 0, cf = 0;
    auto step = [&](int & r, int & f) { if (++f == n_f0) { f = 0; ++r; } };
#define PKV_LOAD_NEXT(S) { const bool past_ = lr >= n_rp; if (PKV_GEMV_ABLATE == 3 && past_) {} else PKV_UNIT_LOAD(S, past_ ? n_rp - 1 : lr, past_ ? n_f0 - 1 : lf, past_) step(lr, lf); }
#define PKV_COMPUTE_NEXT(S) { PKV_UNIT_COMPUTE(S, cr, cf) step(cr, cf); }
    __builtin_amdgcn_sched_barrier(0);
    PKV_LOAD_NEXT(A)                                                  // the first weight loads fly while the column is prepared
#if PKV_GEMV_ABLATE == 4
    PKV_LOAD_NEXT(B)
#endif
    __builtin_amdgcn_sched_barrier(0);
    PKV_STAMP(2)                                                      // column + unit A requested
#if PKV_GEMV_STAMPS == 1
    __builtin_amdgcn_s_waitcnt(0x0F70 | 3);                           // vmcnt(3): (q4_K) the column has landed — clean phase shares
    PKV_STAMP(3)
#endif

    // tail of a group's record from the quantizer's registers: lane = float4 i of the column (dword i & 15 of group i >> 4)
    auto put_tail = [&](int i, float d, int bs32, int bs16) {
        uint32_t * r = sq + (i >> 4) * REC + 16;
        if constexpr (TAIL == 0) { if ((i & 7) == 0) r[(i >> 3) & 1] = (uint32_t) bs32; }
        else                     { if ((i & 3) == 0) reinterpret_cast<uint16_t *>(r)[(i >> 2) & 3] = (uint16_t) bs16; }
        if ((i & 15) == 0) r[2] = __float_as_uint(d);                 // the 256-block's scale, once per 64-group
    };
    if (PKV_GEMV_ABLATE == 2) {} else
    if (imgk) {
        // the image sits in registers (requested ahead of the weights): K / 4 dwords of int8, the 256-block scales, the per-32 sums
#pragma unroll
        for (int j = 0; j < NIQ; ++j) {
            const int i = threadIdx.x + NT * j;
            if (i - lane < n4) {                                      // wave-uniform (n4 % 64 == 0)
                const uint32_t w = iq[j];
                sq[(i >> 4) * REC + (i & 15)] = w;
                if constexpr (TAIL == 1) {
                    int x = __builtin_amdgcn_sdot4((int) w, 0x01010101, 0, false);
                    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
                    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
                    if ((i & 3) == 0) reinterpret_cast<uint16_t *>(sq + (i >> 4) * REC + 16)[(i >> 2) & 3] = (uint16_t) x;
                }
            }
        }
        if ((int) threadIdx.x < (aK >> 8)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[(4 * threadIdx.x + k) * REC + 18] = __float_as_uint(dpre);
        }
        if constexpr (TAIL == 0) { if ((int) threadIdx.x < (aK >> 5)) sq[(threadIdx.x >> 1) * REC + 16 + (threadIdx.x & 1)] = (uint32_t) bpre; }
    } else if (aimg) {
        // the column image was made by a launch of its own: K / 4 dwords of int8, the 256-block scales, the per-32 sums
        const uint32_t * qi = reinterpret_cast<const uint32_t *>(aimg);
        const float * dd = reinterpret_cast<const float *>(aimg + aK);
        const int *   bb = reinterpret_cast<const int *>(aimg + aK + (aK >> 8) * 4);
        for (int i = threadIdx.x; i < (int) (aK >> 2); i += NT) {
            const uint32_t w = qi[i];
            sq[(i >> 4) * REC + (i & 15)] = w;
            if constexpr (TAIL == 1) {                                // per-16 sums are not in the image: from the quants (4 lanes = one run of 16)
                int x = __builtin_amdgcn_sdot4((int) w, 0x01010101, 0, false);
                x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
                if ((i & 3) == 0) reinterpret_cast<uint16_t *>(sq + (i >> 4) * REC + 16)[(i >> 2) & 3] = (uint16_t) x;
            }
        }
        for (int g = threadIdx.x; g < gpr; g += NT) sq[g * REC + 18] = __float_as_uint(dd[g >> 2]);
        if constexpr (TAIL == 0)
            for (int b = threadIdx.x; b < (int) (aK >> 5); b += NT) sq[(b >> 1) * REC + 16 + (b & 1)] = (uint32_t) bb[b];
    } else {
        // ... or here: [rms_norm(x) * w ->] q8_K straight into LDS.
        // The sum of squares is the canonical one of pkv_common.h (f32 squares accumulated in f64 per 256-block, blocks added in order), the same
        // bits whatever the workgroup shape: the fused launch and the norm + quantize + product launches produce the same image.
        auto put = [&](int i, const float4 o) {
            uint32_t packed; float d; int bs, bs16;
            quant_block_q8k(o, lane, packed, d, bs, bs16);
            sq[(i >> 4) * REC + (i & 15)] = packed;
            put_tail(i, d, bs, bs16);
        };
        if (anw) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int b = wv + NWV * j;                           // the wave's 256-block of pass j
                if (b * 64 < n4) {                                    // wave-uniform
                    const double bsum = block_sumsq(v[j]);
                    if (lane == 0) sred[b] = bsum;
                }
            }
            PKV_STAMP(16)
            __syncthreads();
            PKV_STAMP(17)
            const float scale = rms_scale(sred, (n4 + 63) >> 6, aK, aeps);
#if PKV_GEMV_STAMPS
            st_[19] = (unsigned long long) __float_as_uint(scale);   // (forces the value)
#endif
            PKV_STAMP(18)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = threadIdx.x + NT * j;                   // n4 % 64 == 0: a wave is in or out as a whole
                if (i - lane < n4) {
                    const float4 nn = make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale);          // rms_norm(x): as rms_norm_mul_kernel rounds it
                    const float4 oo = make_float4(nn.x * w4[j].x, nn.y * w4[j].y, nn.z * w4[j].z, nn.w * w4[j].w);
                    if (blockIdx.x == 0 && i < n4) {
                        if (a.norm_out) reinterpret_cast<float4 *>(a.norm_out)[i] = nn;
                        if (a.col_out)  reinterpret_cast<float4 *>(a.col_out)[i] = oo;
                    }
                    put(i, oo);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NVP; ++j) { const int i = threadIdx.x + NT * j; if (i - lane < n4) put(i, v[j]); }
            constexpr int NB8 = 8;                                    // a longer column: batches of 8 float4 per thread, the loads of a batch share one round trip
            for (int i0 = threadIdx.x + NT * NVP; i0 - lane < n4; i0 += NT * NB8) {
                float4 vb[NB8];
#pragma unroll
                for (int j = 0; j < NB8; ++j) { const int i = i0 + NT * j; vb[j] = reinterpret_cast<const float4 *>(ax)[i - lane < n4 ? i : i0]; }
#pragma unroll
                for (int j = 0; j < NB8; ++j) { const int i = i0 + NT * j; if (i - lane < n4) put(i, vb[j]); }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#if PKV_GEMV_ABLATE != 4
    PKV_LOAD_NEXT(B)                                                  // the wave's second unit: requested as soon as its share of the column is in LDS
#endif
    __builtin_amdgcn_sched_barrier(0);
    PKV_STAMP(4)                                                      // this wave's share of the column is in LDS, unit B requested
    __syncthreads();
    PKV_STAMP(5)

    if constexpr (SHORT) {
        PKV_COMPUTE_NEXT(A)
        PKV_STAMP(6)
        if (n_it > 1) PKV_COMPUTE_NEXT(B)
        PKV_STAMP(7)
    } else {
        for (int it = 0; it < n_it; it += 2) {           // two units in flight behind the one being consumed
            PKV_COMPUTE_NEXT(A)
#if PKV_GEMV_STAMPS
            if (it == 0) PKV_STAMP(6) else if (it == 2) PKV_STAMP(8) else if (it + 2 >= n_it) PKV_STAMP(10)
#endif
            PKV_LOAD_NEXT(A)
            if (it + 1 < n_it) PKV_COMPUTE_NEXT(B)
#if PKV_GEMV_STAMPS
            if (it == 0) PKV_STAMP(7) else if (it == 2) PKV_STAMP(9) else if (it + 2 >= n_it) PKV_STAMP(11)
#endif
            PKV_LOAD_NEXT(B)
        }
    }
#if PKV_GEMV_STAMPS
    PKV_STAMP(12)
    __builtin_amdgcn_s_waitcnt(0);                                    // stores acknowledged, stray requests back
    PKV_STAMP(13)
    st_[14] = __builtin_amdgcn_s_memrealtime();
    st_[15] = (unsigned long long) n_it;
    if (a.stamps && lane == 0) for (int i = 0; i < 20; ++i) a.stamps[((size_t) blockIdx.x * NWV + wv) * 20 + i] = st_[i];
#endif
}

// ---- ggml_get_rows (/ref/src/llama-model.cpp:4047 inp_embd = get_rows(tok_embd, tokens); CPU kernel ggml_compute_forward_get_rows,
//      /ref/ggml/src/ggml-cpu/ggml-cpu.c:8940-9090): out[i][:] = dequantize(row ids[i]) ----------------------------------------------
// PK: the row is in the plane layout of pkv_wlayout.h; byte `o` of the GGUF-order row is at row[pk(o)]
template <int T, bool PK = false>
__global__ void __launch_bounds__(256) get_rows_kernel(const char * w, const int32_t * ids, float * out, int64_t K, int64_t n_rows_w) {
    const int gpr = (int) (K >> 6);
    auto pk = [&](size_t o) -> size_t { if constexpr (!PK) return o; else return T == PKV_TYPE_Q4_K ? pk_offset_q4k(gpr, o) : (T == PKV_TYPE_Q5_K ? pk_offset_q5k(gpr, o) : pk_offset_q6k(gpr, o)); };
    // blockIdx.y = slice of the row: one looked-up row of a decode step is spread over K / 256 workgroups (as ONE workgroup walking 4096 elements with
    // dependent byte loads it took 12.5 us per token, profiles/r03_e2e_kernel_stats.csv)
    const int64_t k_first = (int64_t) blockIdx.y * 256 + threadIdx.x, k_step = 256 * (int64_t) gridDim.y;
    const int64_t r = ids[blockIdx.x];
    float * o = out + (size_t) blockIdx.x * K;
    if (r < 0 || r >= n_rows_w) { for (int64_t k = k_first; k < K; k += k_step) o[k] = 0.0f; return; }
    const unsigned char * row = reinterpret_cast<const unsigned char *>(w) + (size_t) r * row_size(T, K);
    if constexpr (T == PKV_TYPE_F32) {
        for (int64_t k = k_first; k < K; k += k_step) o[k] = reinterpret_cast<const float *>(row)[k];
    } else if constexpr (T == PKV_TYPE_F16) {
        for (int64_t k = k_first; k < K; k += k_step) o[k] = (float) reinterpret_cast<const _Float16 *>(row)[k];
    } else if constexpr (T == PKV_TYPE_Q8_0) {
        for (int64_t k = k_first; k < K; k += k_step) {
            const unsigned char * b = row + (k >> 5) * 34;
            o[k] = h2f(*reinterpret_cast<const uint16_t *>(b)) * (float) (int) (signed char) b[2 + (k & 31)];
        }
    } else if constexpr (T == PKV_TYPE_Q4_K) {      // dequantize_row_q4_K (/ref/ggml/src/ggml-quants.c:1280-1302)
        for (int64_t k = k_first; k < K; k += k_step) {
            const size_t b = (size_t) (k >> 8) * 144;                       // GGUF-order offset of the block inside the row
            const int e = (int) (k & 255), j = e >> 5, l = e & 31;           // sub-block j, element l
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + pk(b))), dmin = h2f(*reinterpret_cast<const uint16_t *>(row + pk(b + 2)));
            auto s = [&](int i) -> int { return (int) row[pk(b + 4 + i)]; };
            int sc, mn;
            if (j < 4) { sc = s(j) & 63; mn = s(j + 4) & 63; }
            else       { sc = (s(j + 4) & 0xF) | ((s(j - 4) >> 6) << 4); mn = (s(j + 4) >> 4) | ((s(j) >> 6) << 4); }
            const unsigned char qb = row[pk(b + 16 + 32 * (j >> 1) + l)];
            const int q = (j & 1) ? (qb >> 4) : (qb & 0xF);
            const float d1 = d * (float) sc, m1 = dmin * (float) mn;         // f32 products, as the reference forms them
            o[k] = d1 * (float) q - m1;
        }
    } else if constexpr (T == PKV_TYPE_Q5_K) {      // dequantize_row_q5_K (:1482-1507)
        for (int64_t k = k_first; k < K; k += k_step) {
            const size_t b = (size_t) (k >> 8) * 176;                       // GGUF-order offset of the block inside the row
            const int e = (int) (k & 255), j = e >> 5, l = e & 31;
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + pk(b))), dmin = h2f(*reinterpret_cast<const uint16_t *>(row + pk(b + 2)));
            auto s = [&](int i) -> int { return (int) row[pk(b + 4 + i)]; };
            int sc, mn;
            if (j < 4) { sc = s(j) & 63; mn = s(j + 4) & 63; }
            else       { sc = (s(j + 4) & 0xF) | ((s(j - 4) >> 6) << 4); mn = (s(j + 4) >> 4) | ((s(j) >> 6) << 4); }
            const unsigned char qb = row[pk(b + 48 + 32 * (j >> 1) + l)];
            const int q = ((j & 1) ? (qb >> 4) : (qb & 0xF)) + (((row[pk(b + 16 + l)] >> j) & 1) ? 16 : 0);
            const float d1 = d * (float) sc, m1 = dmin * (float) mn;
            o[k] = d1 * (float) q - m1;
        }
    } else {                                         // q6_K: dequantize_row_q6_K (:1690-1719)
        for (int64_t k = k_first; k < K; k += k_step) {
            const size_t b = (size_t) (k >> 8) * 210;
            const int e = (int) (k & 255), n = e >> 7, r128 = e & 127, j = r128 >> 5, l = r128 & 31;
            const unsigned char lq = row[pk(b + 64 * n + l + 32 * (j & 1))], hq = row[pk(b + 128 + 32 * n + l)];
            const int q = (int) (((j >> 1) ? (lq >> 4) : (lq & 0xF)) | (((hq >> (2 * j)) & 3) << 4)) - 32;
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + pk(b + 208)));
            const int sc = (int) (signed char) row[pk(b + 192 + 8 * n + (l >> 4) + 2 * j)];
            o[k] = d * (float) sc * (float) q;
        }
    }
}

// GGUF-order rows <-> plane-layout rows (pkv_wlayout.h), 2 bytes per thread-step (a block_q6_K is only 2-byte aligned; every field of both
// formats is a whole number of 2-byte units).  One-time work at upload, or the inverse for tensor_get.
template <int T>
__global__ void __launch_bounds__(256) pack_rows_kernel(const uint16_t * src, uint16_t * dst, int64_t n_rows, int64_t K, int inverse) {
    const int gpr = (int) (K >> 6);
    const int64_t rb2 = (int64_t) (row_size(T, K) >> 1), n = n_rows * rb2;
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) {
        const int64_t r = i / rb2, o = (i - r * rb2) * 2;
        const int64_t po = (int64_t) (T == PKV_TYPE_Q4_K ? pk_offset_q4k(gpr, (size_t) o) : (T == PKV_TYPE_Q5_K ? pk_offset_q5k(gpr, (size_t) o) : pk_offset_q6k(gpr, (size_t) o)));
        if (inverse) dst[i] = src[r * rb2 + (po >> 1)];
        else         dst[r * rb2 + (po >> 1)] = src[i];
    }
}

}  // namespace

int launch_pack_rows(const void * src, void * dst, int type, int64_t n_rows, int64_t K, int inverse, hipStream_t st) {
    PKV_REQUIRE(src && dst && src != dst && n_rows >= 0 && K > 0 && K % 256 == 0, "pkv_weights_pack: bad arguments (src != dst, K %% 256 == 0)");
    PKV_REQUIRE(type == PKV_TYPE_Q4_K || type == PKV_TYPE_Q5_K || type == PKV_TYPE_Q6_K, "pkv_weights_pack: q4_K / q5_K / q6_K rows only");
    PKV_REQUIRE(((uintptr_t) src % 2) == 0 && ((uintptr_t) dst % 2) == 0, "pkv_weights_pack: 2-byte aligned pointers");
    if (n_rows == 0) return PKV_OK;
    const int64_t n = n_rows * (int64_t) (row_size(type, K) >> 1);
    const unsigned grid = (unsigned) std::min<int64_t>((n + 255) / 256, 1 << 16);
    if (type == PKV_TYPE_Q4_K) hipLaunchKernelGGL(pack_rows_kernel<PKV_TYPE_Q4_K>, dim3(grid), dim3(256), 0, st, (const uint16_t *) src, (uint16_t *) dst, n_rows, K, inverse);
    else if (type == PKV_TYPE_Q5_K) hipLaunchKernelGGL(pack_rows_kernel<PKV_TYPE_Q5_K>, dim3(grid), dim3(256), 0, st, (const uint16_t *) src, (uint16_t *) dst, n_rows, K, inverse);
    else                       hipLaunchKernelGGL(pack_rows_kernel<PKV_TYPE_Q6_K>, dim3(grid), dim3(256), 0, st, (const uint16_t *) src, (uint16_t *) dst, n_rows, K, inverse);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

size_t act_image_bytes(int64_t K) { return (size_t) K + (size_t) (K / 256) * 4 + (size_t) (K / 32) * 4; }

int launch_quantize_act(const float * x, void * img, int64_t K, hipStream_t st) {
    PKV_REQUIRE(K > 0 && K % 256 == 0, "pkv_quantize_act: K must be a positive multiple of 256");
    hipLaunchKernelGGL(quant_act_kernel, dim3((unsigned) ((K / 4 + 255) / 256)), dim3(256), 0, st, x, (char *) img, K);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_gemv(const pkv_gemv & g, hipStream_t st) {
    PKV_REQUIRE(g.n >= 1 && g.n <= PKV_GEMV_MAX && (g.xq || g.x) && g.K > 0 && g.K % 256 == 0 && g.K <= 49152, "pkv_gemv_q: bad arguments");
    PKV_REQUIRE(g.xq || (((uintptr_t) g.x % 16) == 0 && (!g.norm_w || (g.K <= 8192 && ((uintptr_t) g.norm_w % 16) == 0 && g.eps >= 0.0f))),
                "pkv_gemv_q: in-kernel column: x (and norm_w) 16-byte aligned, K <= 8192 with a norm");
    PKV_REQUIRE(!g.silu_mul || (g.n == 2 && g.M[0] == g.M[1] && !g.residual), "pkv_gemv_q: silu_mul needs two products of equal M and no residual");
    PKV_REQUIRE(!g.residual || g.n == 1, "pkv_gemv_q: the residual epilogue takes one product");
    GemvArgs a = {};
    int tot = 0;
    const bool packed = (g.type_w[0] & PKV_TYPE_PACKED) != 0;       // plane-layout rows (pkv_wlayout.h): all matrices of a launch, or none
    int tw[PKV_GEMV_MAX] = {};
    for (int i = 0; i < g.n; ++i) {
        PKV_REQUIRE(g.w[i] && g.y[i] && g.M[i] > 0 && g.M[i] % 2 == 0, "pkv_gemv_q: matrix %d: null pointer or odd row count", i);
        PKV_REQUIRE(((g.type_w[i] & PKV_TYPE_PACKED) != 0) == packed, "pkv_gemv_q: the matrices of one launch are all packed or all in GGUF order");
        tw[i] = g.type_w[i] & ~PKV_TYPE_PACKED;
        if (tw[i] != PKV_TYPE_Q4_K && tw[i] != PKV_TYPE_Q5_K && tw[i] != PKV_TYPE_Q6_K) { set_error("pkv_gemv_q: weight type %d not supported (q4_K, q5_K, q6_K)", g.type_w[i]); return PKV_ERR_UNSUPPORTED; }
        a.w[i] = (const char *) g.w[i]; a.y[i] = g.y[i]; a.M[i] = g.M[i]; a.type[i] = tw[i];
        tot += g.silu_mul ? (i == 0 ? (int) g.M[0] : 0) : (int) (g.M[i] / 2);
        a.rp_end[i] = tot;
    }
    for (int i = g.n; i < PKV_GEMV_MAX; ++i) { a.rp_end[i] = tot; a.w[i] = a.w[0]; a.y[i] = a.y[0]; a.M[i] = 0; a.type[i] = a.type[0]; }
    a.n = g.n; a.silu_mul = g.silu_mul; a.residual = g.residual; a.img = (const char *) g.xq; a.K = g.K;
    a.x = g.x; a.norm_w = g.xq ? nullptr : g.norm_w; a.eps = g.eps;
    a.norm_out = a.norm_w ? g.norm_out : nullptr; a.col_out = a.norm_w ? g.col_out : nullptr;
#if PKV_GEMV_STAMPS
    if (const char * e = getenv("PKV_GEMV_STAMP_PTR")) a.stamps = (unsigned long long *) strtoull(e, nullptr, 0);   // device buffer [grid][waves][16] of the caller
#endif
    a.gpr = (int) (g.K / 64); a.gs = 0;
    const size_t lds = (size_t) REC * a.gpr * 4;
    // one workgroup per CU; its wave count follows the row pairs there are (every wave one row pair at least where the product has them)
    bool all4 = true, all5 = true, all6 = true;
    for (int i = 0; i < g.n; ++i) { all4 = all4 && tw[i] == PKV_TYPE_Q4_K; all5 = all5 && tw[i] == PKV_TYPE_Q5_K; all6 = all6 && tw[i] == PKV_TYPE_Q6_K; }
    bool any5 = false;
    for (int i = 0; i < g.n; ++i) any5 = any5 || tw[i] == PKV_TYPE_Q5_K;
    // (a group that mixes q5_K rows with others — wq / wk q5_K + wv q6_K of a Q5_K_M file — runs as MODE 1: as two launches it cost 16.8 us against 8.2)
    const int mode = all4 ? PKV_TYPE_Q4_K : (all5 ? PKV_TYPE_Q5_K : (all6 ? PKV_TYPE_Q6_K : (any5 ? 1 : 0)));
    int nwv  = tot >= 256 * 12 ? 16 : (tot >= 256 * 6 ? 8 : 4);
    // (8 waves with exactly 7 row pairs each instead of 16 with 3 or 4 — gate + up at Llama-3-8B sizes — is SLOWER, 16.4 vs 14.3 us: the waves hide
    // each other's load latency; profiles/r03_gemv_shapes.txt)
    if (tot >= 256 * 128) nwv = 8;   // ... a very long stream (lm_head: 250 row pairs per CU) runs 5 % faster with 8: 73.3 vs 76.9 us (sweep of PKV_GEMV_NWV, same file)
    if (const char * e = getenv("PKV_GEMV_NWV")) { const int v = atoi(e); if (v == 4 || v == 8 || v == 16) nwv = v; }   // experiments
    int grid = tot >= 256 * nwv ? 256 : (tot + nwv - 1) / nwv;
    // long rows, few of them (ffn_down): two waves per row pair, 16 waves per CU
    const int f_half = ((a.gpr + 63) / 64) * 64;
    const bool split = nwv == 8 && grid * 8 >= tot && 2 * a.gpr > 256;
    if (split) nwv = 16;
    const int ui = 1;                                                // items per lane and unit (the kernel's UI)
    const int n_f0 = ((split ? f_half : 2 * a.gpr) + 64 * ui - 1) / (64 * ui);
    const bool shrt = (int64_t) (split ? 1 : (tot + grid * nwv - 1) / (grid * nwv)) * n_f0 <= 2;
#define PKV_GEMV_LAUNCH(MODE_, NWV_, SHORT_, KS_, PK_) hipLaunchKernelGGL((gemv_kernel<MODE_, NWV_, SHORT_, KS_, PK_>), dim3((unsigned) grid), dim3(64 * NWV_), lds, st, \
                                                                          a.x, a.norm_w, a.img, (int) a.K, a.eps, a)
#define PKV_GEMV_PICK_S(MODE_, NWV_, KS_, PK_) do { if (shrt) PKV_GEMV_LAUNCH(MODE_, NWV_, true, KS_, PK_); else PKV_GEMV_LAUNCH(MODE_, NWV_, false, KS_, PK_); } while (0)
#define PKV_GEMV_PICK_W(MODE_, PK_) do { if (split) PKV_GEMV_PICK_S(MODE_, 16, 2, PK_); else if (nwv == 16) PKV_GEMV_PICK_S(MODE_, 16, 1, PK_); else if (nwv == 8) PKV_GEMV_PICK_S(MODE_, 8, 1, PK_); else PKV_GEMV_PICK_S(MODE_, 4, 1, PK_); } while (0)
#define PKV_GEMV_PICK_P(MODE_) do { if (packed) PKV_GEMV_PICK_W(MODE_, true); else PKV_GEMV_PICK_W(MODE_, false); } while (0)
    if (mode == PKV_TYPE_Q4_K)      PKV_GEMV_PICK_P(PKV_TYPE_Q4_K);
    else if (mode == PKV_TYPE_Q5_K) PKV_GEMV_PICK_P(PKV_TYPE_Q5_K);
    else if (mode == PKV_TYPE_Q6_K) PKV_GEMV_PICK_P(PKV_TYPE_Q6_K);
    else if (mode == 1)             PKV_GEMV_PICK_P(1);
    else                            PKV_GEMV_PICK_P(0);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

int launch_get_rows(const void * w, int type_w, const int32_t * ids, float * out, int64_t n_ids, int64_t K, int64_t n_rows_w, hipStream_t st) {
    if (n_ids == 0) return PKV_OK;
    const int64_t slices = std::max<int64_t>(1, std::min<int64_t>((K + 255) / 256, 2048 / n_ids));   // few rows: cut them; a prompt's rows fill the chip as they are
    const dim3 grid((unsigned) n_ids, (unsigned) slices);
    if (type_w == (PKV_TYPE_Q4_K | PKV_TYPE_PACKED)) { hipLaunchKernelGGL((get_rows_kernel<PKV_TYPE_Q4_K, true>), grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); PKV_HIP(hipGetLastError()); return PKV_OK; }
    if (type_w == (PKV_TYPE_Q5_K | PKV_TYPE_PACKED)) { hipLaunchKernelGGL((get_rows_kernel<PKV_TYPE_Q5_K, true>), grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); PKV_HIP(hipGetLastError()); return PKV_OK; }
    if (type_w == (PKV_TYPE_Q6_K | PKV_TYPE_PACKED)) { hipLaunchKernelGGL((get_rows_kernel<PKV_TYPE_Q6_K, true>), grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); PKV_HIP(hipGetLastError()); return PKV_OK; }
    switch (type_w) {
        case PKV_TYPE_F32:  hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_F32>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_F16:  hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_F16>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q8_0: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q8_0>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q4_K: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q4_K>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q5_K: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q5_K>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        case PKV_TYPE_Q6_K: hipLaunchKernelGGL(get_rows_kernel<PKV_TYPE_Q6_K>, grid, dim3(256), 0, st, (const char *) w, ids, out, K, n_rows_w); break;
        default: set_error("pkv_get_rows: type %d not supported", type_w); return PKV_ERR_UNSUPPORTED;
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

}  // namespace pkv
