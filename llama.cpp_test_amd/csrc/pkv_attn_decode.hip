// pkv_attn_decode.hip — single-token paged decode attention for gfx950 (MI355X), hand-written HIP.
//
// Replaces, for one query token per sequence, the reference's K/V gather views + GGML_OP_FLASH_ATTN_EXT
// (/ref/src/llama-graph.cpp:1404-1432, CPU kernel /ref/ggml/src/ggml-cpu/ggml-cpu.c:12219-12432).
//
// Structure (HBM-bound: 4 flop/B at gqa 4, f16):
//   grid  = (splits, n_head_kv, n_seq); one 256-thread workgroup per (sequence, kv head, 128-token KV block),
//           i.e. 8 x 32 = 256 workgroups = one per CU for Llama-3-8B at ctx 4096.  Each K/V tile is read ONCE and
//           serves all `G = n_head/n_head_kv` query heads (the reference's CUDA vec kernel re-reads it G times).
//   loads : every lane issues ALL its 16-byte K and V tile loads up front (64 KiB in flight per workgroup,
//           1 KiB fully-coalesced per wave instruction), then Q.  No LDS round trip for K/V: the tile is
//           consumed straight from VGPRs (cdna_hip_programming.md §5 "GEMV / M <= 16" row).
//   QK^T  : v_dot2c_f32_f16 (f16 K, Q rounded to f16 as the reference's q_to_vec_dot) or v_dot4c_i32_i8 (q8_0/q4_0 K,
//           Q quantized to q8_0 exactly as /ref/ggml/src/ggml-cpu/ggml-cpu.c:12343 does); the per-row partial sums
//           are finished with DPP row reductions (no LDS).
//   softmax: logits -> LDS, workgroup max / sum with wave DPP reductions; P -> LDS.
//   PV    : f32 accumulators (the reference accumulates f16-V in f16; we are strictly more accurate),
//           cross-lane-row and cross-wave sums, then one (m, l, O[G][D]) partial per workgroup.
//   A second tiny kernel merges the per-block partials (flash-decoding).
#include "pkv_common.h"
#include "pkv_rowcodec.h"

#include <hip/hip_ext.h>
#include <cstdlib>

namespace pkv {
namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int WG_TOK = 128;  // KV tokens per workgroup (== one block at block_size 128)

template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// all-reduce (sum) over aligned groups of LPR consecutive lanes, LPR in {1,2,4,8,16}: DPP only
template <int LPR> __device__ __forceinline__ float group_sum(float x) {
    if (LPR >= 2)  x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    if (LPR >= 4)  x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    if (LPR >= 8)  x += dpp_mov<0x141>(x);  // row_half_mirror
    if (LPR >= 16) x += dpp_mov<0x140>(x);  // row_mirror
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x));
    x = fmaxf(x, dpp_mov<0x4E>(x));
    x = fmaxf(x, dpp_mov<0x141>(x));
    x = fmaxf(x, dpp_mov<0x140>(x));
    x = fmaxf(x, __shfl_xor(x, 16));
    x = fmaxf(x, __shfl_xor(x, 32));
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
    x = group_sum<16>(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

// ---- per-type tile access -------------------------------------------------------------------------
// K side: a lane owns KE consecutive elements (16 bytes) of one token row; LPR lanes cover a row.
template <int T, int D> struct KTr;
template <int D> struct KTr<PKV_TYPE_F16, D>  { static constexpr int E = 8,  LPR = D / 8;  };
template <int D> struct KTr<PKV_TYPE_Q8_0, D> { static constexpr int E = 16, LPR = D / 16; };
template <int D> struct KTr<PKV_TYPE_Q4_0, D> { static constexpr int E = 32, LPR = D / 32; };

struct DecodeArgs {
    const float *   q;
    int64_t         q_stride_tok, q_stride_head;
    const char *    k_pool;
    const char *    v_pool;
    const int32_t * block_table;
    const int32_t * seq_lens;
    const int32_t * slot_pos;
    const int32_t * q_pos;
    const _Float16 * mask;    // optional additive KQ_mask row per sequence (f16 [n_seq][mask_stride], column = logical cell):
    int64_t         mask_stride;   // -inf hides the cell; rides on the SLOTPOS variant of the kernel
    float *         part_o;   // [n_seq][n_head][n_splits][D]
    float *         part_ml;  // [n_seq][n_head][n_splits][2]
    float *         out;      // [n_seq][n_head][D]; NULL: leave the partials (PKV_ATTN_NO_COMBINE)
    float *         out_direct; // the caller's out whatever the flags (the one-workgroup-per-stream form writes it itself)
    char *          out_image;  // optional: q8_K activation image of every out row (attn_combine_kernel<D, true>), image_bytes per token
    int             image_bytes;
    unsigned *      counters; // [n_seq][n_head_kv] arrival tickets (fused merge), zero between launches
    int             part_o_bytes, part_ml_bytes;   // extents of the partial slabs (buffer descriptors of the fused merge)
    int   n_head, n_head_kv, block_size, max_blocks, n_splits, bs_shift;  // block_size = WG_TOK << bs_shift
    int   kv_div, n_head_kv_pool;  // gqa ratios 16 / 32 run as ratio 8 over kv_div = 2 / 4 times as many "virtual" kv heads (n_head_kv above): virtual head hk reads
                                   // the tiles of real kv head hk / kv_div of the pool's n_head_kv_pool, its query heads hk * 8 + g are the real ones
    int   n_tile;             // MULTI: consecutive 128-token tiles one workgroup walks (n_splits = partials per (sequence, head) = ceil(tiles / n_tile))
    int   final_out;          // MULTI with n_splits == 1: the workgroup's result IS the head's output -> normalised straight into `out`, no partials, no merge launch
    float scale, softcap;
    // fused decode step (PREP): this step's un-rotated K / V rows, the new token's position, RoPE
    const float *   k_cur;    // [n_seq][n_head_kv][D]
    const float *   v_cur;
    const float4 *  rope_cache;  // [n_seq][D/2] (cos, sin): pkv_rope_cache rows of the new tokens, NULL = no RoPE
    int             has_rope;
    RopeDev         rope;
    unsigned long long * stamps; // diagnostic build only (STAMP): [workgroup][wave][PKV_DECODE_NSTAMP] s_memtime / s_memrealtime values
};

// ---- in-kernel phase stamps (cdna_hip_programming.md §7 "In-kernel stamps"): a SEPARATE instantiation of the kernel, reached only
// through pkv_attn_decode_stamps(); the product instantiations contain no stamp.  Stamp values leave through DecodeArgs::stamps only.
constexpr int NSTAMP = PKV_DECODE_NSTAMP;
template <int STAMP> struct Stamps {   // 0 = product (no code), 1 = stamps + forced vmcnt waits (phase shares), 2 = stamps only (the product's own overlap)
    unsigned long long t[STAMP ? NSTAMP : 1];
    __device__ __forceinline__ void at(int i) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long v;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory");
            t[i] = v;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __device__ __forceinline__ void real(int i) {   // 100 MHz wall clock: aligns workgroups with each other
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long v;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory");
            t[i] = v;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    template <int N> __device__ __forceinline__ void wait_vm() {   // all but the N youngest vector-memory operations have completed
        if constexpr (STAMP == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    }
};

// Q operand of QK^T.  It is converted ONCE per workgroup into LDS (f16 for f16 K — GGML_FP32_TO_FP16 of the reference's
// q_to_vec_dot; q8_0 for quantized K — quantize_row_q8_0 as the reference's CPU backend runs it for Q, AVX2 branch,
// /ref/ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862: d = amax/127 stored as f16, id = 127/amax, round-half-even), one
// element per thread, and every lane then reads the fragment that matches its 16-byte K chunk.
template <int KT, int D, int G> struct QShared;
template <int D, int G> struct QShared<PKV_TYPE_F16, D, G> { _Float16 h[G][D]; };
template <int KT, int D, int G> struct QSharedQ { signed char qs[G][D]; float d[G][D / PKV_QK]; int qsum[G][D / PKV_QK]; };
template <int D, int G> struct QShared<PKV_TYPE_Q8_0, D, G> : QSharedQ<PKV_TYPE_Q8_0, D, G> {};
template <int D, int G> struct QShared<PKV_TYPE_Q4_0, D, G> : QSharedQ<PKV_TYPE_Q4_0, D, G> {};

// per-lane fragment: f16 path = 8 halfs per head; quantized path = E int8 per head + f16-rounded block scale (+ block sum)
template <int KT, int D, int G> struct QFrag;
template <int D, int G> struct QFrag<PKV_TYPE_F16, D, G> {
    half2_t h[G][4];
    __device__ __forceinline__ void load(const QShared<PKV_TYPE_F16, D, G> & sq, int kc) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&sq.h[g][kc * 8]);
            h[g][0] = *reinterpret_cast<const half2_t *>(&v.x); h[g][1] = *reinterpret_cast<const half2_t *>(&v.y);
            h[g][2] = *reinterpret_cast<const half2_t *>(&v.z); h[g][3] = *reinterpret_cast<const half2_t *>(&v.w);
        }
    }
};
template <int KT, int D, int G> struct QFragQ {
    static constexpr int E  = KTr<KT, D>::E;   // 16 (q8_0 K) or 32 (q4_0 K) elements per lane
    static constexpr int NW = E / 4;
    int   qs[G][NW];
    float d[G];
    int   qsum[G];  // sum of the block's int8 (q4_0 K needs it for the -8 offset)
    __device__ __forceinline__ void load(const QShared<KT, D, G> & sq, int kc) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&sq.qs[g][kc * E + w * 4]);
                qs[g][w] = (int) v.x; qs[g][w + 1] = (int) v.y; qs[g][w + 2] = (int) v.z; qs[g][w + 3] = (int) v.w;
            }
            d[g]    = sq.d[g][kc * E / PKV_QK];
            qsum[g] = sq.qsum[g][kc * E / PKV_QK];
        }
    }
};
template <int D, int G> struct QFrag<PKV_TYPE_Q8_0, D, G> : QFragQ<PKV_TYPE_Q8_0, D, G> {};
template <int D, int G> struct QFrag<PKV_TYPE_Q4_0, D, G> : QFragQ<PKV_TYPE_Q4_0, D, G> {};

// partial dot of one lane's 16-byte K chunk against the Q fragment (all G heads)
template <int KT, int D, int G>
__device__ __forceinline__ void k_dot(const uint4 & kv, float kscale, const QFrag<KT, D, G> & qf, float (&s)[G]) {
    if constexpr (KT == PKV_TYPE_F16) {
        const half2_t * kh = reinterpret_cast<const half2_t *>(&kv);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2(kh[j], qf.h[g][j], acc, false);
            s[g] = acc;
        }
    } else if constexpr (KT == PKV_TYPE_Q8_0) {
        const int * kw = reinterpret_cast<const int *>(&kv);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_sdot4(kw[j], qf.qs[g][j], acc, false);
            s[g] = (float) acc * (kscale * qf.d[g]);
        }
    } else {  // q4_0: byte j = elem j (low nibble) | elem j+16 (high nibble), value - 8
        const uint32_t * kw = reinterpret_cast<const uint32_t *>(&kv);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_sdot4((int) (kw[j] & 0x0F0F0F0Fu), qf.qs[g][j], acc, false);
                acc = __builtin_amdgcn_sdot4((int) ((kw[j] >> 4) & 0x0F0F0F0Fu), qf.qs[g][4 + j], acc, false);
            }
            s[g] = (float) (acc - 8 * qf.qsum[g]) * (kscale * qf.d[g]);
        }
    }
}

// V side ("Pool layout"): tile = [tok/TL][D] dwords, every dword holds the TL tokens of one d
//   f16 : TL 2 (half2 = tokens t, t+1)   q8_0 : TL 4 (4 int8)   q4_0 : TL 8 (8 nibbles), then half d[tok/TL][D/32][TL].
// A lane owns 4 consecutive d (16 bytes) of TL tokens; D/4 lanes cover the head, so PV keeps only 4 accumulators per
// query head and the f16 case is a pure v_dot2c_f32_f16 stream (P packed as half2 per token pair).
template <int T> struct VTr;
template <> struct VTr<PKV_TYPE_F16>  { static constexpr int TL = 2; };
template <> struct VTr<PKV_TYPE_Q8_0> { static constexpr int TL = 4; };
template <> struct VTr<PKV_TYPE_Q4_0> { static constexpr int TL = 8; };

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }
constexpr int imax(int a, int b) { return a > b ? a : b; }

// One step of the logit reduce-scatter over the K lanes of a row: lanes whose DIST bit is clear keep the lower half
// of the value list, the others the upper half, and each adds what its partner (DPP pattern CTRL) holds for it.
template <int NV, int DIST, int CTRL>
__device__ __forceinline__ void rs_step(float * x, int lane) {
    if constexpr (NV >= 2) {
        const bool hi = lane & DIST;
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) {
            const float keep = hi ? x[j + NV / 2] : x[j];
            const float send = hi ? x[j] : x[j + NV / 2];
            x[j] = keep + dpp_mov<CTRL>(send);
        }
    } else {
        x[0] += dpp_mov<CTRL>(x[0]);
    }
}
// NV partial sums per lane, LPR lanes per row -> NVF = max(1, NV/LPR) finished sums per lane;
// finished x[j] is original index  j + NVF * (c >> (log2 LPR - NS)),  NS = min(log2 LPR, log2 NV), c = lane % LPR
template <int NV, int LPR>
__device__ __forceinline__ void reduce_scatter(float * x, int lane) {
    constexpr int n1 = NV, n2 = LPR >= 16 ? imax(1, n1 / 2) : n1, n3 = LPR >= 8 ? imax(1, n2 / 2) : n2,
                  n4 = LPR >= 4 ? imax(1, n3 / 2) : n3;
    if constexpr (LPR >= 16) rs_step<n1, 8, 0x140>(x, lane);  // row_mirror
    if constexpr (LPR >= 8)  rs_step<n2, 4, 0x141>(x, lane);  // row_half_mirror
    if constexpr (LPR >= 4)  rs_step<n3, 2, 0x4E>(x, lane);   // quad_perm [2,3,0,1]
    if constexpr (LPR >= 2)  rs_step<n4, 1, 0xB1>(x, lane);   // quad_perm [1,0,3,2]
}

// all-reduce over the lanes that hold the same query head: lane bits [log2(64/TPW), 6)
template <int TPW_, bool IS_MAX>
__device__ __forceinline__ float head_reduce(float x) {
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
    if constexpr (TPW_ >= 64) x = op(x, dpp_mov<0xB1>(x));
    if constexpr (TPW_ >= 32) x = op(x, dpp_mov<0x4E>(x));
    x = op(x, dpp_mov<0x124>(x));  // row_ror:4
    x = op(x, dpp_mov<0x128>(x));  // row_ror:8   -> lanes i, i+4, i+8, i+12 of the row
    x = op(x, __shfl_xor(x, 16));
    x = op(x, __shfl_xor(x, 32));
    return x;
}

// ---- fused decode step (PREP): the per-token work that precedes attention runs inside the decode launch ------------
// A "row job" is one D-element f32 row held by one wave (lane i = elements 2i, 2i+1, the layout of pkv_rowcodec.h):
// jobs 0..G-1 are this kv head's query rows (RoPE, then the QK^T operand format straight into LDS: the rotated Q never
// goes back to memory), job G is the new token's K row (RoPE + quantize), job G+1 its V row (quantize).  The workgroup
// that owns the sequence's last block patches the new token's K / V bits into the registers it loaded from the pool
// (the pool slot itself is written at the end of the kernel, for the steps to come), so no launch and no memory round
// trip separates ggml_rope_ext + ggml_cpy from the attention that consumes them (/ref/src/llama-model.cpp:4091-4105).
template <int KT, int D, int G>
__device__ __forceinline__ void q_row_to_lds(QShared<KT, D, G> & sq, int g, int lane, float2 x) {
    if constexpr (KT == PKV_TYPE_F16) {
        const Q2 v = quantize<PKV_TYPE_F16>(x, lane);
        if (2 * lane < D) *reinterpret_cast<uint32_t *>(&sq.h[g][2 * lane]) = (uint32_t) (v.q0 & 0xFFFF) | ((uint32_t) v.q1 << 16);
    } else {
        const Q2 v = quantize<PKV_TYPE_Q8_0>(x, lane);     // the reference quantizes Q to q8_0 for every quantized K type
        int sum = v.q0 + v.q1;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
        if (2 * lane < D) {
            *reinterpret_cast<uint16_t *>(&sq.qs[g][2 * lane]) = (uint16_t) ((v.q0 & 0xFF) | ((v.q1 & 0xFF) << 8));
            if ((lane & 15) == 0) { sq.d[g][2 * lane / PKV_QK] = h_val(v.d); sq.qsum[g][2 * lane / PKV_QK] = sum; }
        }
    }
}
// the new token's K row in the K-tile row format (quants, then one f32 scale per 32-element block)
template <int KT, int D> struct KCur { __attribute__((aligned(16))) unsigned char q[D * 2]; float d[D / PKV_QK]; };
template <int KT, int D>
__device__ __forceinline__ void k_row_to_lds(KCur<KT, D> & sk, int lane, const Q2 & v) {
    const int e0 = 2 * lane;
    if constexpr (KT == PKV_TYPE_Q4_0) {   // byte j of a block = elem j | elem (j+16) << 4
        const int p0 = __shfl_xor(v.q0, 8), p1 = __shfl_xor(v.q1, 8);
        if (e0 >= D) return;
        if ((lane & 15) < 8)
            *reinterpret_cast<uint16_t *>(&sk.q[(e0 / PKV_QK) * 16 + 2 * (lane & 15)]) = (uint16_t) ((v.q0 | (p0 << 4)) | ((v.q1 | (p1 << 4)) << 8));
    } else {
        if (e0 >= D) return;
        if constexpr (KT == PKV_TYPE_F16) *reinterpret_cast<uint32_t *>(&sk.q[e0 * 2]) = (uint32_t) (v.q0 & 0xFFFF) | ((uint32_t) v.q1 << 16);
        else                              *reinterpret_cast<uint16_t *>(&sk.q[e0]) = (uint16_t) ((v.q0 & 0xFF) | ((v.q1 & 0xFF) << 8));
    }
    if (KT != PKV_TYPE_F16 && (lane & 15) == 0) sk.d[e0 / PKV_QK] = h_val(v.d);
}
// the new token's V row as V-tile dwords: dword d carries the token's bits at its place `sub` in the token group, zeros elsewhere
template <int D> struct VCur { __attribute__((aligned(16))) uint32_t w[D]; float d[D / PKV_QK]; };
template <int VT, int D>
__device__ __forceinline__ void v_row_to_lds(VCur<D> & sv, int lane, const Q2 & v, int sub) {
    const int e0 = 2 * lane;
    if (e0 >= D) return;
    constexpr int BITS = 32 / VTr<VT>::TL;
    constexpr uint32_t M = BITS == 32 ? 0xFFFFFFFFu : (1u << BITS) - 1u;
    sv.w[e0]     = ((uint32_t) v.q0 & M) << (BITS * sub);
    sv.w[e0 + 1] = ((uint32_t) v.q1 & M) << (BITS * sub);
    if (VT != PKV_TYPE_F16 && (lane & 15) == 0) sv.d[e0 / PKV_QK] = h_val(v.d);
}

template <int KT, int VT, int D, int TPWQ = 16>
struct Geo {
    using KT_ = KTr<KT, D>;
    static constexpr int K_LPR = KT_::LPR, K_RPL = 64 / K_LPR;
    static constexpr int V_LPR = D / 4, V_TL = VTr<VT>::TL, V_RPL = (64 / V_LPR) * V_TL;
    // quantized tiles are 2-4x fewer bytes per token: a wave that owns 16 tokens spends more instructions on its fixed work (softmax, reductions, the
    // strip, the merge) than on the dots.  TPWQ = 32 tokens per wave for them when the launch has workgroups to spare (many sequences: q8_0 0.56 -> 0.62,
    // q4_0 0.31 -> 0.36 of HBM peak at 32 sequences); at one sequence the 8-wave workgroup hides more latency (4.8 vs 5.1 us) and stays
    static constexpr int TPW = imax((KT != PKV_TYPE_F16 || VT != PKV_TYPE_F16) ? TPWQ : 16, imax(K_RPL, V_RPL));   // tokens per wave
    static constexpr int NW  = WG_TOK / TPW;                   // waves per workgroup
    static constexpr int K_NL = TPW / K_RPL, V_NL = TPW / V_RPL;
};

// ---- cross-row exchanges without LDS: gfx950's v_permlane16_swap / v_permlane32_swap (one VALU instruction each; the
// ds_bpermute they replace is an LDS round trip with its own wait) ------------------------------------------------------
__device__ __forceinline__ float vmaxf(float a, float b) {   // plain v_max_f32: fmaxf() of a bit-cast value costs two canonicalising v_max more
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// all-reduce over lane ^ 16 (SHIFT 16) or lane ^ 32 (SHIFT 32): swap(x, x) leaves the two partners' values in the two results
template <int SHIFT, bool IS_MAX> __device__ __forceinline__ float xreduce(float x) {
    const unsigned u = __float_as_uint(x);
    float p, q;
    if constexpr (SHIFT == 16) { auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); p = __uint_as_float(r[0]); q = __uint_as_float(r[1]); }
    else                       { auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); p = __uint_as_float(r[0]); q = __uint_as_float(r[1]); }
    return IS_MAX ? vmaxf(p, q) : p + q;
}
// a (lanes 0-31: a_lo + a_hi) | b (lanes 32-63: b_lo + b_hi): two xor-32 sums for one swap and one add
__device__ __forceinline__ float pair_sum32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// all-reduce over the lanes that hold the same query head: lane bits [log2(64/TPW), 6)
template <int TPW_, bool IS_MAX>
__device__ __forceinline__ float head_reduce2(float x) {
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
    if constexpr (TPW_ >= 64) x = op(x, dpp_mov<0xB1>(x));
    if constexpr (TPW_ >= 32) x = op(x, dpp_mov<0x4E>(x));
    x = op(x, dpp_mov<0x124>(x));  // row_ror:4
    x = op(x, dpp_mov<0x128>(x));  // row_ror:8   -> lanes i, i+4, i+8, i+12 of the row
    x = xreduce<16, IS_MAX>(x);
    x = xreduce<32, IS_MAX>(x);
    return x;
}

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rs, int off) {
#ifndef PKV_KV_AUX
#define PKV_KV_AUX 2
#endif
    const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, PKV_KV_AUX);   // aux 2 = nt (experiment)
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ======================================================================================================================
// One workgroup = one (sequence, kv head, 128-token split); wave w owns tokens [16 w, 16 w + 16) of it.  Schedule of a wave
// (r02; in-kernel stamps: profiles/r02_decode_stamps_*.txt — the r01 kernel spent 3.7 of its 6.2 us AFTER its loads were
// issued, on ~1000 executed instructions per wave, while a CU accepts only ~24 KiB of loads in flight):
//   1. scalar: table entry, sequence length.  K and V tiles are addressed through buffer descriptors whose num_records ends
//      at the sequence's last token: rows past the end read as zero in hardware (no clamps, no 64-bit address math; a
//      wave's 16 K rows / 8 V token pairs are 4 KiB of consecutive bytes: ONE voffset + immediates).
//   2. "row jobs" spread over the waves (wave j: query row j -> RoPE -> QK^T operand format -> LDS; wave G / G+1: the new
//      token's K / V row in the fused step), the wave's K loads go out FIRST, the job's arithmetic runs under them.
//   3. ONE early workgroup barrier (Q operand visible), then QK^T consumes K chunk i as it lands and issues V chunk i in its
//      place: the wave never sits in front of a full memory queue with arithmetic left to do.
//   4. DPP reduce-scatter -> one logit per lane; softmax per wave (row_ror + permlane swaps); P -> the wave's LDS strip;
//      PV = dot2 stream; the two half-wave sums by permlane32_swap pairs.
//   5. second barrier, cross-wave merge (one output element per thread), one (m, l, O) partial per workgroup.
#ifndef PKV_OCC
#define PKV_OCC 1
#endif
template <int KT, int VT, int D, int G, bool SOFTCAP, bool SLOTPOS, bool FUSED, bool PREP, int STAMP = 0, int TPWQ = 16, bool MULTI = false>
__global__ void __launch_bounds__((64 * Geo<KT, VT, D, TPWQ>::NW), PKV_OCC) attn_decode_kernel(const int32_t * __restrict__ p_table, const int32_t * __restrict__ p_lens, int p_max_blocks, int p_bs_shift,
                                                                                                   const DecodeArgs a) {
    // the four leading arguments (= a.block_table, a.seq_lens, a.max_blocks, a.bs_shift) are what the FIRST dependent loads need: as explicit kernel
    // arguments they are preloaded into SGPRs at dispatch (-mllvm -amdgpu-kernarg-preload-count, csrc/Makefile), one scalar round trip less on the
    // critical path to the KV addresses
    Stamps<STAMP> ts;
    ts.real(0);
    ts.at(1);
    using GE = Geo<KT, VT, D, TPWQ>;
    using KT_ = KTr<KT, D>;
    constexpr int K_LPR = GE::K_LPR, K_RPL = GE::K_RPL, K_NL = GE::K_NL;
    constexpr int V_LPR = GE::V_LPR, V_TL = GE::V_TL, V_RPL = GE::V_RPL, V_NL = GE::V_NL;
    constexpr int TPW = GE::TPW, NW = GE::NW, NT = 64 * NW;
    constexpr int NV = K_NL * G, NS = ilog2(K_LPR) < ilog2(NV) ? ilog2(K_LPR) : ilog2(NV), NVF = NV >> NS;
    constexpr bool P16 = VT == PKV_TYPE_F16;                   // P travels as half2 token pairs for the dot2 PV
    // quantized V (r02): the weight of a cell is folded with the cell's block scale BEFORE it meets the quants — P''[t][g][b] = f16(p[t][g] * d_V[t][b]),
    // one rounding — and travels as half2 token pairs like the f16 case; the int8 / nibble quants become f16 by a byte permute against the magic
    // 0x6400 (1024 + u) and one packed add, so that P.V is a v_dot2c_f32_f16 stream for every V type (r01: 16 int->float converts + 64 FMAs per 16 V bytes)
    constexpr bool PQ = VT != PKV_TYPE_F16;
    constexpr int K_ROWB = KT == PKV_TYPE_F16 ? D * 2 : (KT == PKV_TYPE_Q8_0 ? D : D / 2);   // quant bytes per token row of a K tile
    constexpr int NB = D / PKV_QK;

    __shared__ __attribute__((aligned(16))) float sP[NW][TPW * G / 2 * (VT != PKV_TYPE_F16 ? D / PKV_QK : 1)];  // per-wave weights, half2 token pairs [x V block]
    __shared__ __attribute__((aligned(16))) float sO[NW][G][D];                     // per-wave partial outputs
    __shared__ __attribute__((aligned(16))) float sML[NW][G][2];                    // per-wave (max, sum)
#ifndef PKV_WGMAX
#define PKV_WGMAX 0
#endif
#if PKV_WGMAX
    __shared__ __attribute__((aligned(16))) float sMX[G][NW];                       // per-wave head maxima (mid barrier)
#endif
    __shared__ __attribute__((aligned(16))) QShared<KT, D, G> sQ;                   // QK^T operand (row jobs)
    __shared__ KCur<KT, D> sKc;                                                     // the new token's rows (fused step)
    __shared__ VCur<D>     sVc;

    static_assert(!MULTI || (!SOFTCAP && !SLOTPOS && !FUSED && STAMP == 0), "the multi-tile form exists for the plain variants only");
    // MULTI (r03): one workgroup walks a.n_tile CONSECUTIVE 128-token tiles of its (sequence, kv head).  The Q operand is prepared once, every wave keeps an online
    // softmax state (running max / sum per head, rescaled accumulators) over its 16 tokens of each tile, the K rows of tile i+1 are requested as soon as QK^T has
    // consumed tile i's (behind tile i's V rows: the memory queue never runs dry), and there is ONE cross-wave merge and ONE partial per workgroup — or, when the
    // workgroup covers the whole sequence (final_out), the head's normalised output itself: no partials and no merge launch.  No barrier inside the loop.
    __shared__ __attribute__((aligned(16))) float sAl[MULTI ? NW : 1][G];          // per-wave rescale factors exp(m_old - m_new) of the running accumulators

    const int split = blockIdx.x, hk = blockIdx.y, s = blockIdx.z;
    const int tile0 = MULTI ? split * a.n_tile : split;         // first 128-token tile of this workgroup
    const int base0 = tile0 * WG_TOK;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // the two scalar loads every KV address depends on are independent of each other: issue both, then branch
    // (the table entry of a split past the end of the sequence is in bounds — grid.x <= max_blocks — and unused)
    const int bs_mask = (1 << p_bs_shift) - 1;
    const int32_t * trow = p_table + (size_t) s * p_max_blocks;
    const int tile_cap = (p_max_blocks << p_bs_shift) - 1;      // last tile index the table row can address
    const int phys0   = trow[tile0 >> p_bs_shift];
    const int seq_len = p_lens[s];
    [[maybe_unused]] int phys_n = 0;                            // MULTI: table entry of the NEXT tile, looked up one tile ahead
    if constexpr (MULTI) phys_n = trow[min(tile0 + 1, tile_cap) >> p_bs_shift];

    // ---- row jobs, loads: they need kernel arguments only, so they go out while the table entry is still on its way, and —
    //      barrier 0 below — before ANY wave of the workgroup has put a K load into the CU's memory queue: a 512-byte row that
    //      queues behind 28 KiB of other waves' K chunks comes back a microsecond late, and the whole workgroup waits for the
    //      Q operand (stamps: profiles/r02_decode_stamps_*.txt).  Every load is unconditional and in bounds — a load behind a
    //      branch makes hipcc drain the whole queue at the join.
    constexpr int NJ = G + (PREP ? 2 : 0), NJW = (NJ + NW - 1) / NW;
    float2 jx[NJW];
    float4 jcs = make_float4(1.0f, 0.0f, 1.0f, 0.0f);           // this lane's (cos, sin) entries of the new token's row
#pragma unroll
    for (int k = 0; k < NJW; ++k) {
        const int job = min(wv + k * NW, NJ - 1);
        const float * src = a.q + (size_t) s * a.q_stride_tok + (size_t) (hk * G + min(job, G - 1)) * a.q_stride_head;
        if constexpr (PREP) { if (job >= G) src = (job == G ? a.k_cur : a.v_cur) + ((size_t) s * a.n_head_kv + hk) * D; }
        jx[k] = *reinterpret_cast<const float2 *>(src + min(2 * lane, D - 2));
    }
    if constexpr (PREP) {   // without RoPE the row is not used: any readable 16-byte-aligned address will do (q is)
        const float4 * cs = a.has_rope ? reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(a.rope_cache) + (size_t) s * (D / 2) + rope_cache_base<D>(lane, a.rope))
                                       : reinterpret_cast<const float4 *>(a.q);
        jcs = *cs;
    }
    __builtin_amdgcn_sched_barrier(0);  // program order = issue order: the rows are OLDER than the K stream (their wait must not drain it)
    // everything the KV addresses need is requested before the branch (hipcc otherwise sinks the table load and the
    // remaining kernel-argument loads below it, one scalar round trip each)
    asm volatile("" ::"s"(phys0), "s"(seq_len), "s"(a.k_pool), "s"(a.v_pool), "s"(a.q));
    ts.at(2);                     // kernel arguments, table entry and sequence length are in SGPRs; row loads issued
    if (base0 >= seq_len) {       // workgroup-uniform
        if constexpr (MULTI) {    // an empty sequence whose only workgroup owns the output: zeros, as the merge launch writes them
            if (a.final_out && tid < G * D / 4)
                *reinterpret_cast<float4 *>(&a.out[((size_t) s * a.n_head + hk * G) * D + tid * 4]) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        return;
    }
    __builtin_amdgcn_s_barrier(); // barrier 0 (no memory to order: a bare s_barrier): every wave's row loads are in the queue
    __builtin_amdgcn_sched_barrier(0);

    const int n_tiles_seq = (seq_len + WG_TOK - 1) / WG_TOK;
    const int n_it = MULTI ? min(a.n_tile, n_tiles_seq - tile0) : 1;   // tiles this workgroup walks (>= 1)
    const bool is_last = PREP && tile0 + n_it == n_tiles_seq;           // the new token sits in this workgroup's last tile
    const size_t tb_k = tile_bytes(KT, a.block_size, D), tb_v = tile_bytes(VT, a.block_size, D);
    const size_t tq_k = tile_quant_bytes(KT, a.block_size, D), tq_v = tile_quant_bytes(VT, a.block_size, D);
    // buffer descriptors of a tile's 128-token range: records end at the last valid token / token group (n_valid 0: every load returns 0)
    const int hk_pool = a.kv_div > 1 ? hk / a.kv_div : hk;
    auto tile_k  = [&](int phys) { return a.k_pool + ((size_t) phys * a.n_head_kv_pool + hk_pool) * tb_k; };
    auto tile_v  = [&](int phys) { return a.v_pool + ((size_t) phys * a.n_head_kv_pool + hk_pool) * tb_v; };
    auto mk_rs_k = [&](int phys, int boff, int nv) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(tile_k(phys)) + (size_t) boff * K_ROWB, 0, nv * K_ROWB, 0x00020000); };
    auto mk_rs_ks = [&](int phys, int boff, int nv) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(tile_k(phys)) + tq_k + (size_t) boff * NB * 2, 0, nv * NB * 2, 0x00020000); };
    auto mk_rs_v = [&](int phys, int boff, int nv) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(tile_v(phys)) + (size_t) (boff / V_TL) * D * 4, 0, ((nv + V_TL - 1) / V_TL) * D * 4, 0x00020000); };
    auto mk_rs_vs = [&](int phys, int boff, int nv) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(tile_v(phys)) + tq_v + (size_t) (boff / V_TL) * NB * V_TL * 2, 0,
                                                 ((nv + V_TL - 1) / V_TL) * NB * V_TL * 2, 0x00020000); };

    // ---- the wave's K rows: 16 tokens = K_NL consecutive 1 KiB chunks ---------------------------------------------------
    const int kr = lane / K_LPR, kc = lane % K_LPR;
    const int vl = lane / V_LPR, vc = lane % V_LPR;
    const int k_off = (wave * TPW * K_LPR + lane) * 16;
    const int v_off = (wave * (TPW / V_TL) * V_LPR + lane) * 16;
    uint4 kreg[K_NL];
    float ksc[K_NL];                                            // single-tile form: block scales as floats
    [[maybe_unused]] unsigned short ksb[MULTI ? K_NL : 1];      // MULTI: their f16 bits (converted where they are used: no wait at the prefetch)
    constexpr int NVF_ = ((K_NL * G) >> (ilog2(K_LPR) < ilog2(K_NL * G) ? ilog2(K_LPR) : ilog2(K_NL * G)));
    [[maybe_unused]] unsigned short vsb[PQ ? NVF_ : 1][PQ ? NB : 1];
    auto issue_k = [&](__amdgpu_buffer_rsrc_t rk, [[maybe_unused]] __amdgpu_buffer_rsrc_t rks) {
#pragma unroll
        for (int i = 0; i < K_NL; ++i) {
            kreg[i] = buf_load16(rk, k_off + i * 1024);
            if constexpr (!MULTI) ksc[i] = 0.0f;
            if constexpr (KT != PKV_TYPE_F16) {
                const int row = wave * TPW + i * K_RPL + kr;
                const unsigned short hb = __builtin_amdgcn_raw_buffer_load_b16(rks, (row * NB + kc * KT_::E / PKV_QK) * 2, 0, 0);
                if constexpr (MULTI) ksb[i] = hb; else ksc[i] = h_val(hb);   // a row past the end reads scale 0
            }
        }
    };
    // quantized V: the block scales of the cells whose logits this lane will own (NVF cells x NB blocks, 2 bytes each), behind the K stream
    auto issue_vs = [&]([[maybe_unused]] __amdgpu_buffer_rsrc_t rvs) {
        if constexpr (PQ) {
            constexpr int NS_ = ilog2(K_LPR) < ilog2(K_NL * G) ? ilog2(K_LPR) : ilog2(K_NL * G);
            const int cidx_ = NVF_ * (kc >> (ilog2(K_LPR) - NS_));
#pragma unroll
            for (int j = 0; j < NVF_; ++j) {
                const int t = wave * TPW + ((cidx_ + j) / G) * K_RPL + kr;          // cell inside the tile's 128
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    vsb[j][b] = __builtin_amdgcn_raw_buffer_load_b16(rvs, (((t / V_TL) * NB + b) * V_TL + t % V_TL) * 2, 0, 0);   // past the end: 0
            }
        }
    };
    int phys = phys0, tile = tile0;                             // (scalar state of the tile loop: kept in SGPRs by construction — a table entry fetched with a
    int n_valid = min(WG_TOK, seq_len - base0);                 //  VECTOR load inside the loop made hipcc wait for the whole K prefetch right behind it; >= 1)
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_ks, rs_vs;
    if constexpr (KT != PKV_TYPE_F16) rs_ks = mk_rs_ks(phys, (tile & bs_mask) * WG_TOK, n_valid);
    if constexpr (VT != PKV_TYPE_F16) rs_vs = mk_rs_vs(phys, (tile & bs_mask) * WG_TOK, n_valid);
    issue_k(mk_rs_k(phys, (tile & bs_mask) * WG_TOK, n_valid), rs_ks);
    issue_vs(rs_vs);
    __builtin_amdgcn_sched_barrier(0);
    ts.at(3);                           // row and K loads issued

    // ---- row jobs: arithmetic under the K stream, results into LDS -------------------------------------------------------
    Q2 curq[NJW];
#pragma unroll
    for (int k = 0; k < NJW; ++k) {
        const int job = wv + k * NW;                            // wave-uniform
        curq[k] = Q2{0, 0, 0};
        if (job < G) {                                          // ggml_rope_ext(Qcur), then q_to_vec_dot
            float2 x = jx[k];
            if constexpr (PREP) { if (a.has_rope) x = rope_row_cached<D>(x, lane, jcs, a.rope); }
            q_row_to_lds<KT, D, G>(sQ, job, lane, x);
        } else if constexpr (PREP) {
            if (job == G && is_last) {                          // ggml_rope_ext(Kcur) + ggml_cpy -> k cache
                float2 x = jx[k];
                if (a.has_rope) x = rope_row_cached<D>(x, lane, jcs, a.rope);
                curq[k] = quantize<KT>(x, lane);
                k_row_to_lds<KT, D>(sKc, lane, curq[k]);
            } else if (job == G + 1 && is_last) {               // ggml_cpy(Vcur) -> v cache
                curq[k] = quantize<VT>(jx[k], lane);
                v_row_to_lds<VT, D>(sVc, lane, curq[k], (seq_len - 1) % V_TL);   // (tiles start on multiples of 128: the token's place in its group)
            }
        }
    }
#ifndef PKV_VSCHED
#define PKV_VSCHED 0
#endif
    uint4 vreg[V_NL];
    __amdgpu_buffer_rsrc_t rs_v = mk_rs_v(phys, (tile & bs_mask) * WG_TOK, n_valid);
    auto issue_v = [&](int i) {
        __builtin_amdgcn_sched_barrier(0);                      // program order = issue order
        vreg[i] = buf_load16(rs_v, v_off + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (PKV_VSCHED == 2 && !MULTI) {    // every V chunk right behind the row job, ahead of the barrier (all waves: the counts must not depend on the path)
#pragma unroll
        for (int i = 0; i < V_NL; ++i) issue_v(i);
    }
    __syncthreads();                    // the workgroup's FIRST barrier: Q operand (and the new token's rows) visible

    // state that lives across the tiles of the MULTI form (single-tile: written once)
    [[maybe_unused]] float mrun[NVF], lrun[NVF];              // running max / sum of this lane's heads (every lane of a head's group holds the same values)
#pragma unroll
    for (int j = 0; j < NVF; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.0f; }
    float acc[G][4];                    // PV: 4 f32 accumulators per query head, lane owns 4 d of TL tokens per load
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[g][j] = 0.0f;
    const bool active1 = wv * TPW < n_valid;                    // single-tile form, wave-uniform: a wave past the end only runs its row job
    QFrag<KT, D, G> qf;
    if (MULTI || active1) qf.load(sQ, kc);

    if (MULTI || active1) {   // ======== the wave's 16 tokens of every tile ========
    for (int it = 0; it < n_it; ++it) {
    if constexpr (MULTI) {
        tile    = __builtin_amdgcn_readfirstlane(tile0 + it);
        n_valid = __builtin_amdgcn_readfirstlane(min(WG_TOK, seq_len - tile * WG_TOK));
    }
    const int  base    = tile * WG_TOK;
    const bool last_it = !MULTI || it + 1 == n_it;
    const bool owner   = is_last && last_it && wv == (n_valid - 1) / TPW;   // the new token's row is among this wave's 16 tokens of this tile
    [[maybe_unused]] int kpatch = -1;
    if constexpr (PREP) {
        if (owner) {   // the new token's K bits replace whatever the pool slot held (its store happens at the end of the kernel)
            const int tl = n_valid - 1;
#pragma unroll
            for (int i = 0; i < K_NL; ++i)
                if (wave * TPW + i * K_RPL + kr == tl) {
                    kreg[i] = *reinterpret_cast<const uint4 *>(&sKc.q[kc * 16]);
                    if constexpr (KT != PKV_TYPE_F16) { if constexpr (MULTI) kpatch = i; else ksc[i] = sKc.d[kc * KT_::E / PKV_QK]; }
                }
        }
    }
    ts.at(4);                           // Q operand in registers
    static_assert(STAMP == 0 || (KT == PKV_TYPE_F16 && VT == PKV_TYPE_F16), "the stamp build counts the f16 load stream only");

    // ---- QK^T: K chunk i is consumed as it lands, V chunk i is requested in its place --------------------------------------
    float x[NV];
#pragma unroll
    for (int i = 0; i < (K_NL > V_NL ? K_NL : V_NL); ++i) {
        if (i < K_NL) {
            float sv[G];
            float sc = 0.0f;
            if constexpr (MULTI) {
                if constexpr (KT != PKV_TYPE_F16) { sc = h_val(ksb[i]); if constexpr (PREP) { if (kpatch == i) sc = sKc.d[kc * KT_::E / PKV_QK]; } }
            } else sc = ksc[i];
            k_dot<KT, D, G>(kreg[i], sc, qf, sv);
#pragma unroll
            for (int g = 0; g < G; ++g) x[i * G + g] = sv[g];
        }
        if ((MULTI || PKV_VSCHED == 0) && i < V_NL) issue_v(i); // V chunk i goes out when K chunk i has been consumed
    }
    [[maybe_unused]] int n_valid_n = 0, phys_nn = 0;
    if constexpr (MULTI) {
        // the NEXT tile's K rows, behind this tile's V rows and ahead of this tile's softmax + PV.  Branch-free: past the workgroup's last tile the
        // descriptor has no records (every load returns 0 without touching memory) — a load behind a branch would make hipcc drain the queue at the join
        __builtin_amdgcn_sched_barrier(0);
        // the table entry of the tile after next FIRST: if hipcc makes it a vector load (it does behind the loop's LDS fence), it must be OLDER than the K
        // prefetch — the wait for it at the end of the iteration then leaves the K rows in flight
        phys_nn = trow[__builtin_amdgcn_readfirstlane(min(tile0 + it + 2, tile_cap) >> p_bs_shift)];
        __builtin_amdgcn_sched_barrier(0);
        n_valid_n = last_it ? 0 : min(WG_TOK, seq_len - base - WG_TOK);
        const int boff_n = ((tile + 1) & bs_mask) * WG_TOK;
        [[maybe_unused]] __amdgpu_buffer_rsrc_t rks_n;
        if constexpr (KT != PKV_TYPE_F16) rks_n = mk_rs_ks(phys_n, boff_n, n_valid_n);
        issue_k(mk_rs_k(phys_n, boff_n, n_valid_n), rks_n);
        __builtin_amdgcn_sched_barrier(0);
    }
    reduce_scatter<NV, K_LPR>(x, lane);
    ts.at(5);                           // QK^T done, every V load issued

    const bool use_mask = SLOTPOS && a.mask != nullptr;       // launch-uniform: explicit mask row instead of cell positions
    const int32_t * spos = SLOTPOS && !use_mask ? a.slot_pos + (size_t) s * a.max_blocks * a.block_size + base : nullptr;
    const _Float16 * mrow = use_mask ? a.mask + (size_t) s * a.mask_stride + base : nullptr;
    const int qpos = SLOTPOS && !use_mask ? a.q_pos[s] : 0;
    const int cidx = NVF * (kc >> (ilog2(K_LPR) - NS));       // original index of x[0]
    float mg[NVF], pj[NVF];
    int   tok[NVF];
#pragma unroll
    for (int j = 0; j < NVF; ++j) {
        const int idx = cidx + j;
        tok[j] = wave * TPW + (idx / G) * K_RPL + kr;         // token inside the tile's 128
        bool ok = tok[j] < n_valid;
        float mv = 0.0f;
        if constexpr (SLOTPOS) {
            if (use_mask) { mv = (float) mrow[min(tok[j], n_valid - 1)]; ok = ok && mv != -INFINITY; }   // :12349-12351
            else          { const int p = spos[min(tok[j], n_valid - 1)]; ok = ok && p >= 0 && p <= qpos; }
        }
        float v = x[j] * a.scale;                             // :12359
        if constexpr (SOFTCAP) v = a.softcap * tanhf(v);      // :12361-12363 (scale pre-divided on the host)
        v += mv;                                              // :12365 (slope 1: ALiBi goes through pkv_attn_prefill)
        x[j]  = ok ? v : -INFINITY;                           // mask == -inf -> skipped cell (:12349)
        mg[j] = head_reduce2<TPW, true>(x[j]);                // wave-local max of this head
#if PKV_WGMAX
        if ((lane >> ilog2(64 / TPW)) == 0) sMX[idx % G][wv] = mg[j];
    }
    __syncthreads();                    // mid barrier: every wave's head maxima are in LDS
#pragma unroll
    for (int j = 0; j < NVF; ++j) {
        const int g = (cidx + j) % G;
        float M = sMX[g][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, sMX[g][w]);
        mg[j] = M;                                            // the WORKGROUP's max: every wave weighs against the same one
#endif
        if constexpr (MULTI) {          // online softmax over the tiles: weigh against the running max, rescale what has been accumulated
            const float m_new = fmaxf(mrun[j], mg[j]);
            const float al = mrun[j] == m_new ? 1.0f : __expf(mrun[j] - m_new);   // (first tile: exp(-inf) = 0 on accumulators that are 0)
            if ((lane >> ilog2(64 / TPW)) == 0) sAl[wave][idx % G] = al;
            lrun[j] *= al;
            mrun[j] = m_new;
            mg[j]   = m_new;
        }
        pj[j] = mg[j] == -INFINITY ? 0.0f : __expf(x[j] - mg[j]);
        if constexpr (P16) pj[j] = (float) (_Float16) pj[j];  // the weight PV really applies; l is summed from the same values
    }
    // probabilities -> this wave's LDS strip, in the layout the PV loop reads (when there are fewer logits than K
    // lanes the low lane bits hold duplicates: they write the same value to the same place)
#pragma unroll
    for (int j = 0; j < NVF; ++j) {
        const int g = (cidx + j) % G, t = tok[j] - wave * TPW;
        if constexpr (P16) reinterpret_cast<_Float16 *>(sP[wave])[((t >> 1) * G + g) * 2 + (t & 1)] = (_Float16) pj[j];
        else {   // pair of a cell: q8_0 (t, t + 1), q4_0 (t, t + 4) inside its group of 8 — the cells one V dword's byte / nibble pairs hold
            const int pair = V_TL == 4 ? (t >> 1) : ((t >> 3) * 4 + (t & 3)), el = V_TL == 4 ? (t & 1) : ((t >> 2) & 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float sc = h_val(vsb[j][b]);
                if constexpr (PREP) { if (is_last && last_it && tok[j] == n_valid - 1) sc = sVc.d[b]; }   // the new token: its scale is not in the pool yet
                const float pw = pj[j] == 0.0f ? 0.0f : pj[j] * sc;      // a stale scale of a cell past the end must not reach 0 * NaN
                reinterpret_cast<_Float16 *>(sP[wave])[((pair * NB + b) * G + g) * 2 + el] = (_Float16) pw;
            }
        }
        const float l = head_reduce2<TPW, false>(pj[j]);      // sums over the token bits of the lane id only
        if constexpr (MULTI) lrun[j] += l;
        else if ((lane >> ilog2(64 / TPW)) == 0) *reinterpret_cast<float2 *>(&sML[wave][g][0]) = make_float2(mg[j], l);
    }
    if constexpr (MULTI && PQ) {        // the next tile's V scales: this tile's have just been used
        __builtin_amdgcn_sched_barrier(0);
        issue_vs(mk_rs_vs(phys_n, ((tile + 1) & bs_mask) * WG_TOK, n_valid_n));
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // LDS strip is read back by other lanes of this wave only
    __builtin_amdgcn_wave_barrier();
    ts.at(6);                           // logits, softmax weights and P strip done
    ts.template wait_vm<0>();
    ts.at(7);                           // (stamp mode 1 only: the wave's V rows have landed)

    if constexpr (PREP) {
        if (owner) {   // the new token's V bits
            const int tl = n_valid - 1;
            constexpr int BITS = 32 / V_TL;
            const int      gcur = tl / V_TL, sub = tl % V_TL;
            const uint32_t keep = ~((BITS == 32 ? 0xFFFFFFFFu : (1u << BITS) - 1u) << (BITS * sub));
#pragma unroll
            for (int i = 0; i < V_NL; ++i) {
                const int tw = wave * TPW + i * V_RPL + vl * V_TL;
                if (tw / V_TL == gcur) {
                    const uint4 nv = *reinterpret_cast<const uint4 *>(&sVc.w[vc * 4]);
                    vreg[i].x = (vreg[i].x & keep) | nv.x; vreg[i].y = (vreg[i].y & keep) | nv.y;
                    vreg[i].z = (vreg[i].z & keep) | nv.z; vreg[i].w = (vreg[i].w & keep) | nv.w;
                }
            }
        }
    }

    // ---- PV: lane owns 4 d of TL tokens per load ------------------------------------------------------------------------
    if constexpr (MULTI) {              // what the earlier tiles accumulated is weighed against the new running max (wave-uniform values)
        float al[G];
        bool  resc = false;
#pragma unroll
        for (int g = 0; g < G; ++g) { al[g] = sAl[wave][g]; resc = resc || al[g] != 1.0f; }
        if (resc) {                     // after the first tiles the running max rarely moves
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[g][j] *= al[g];
        }
    }
#pragma unroll
    for (int i = 0; i < V_NL; ++i) {
        const int t0 = i * V_RPL + vl * V_TL;                 // first token (inside the wave) of this lane-load
        const uint32_t * vw = reinterpret_cast<const uint32_t *>(&vreg[i]);
        if constexpr (P16) {
            half2_t pp[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pp[g] = reinterpret_cast<const half2_t *>(sP[wave])[(t0 >> 1) * G + g];
            uint32_t w[4] = {vw[0], vw[1], vw[2], vw[3]};
            if (wave * TPW + t0 + 1 >= n_valid) {             // odd tail: the pair's second token is not part of the sequence
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] &= 0x0000FFFFu;
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[g][j] = __builtin_amdgcn_fdot2(*reinterpret_cast<const half2_t *>(&w[j]), pp[g], acc[g][j], false);
        } else {
            constexpr int NP = V_TL / 2;                      // token pairs per V dword: q8_0 (t0, t0+1) (t0+2, t0+3); q4_0 (t0+k, t0+k+4), k = 0..3
            const int p0 = V_TL == 4 ? (t0 >> 1) : (t0 >> 3) * 4, blk = vc / 8;   // the lane's 4 d sit in V block vc * 4 / 32
            half2_t pp[NP][G];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr)
#pragma unroll
                for (int g = 0; g < G; ++g) pp[pr][g] = reinterpret_cast<const half2_t *>(sP[wave])[((p0 + pr) * NB + blk) * G + g];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half2_t h[NP];
                if constexpr (VT == PKV_TYPE_Q8_0) {          // int8 b -> offset binary b + 128 -> f16 bits 0x6400 | u = 1024 + u -> minus 1152
                    const uint32_t y = vw[j] ^ 0x80808080u;
                    const uint32_t a01 = __builtin_amdgcn_perm(0x64646464u, y, 0x04010400u), a23 = __builtin_amdgcn_perm(0x64646464u, y, 0x04030402u);
                    const half2_t off = {(_Float16) -1152.0f, (_Float16) -1152.0f};
                    h[0] = *reinterpret_cast<const half2_t *>(&a01) + off;
                    h[1] = *reinterpret_cast<const half2_t *>(&a23) + off;
                } else {                                      // nibbles k and k + 4: 1024 + n -> minus 1032 (the -8 of q4_0 included)
                    const half2_t off = {(_Float16) -1032.0f, (_Float16) -1032.0f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t a = ((vw[j] >> (4 * k)) & 0x000F000Fu) | 0x64006400u;
                        h[k] = *reinterpret_cast<const half2_t *>(&a) + off;
                    }
                }
#pragma unroll
                for (int pr = 0; pr < NP; ++pr)
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][j] = __builtin_amdgcn_fdot2(h[pr], pp[pr][g], acc[g][j], false);
            }
        }
    }
    if constexpr (MULTI) {              // on to the next tile (scalar state only)
        if (!last_it) {
            phys = __builtin_amdgcn_readfirstlane(phys_n); phys_n = __builtin_amdgcn_readfirstlane(phys_nn);
            rs_v = mk_rs_v(phys, ((tile + 1) & bs_mask) * WG_TOK, n_valid_n);
        }
    }
    }   // tiles
    if constexpr (MULTI) {              // the wave's (max, sum) over all its tiles
        const int cidx = NVF * (kc >> (ilog2(K_LPR) - NS));
        if ((lane >> ilog2(64 / TPW)) == 0) {
#pragma unroll
            for (int j = 0; j < NVF; ++j) *reinterpret_cast<float2 *>(&sML[wave][(cidx + j) % G][0]) = make_float2(mrun[j], lrun[j]);
        }
    }
    // sum the lane-rows of the wave (lanes with equal vc), then hand the wave's partial to the workgroup
    if constexpr (V_LPR <= 16) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][j] = xreduce<16, false>(acc[g][j]);
    }
    if constexpr (G % 2 == 0) {   // heads in pairs: lanes 0-31 end with head g's totals, lanes 32-63 with head g+1's; every lane stores
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            float4 o;
            o.x = pair_sum32(acc[g][0], acc[g + 1][0]); o.y = pair_sum32(acc[g][1], acc[g + 1][1]);
            o.z = pair_sum32(acc[g][2], acc[g + 1][2]); o.w = pair_sum32(acc[g][3], acc[g + 1][3]);
            if (V_LPR == 32 || (lane & 16) == 0) *reinterpret_cast<float4 *>(&sO[wave][g + (lane >> 5)][vc * 4]) = o;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 o;
            o.x = xreduce<32, false>(acc[g][0]); o.y = xreduce<32, false>(acc[g][1]);
            o.z = xreduce<32, false>(acc[g][2]); o.w = xreduce<32, false>(acc[g][3]);
            if (vl == 0) *reinterpret_cast<float4 *>(&sO[wave][g][vc * 4]) = o;
        }
    }
    ts.at(8);                           // PV and the cross-lane sums done, the wave's partial is in LDS
    } else {        // ======== a wave past the end of the sequence: the empty partial ========
        if (lane < G) *reinterpret_cast<float2 *>(&sML[wv][lane][0]) = make_float2(-INFINITY, 0.0f);
        for (int e = lane; e < G * D; e += 64) sO[wv][e / D][e % D] = 0.0f;
#if PKV_WGMAX
        if (lane < G) sMX[lane][wv] = -INFINITY;
        __syncthreads();                // the mid barrier of the active waves
#endif
    }
    // what the epilogue below knows the workgroup's LAST tile by
    const int boff = (tile & bs_mask) * WG_TOK;
    const char * ktile = tile_k(phys);
    const char * vtile = tile_v(phys);
    __syncthreads();
    ts.at(9);                           // every wave of the workgroup has arrived

    // ---- merge the waves (each has its own max) and publish one (m, l, O) partial per workgroup -------------------
#ifndef PKV_PART_WIDE
#define PKV_PART_WIDE 1
#endif
#ifndef PKV_KV_AUX
#define PKV_KV_AUX 2
#endif
    static_assert(!MULTI || PKV_PART_WIDE != 0, "final_out lives in the wide partial store");
    if constexpr (!FUSED && PKV_PART_WIDE != 0) {   // experiment: the partial leaves as 16-byte stores of G * D / 4 threads (2: write-through, no dirty L2 lines at the end of the kernel)
        constexpr int C4w = G * D / 4;
        if (tid < C4w) {
            const int g = tid / (D / 4), d4 = tid % (D / 4);
            float mw[NW], lw[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) { const float2 t = *reinterpret_cast<const float2 *>(&sML[w][g][0]); mw[w] = t.x; lw[w] = t.y; }
            float M = mw[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, mw[w]);
            const float Mf = fmaxf(M, -3.0e38f);
            float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float l = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float  f = __expf(mw[w] - Mf);
                const float4 v = *reinterpret_cast<const float4 *>(&sO[w][g][d4 * 4]);
                o.x = fmaf(v.x, f, o.x); o.y = fmaf(v.y, f, o.y); o.z = fmaf(v.z, f, o.z); o.w = fmaf(v.w, f, o.w);
                l = fmaf(lw[w], f, l);
            }
            const size_t pi = ((size_t) s * a.n_head + hk * G + g) * a.n_splits + split;
            bool final_out = false;
            if constexpr (MULTI) final_out = a.final_out != 0;
            if (final_out) {            // the only partial of this head: what attn_combine_kernel would make of it (weight exp(0) = 1, acc / l), same bits
                *reinterpret_cast<float4 *>(&a.out[((size_t) s * a.n_head + hk * G + g) * D + d4 * 4]) = make_float4(o.x / l, o.y / l, o.z / l, o.w / l);
            } else
            if (PKV_PART_WIDE == 2) {
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.part_o + pi * D, 0, D * 4, 0x00020000);
                u4_t ov; ov.x = __float_as_uint(o.x); ov.y = __float_as_uint(o.y); ov.z = __float_as_uint(o.z); ov.w = __float_as_uint(o.w);
                __builtin_amdgcn_raw_buffer_store_b128(ov, ro, d4 * 16, 0, 16);
                if (d4 == 0) {
                    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.part_ml + pi * 2, 0, 8, 0x00020000);
                    u2_t mv; mv.x = __float_as_uint(M); mv.y = __float_as_uint(l);
                    __builtin_amdgcn_raw_buffer_store_b64(mv, rm, 0, 0, 16);
                }
            } else {
                *reinterpret_cast<float4 *>(&a.part_o[pi * D + d4 * 4]) = o;
                if (d4 == 0) *reinterpret_cast<float2 *>(&a.part_ml[pi * 2]) = make_float2(M, l);
            }
        }
    } else
    if constexpr (!FUSED) {
        static_assert((G * D) % NT == 0 || G * D < NT, "one output element per thread and pass");
        for (int e = tid; e < G * D; e += NT) {
            const int g = e / D, d = e % D;
            float mw[NW], lw[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) { const float2 t = *reinterpret_cast<const float2 *>(&sML[w][g][0]); mw[w] = t.x; lw[w] = t.y; }
            float M = mw[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, mw[w]);
            float o = 0.0f, l = 0.0f;
#if PKV_WGMAX
#pragma unroll
            for (int w = 0; w < NW; ++w) { o += sO[w][g][d]; l += lw[w]; }   // every wave weighed against the workgroup's max already
#else
            const float Mf = fmaxf(M, -3.0e38f);               // all waves empty: exp(-inf - Mf) = 0, never -inf - -inf
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = __expf(mw[w] - Mf);
                o = fmaf(sO[w][g][d], f, o);
                l = fmaf(lw[w], f, l);
            }
#endif
            const size_t pi = ((size_t) s * a.n_head + hk * G + g) * a.n_splits + split;
            a.part_o[pi * D + d] = o;
            if (d == 0) *reinterpret_cast<float2 *>(&a.part_ml[pi * 2]) = make_float2(M, l);
        }
    }
    // FUSED: the partials are handed to the last-arriving workgroup of this (sequence, kv head) inside the launch.  The
    // hand-off is placement-independent (cdna_hip_programming.md §6 Guideline 16, the sc1 form): every payload byte leaves
    // as a 16-/8-byte write-through store (buffer_store ... sc1), every storing wave drains, the workgroup's barrier, ONE
    // lane takes an agent-scope ticket; the last arriver reads every payload byte with sc1 loads (no fences at all).
    constexpr int C4 = G * D / 4;                              // float4 columns of one (m, l, O[G][D]) partial
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_o, rs_ml;
    [[maybe_unused]] unsigned gb = 0;                          // partial index of (s, hk * G, split 0)
    if constexpr (FUSED) {
        static_assert(NT % C4 == 0 && C4 <= NT, "column/part mapping of the in-kernel merge");
        rs_o  = __builtin_amdgcn_make_buffer_rsrc(a.part_o, 0, a.part_o_bytes, 0x00020000);
        rs_ml = __builtin_amdgcn_make_buffer_rsrc(a.part_ml, 0, a.part_ml_bytes, 0x00020000);
        gb = ((unsigned) s * a.n_head + hk * G) * a.n_splits;
        if (tid < C4) {
            const int g = tid / (D / 4), d4 = tid % (D / 4);
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < NW; ++w) M = fmaxf(M, sML[w][g][0]);
            float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float l = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float  f = sML[w][g][0] == -INFINITY ? 0.0f : __expf(sML[w][g][0] - M);
                const float4 v = *reinterpret_cast<const float4 *>(&sO[w][g][d4 * 4]);
                o.x = fmaf(v.x, f, o.x); o.y = fmaf(v.y, f, o.y); o.z = fmaf(v.z, f, o.z); o.w = fmaf(v.w, f, o.w);
                l = fmaf(sML[w][g][1], f, l);
            }
            const unsigned pi = gb + g * a.n_splits + split;
            u4_t ov; ov.x = __float_as_uint(o.x); ov.y = __float_as_uint(o.y); ov.z = __float_as_uint(o.z); ov.w = __float_as_uint(o.w);
            __builtin_amdgcn_raw_buffer_store_b128(ov, rs_o, (int) ((pi * D + d4 * 4) * 4), 0, 16);      // aux 16 = sc1
            if (d4 == 0) {
                u2_t mv; mv.x = __float_as_uint(M); mv.y = __float_as_uint(l);
                __builtin_amdgcn_raw_buffer_store_b64(mv, rs_ml, (int) (pi * 2 * 4), 0, 16);
            }
        }
    }
    if constexpr (PREP) {
        if (is_last) {   // the new token's rows reach the pool for the steps to come (nothing in this launch reads them back)
#pragma unroll
            for (int k = 0; k < NJW; ++k) {
                const int job = wv + k * NW;
                if (job == G)     tile_write<KT, false>(const_cast<char *>(ktile), boff + n_valid - 1, lane, curq[k], a.block_size, D);
                if (job == G + 1) tile_write<VT, true>(const_cast<char *>(vtile), boff + n_valid - 1, lane, curq[k], a.block_size, D);
            }
        }
    }
    if constexpr (STAMP != 0) {
        ts.at(10);                      // cross-wave merge done, partial stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts.real(11);                    // stores acknowledged: the wave can end
        if (lane == 0) {
            unsigned long long * dst = a.stamps + (((size_t) blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NW * NSTAMP) + (size_t) wv * NSTAMP;
#pragma unroll
            for (int i = 0; i < NSTAMP; ++i) dst[i] = ts.t[i];
        }
    }
    if constexpr (FUSED) {
        constexpr int P = NT / C4, CH = 32, JP = (CH + P - 1) / P;   // parts that share a column, splits per pass / per part
        __shared__ unsigned sLast;
        __shared__ float2   sMl2[G][CH];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores ...
        __syncthreads();                                       // ... before ONE lane signals for the workgroup
        const unsigned n_act = (unsigned) (seq_len + WG_TOK - 1) / WG_TOK;   // workgroups that arrive for this (s, hk)
        if (tid == 0) {
            // wrapping ticket: 0, 1, ..., n_act-1, then back to 0 by itself -> no per-launch reset of the counter
            const unsigned old = atomicInc(a.counters + (size_t) s * a.n_head_kv + hk, n_act - 1);
            sLast = old == n_act - 1;
        }
        __syncthreads();
        if (!sLast) return;
        // last arriver: online merge of the n_act partials, CH splits per pass; thread = (column c, part)
        const int ns = (int) n_act;
        const int c = tid % C4, part = tid / C4, g = c / (D / 4), d4 = c % (D / 4);
        float  M = -INFINITY, L = 0.0f;
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int j0 = 0; j0 < ns; j0 += CH) {
            const int n = min(CH, ns - j0);
            u2_t mlv; mlv.x = 0; mlv.y = 0;
            if (tid < G * CH)
                mlv = __builtin_amdgcn_raw_buffer_load_b64(rs_ml, (int) ((gb + (tid / CH) * a.n_splits + j0 + min(tid % CH, n - 1)) * 2 * 4), 0, 16);
            u4_t ov[JP];
#pragma unroll
            for (int k = 0; k < JP; ++k)
                ov[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, (int) (((gb + g * a.n_splits + j0 + min(part + k * P, n - 1)) * D + d4 * 4) * 4), 0, 16);
            if (tid < G * CH)
                sMl2[tid / CH][tid % CH] = tid % CH < n ? make_float2(__uint_as_float(mlv.x), __uint_as_float(mlv.y)) : make_float2(-INFINITY, 0.0f);
            __syncthreads();
            float mc = -INFINITY;
#pragma unroll
            for (int j = 0; j < CH; ++j) mc = fmaxf(mc, sMl2[g][j].x);
            const float Mn = fmaxf(M, mc);
            const float sc = M == -INFINITY ? 0.0f : __expf(M - Mn);
            acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
            L *= sc;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float2 v = sMl2[g][j];
                L = fmaf(v.y, v.x == -INFINITY ? 0.0f : __expf(v.x - Mn), L);
            }
#pragma unroll
            for (int k = 0; k < JP; ++k) {
                const int j = part + k * P;
                if (j < n) {
                    const float mj = sMl2[g][j].x;
                    const float w  = mj == -INFINITY ? 0.0f : __expf(mj - Mn);
                    acc.x = fmaf(__uint_as_float(ov[k].x), w, acc.x); acc.y = fmaf(__uint_as_float(ov[k].y), w, acc.y);
                    acc.z = fmaf(__uint_as_float(ov[k].z), w, acc.z); acc.w = fmaf(__uint_as_float(ov[k].w), w, acc.w);
                }
            }
            M = Mn;
            __syncthreads();                                   // sMl2 is rewritten by the next pass
        }
        __shared__ float4 red[NT];
        red[part * C4 + c] = acc;
        __syncthreads();
        if (tid < C4) {
            float4 r = red[c];
#pragma unroll
            for (int q = 1; q < P; ++q) { const float4 v = red[q * C4 + c]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
            const float inv = 1.0f / L;                        // all cells masked -> inf * 0 = NaN, like the reference's 1/S with S == 0
            *reinterpret_cast<float4 *>(&a.out[((size_t) s * a.n_head + hk * G + g) * D + d4 * 4]) = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
        }
    }
}

// merge the per-block partials: out = sum_j O_j e^{m_j - M} / sum_j l_j e^{m_j - M}.
// ONE WAVE per (sequence, query head), lane = D / 64 adjacent output columns.  The kernel is pure latency (a few KB per wave), so every load
// it will ever need — this lane's columns of up to CH partials AND the (m, l) pairs — is issued before the first dependent instruction: one
// memory round trip instead of two.  r03: no LDS and no barrier any more (r02: D threads per head, one 4-byte column each, three barriers, the
// weights handed round through LDS; 4.6 us per launch against a 1.8 us empty launch): lane j < 32 owns split j of a pass exactly as thread j
// did, the maximum and the sum of the weights are wave reductions, weight u reaches every lane as an SGPR (v_readlane) — the same additions in
// the same order as before, bit for bit, with half the load instructions (D = 128: 8-byte loads) and a quarter of the waves.
// IMG (r03, pkv_attn_args::out_image): the workgroup is the 256 / D waves whose heads make up one 256-element block of the token's row, and the row ALSO leaves
// as the q8_K activation image of pkv_gemv.hip (quantize_row_q8_K_ref as quant_block_q8k states it: iscale = -127 / amax over the block, q = rint(iscale x),
// d = 1 / iscale, sums of q per 32) — the bytes pkv_quantize_act would make of `out`, so the layer's wo product starts from integers.
template <int D, bool IMG = false>
__global__ void __launch_bounds__(IMG ? 256 / D * 64 : 64) attn_combine_kernel(const DecodeArgs a) {
    constexpr int CH = 32;  // splits per pass (ctx 4096 = one pass)
    constexpr int C = D / 64;                                   // columns per lane
    static_assert(D % 64 == 0 && (C == 1 || C == 2 || C == 4), "head size");
    typedef float colv __attribute__((ext_vector_type(C)));
    constexpr int HPB = IMG ? 256 / D : 1;                      // heads (= waves) per workgroup
    const int lane = threadIdx.x & 63, wv = (int) (threadIdx.x >> 6);
    const int h = blockIdx.x * HPB + wv, s = blockIdx.y;
    const int seq_len = a.seq_lens[s];
    const float * ml = a.part_ml + ((size_t) s * a.n_head + h) * a.n_splits * 2;
    const float * po = a.part_o + ((size_t) s * a.n_head + h) * a.n_splits * D + C * lane;
    // The first pass of columns and of (m, l) pairs is requested BEFORE seq_len is back: the addresses only need the
    // host-known split count (slots past the sequence's own count hold stale partials; they are selected away below,
    // never multiplied), so the sequence length and the data share one memory round trip.
    const int nsm = a.n_splits - 1;
    colv pv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) pv[u] = *reinterpret_cast<const colv *>(po + (size_t) min(u, nsm) * D);
    const float2 ml0 = *reinterpret_cast<const float2 *>(ml + (size_t) min(lane, nsm) * 2);
    const int ns = ((seq_len + WG_TOK - 1) / WG_TOK + a.n_tile - 1) / a.n_tile;   // partials the decode launch wrote for this sequence
    float * o = a.out + ((size_t) s * a.n_head + h) * D + C * lane;
    colv r;
    if (ns <= 0) { for (int c = 0; c < C; ++c) r[c] = 0.0f; *reinterpret_cast<colv *>(o) = r; if constexpr (!IMG) return; }
    else {
    float m = lane < ns ? ml0.x : -INFINITY;
    for (int j = lane + 64; j < ns; j += 64) m = fmaxf(m, ml[j * 2]);
    const float M = wave_max(m);
    float L = 0.0f;
    colv acc;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (int j0 = 0; j0 < ns; j0 += CH) {
        const int n = min(CH, ns - j0);
        float2 v = ml0;
        if (j0 > 0) {
#pragma unroll
            for (int u = 0; u < CH; ++u) pv[u] = *reinterpret_cast<const colv *>(po + (size_t) (j0 + min(u, n - 1)) * D);
            if (lane < n) v = *reinterpret_cast<const float2 *>(ml + (size_t) (j0 + lane) * 2);
        }
        float w = 0.0f;
        if (lane < n) {
            w = v.x == -INFINITY ? 0.0f : __expf(v.x - M);
            L = fmaf(v.y, w, L);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const float wu = __uint_as_float((unsigned) __builtin_amdgcn_readlane((int) __float_as_uint(w), u));   // lanes >= n hold 0
            if (u < n) {
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(pv[u][c], wu, acc[c]);
            }
        }
    }
    const float Ls = wave_sum(L);
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = acc[c] / Ls;  // all cells masked -> 0/0 = NaN, like the reference's 1/S with S == 0
    *reinterpret_cast<colv *>(o) = r;
    }
    if constexpr (IMG) {
        __shared__ float sAm[HPB];
        float am = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) am = fmaxf(am, fabsf(r[c]));
        am = wave_max(am);
        if (lane == 0) sAm[wv] = am;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < HPB; ++w) am = fmaxf(am, sAm[w]);
        const float iscale = am != 0.0f ? -127.0f / am : 0.0f;
        char * img = a.out_image + (size_t) s * a.image_bytes;
        const int K = a.n_head * D, e0 = h * D + C * lane;       // this lane's first element of the row
        int q[C], qs = 0;
        uint32_t pk = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) { q[c] = (int) __builtin_rintf(iscale * r[c]); qs += q[c]; pk |= (uint32_t) (q[c] & 0xFF) << (8 * c); }
        if constexpr (C == 1) reinterpret_cast<uint8_t *>(img)[e0] = (uint8_t) pk;
        else if constexpr (C == 2) *reinterpret_cast<uint16_t *>(img + e0) = (uint16_t) pk;
        else *reinterpret_cast<uint32_t *>(img + e0) = pk;
        // sums per 32 elements = 32 / C lanes: DPP inside the row of 16, one permlane swap more where 32 lanes share a sum
        qs += __builtin_amdgcn_update_dpp(0, qs, 0xB1, 0xf, 0xf, true);
        qs += __builtin_amdgcn_update_dpp(0, qs, 0x4E, 0xf, 0xf, true);
        qs += __builtin_amdgcn_update_dpp(0, qs, 0x141, 0xf, 0xf, true);
        if constexpr (C <= 2) qs += __builtin_amdgcn_update_dpp(0, qs, 0x140, 0xf, 0xf, true);          // 16 lanes
        if constexpr (C == 1) { auto t = __builtin_amdgcn_permlane16_swap((unsigned) qs, (unsigned) qs, false, false); qs = (int) t[0] + (int) t[1]; }   // 32 lanes
        constexpr int LPS = 32 / C;                               // lanes per sum
        if (lane % LPS == 0) reinterpret_cast<int *>(img + K + (K >> 8) * 4)[e0 >> 5] = qs;
        if (threadIdx.x == 0) reinterpret_cast<float *>(img + K)[blockIdx.x] = am != 0.0f ? 1.0f / iscale : 0.0f;
    }
}

// measurement hook (pkv_attn_decode_profiled): when set, the split-KV kernel is launched with start/stop events that
// carry the dispatch's own GPU timestamps (what rocprofv3 --kernel-trace reports), not host-side event records
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
thread_local unsigned long long * g_stamps = nullptr;
thread_local int g_stamp_mode = 1;   // pkv_attn_decode_stamps: where the diagnostic instantiation writes

template <typename K>
void launch_kernel(K kernel, dim3 grid, dim3 block, hipStream_t st, const DecodeArgs & da) {
    if (g_prof_start) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, g_prof_start, g_prof_stop, 0, da.block_table, da.seq_lens, da.max_blocks, da.bs_shift, da);
    else              hipLaunchKernelGGL(kernel, grid, block, 0, st, da.block_table, da.seq_lens, da.max_blocks, da.bs_shift, da);
}

// (decode_tiles_per_wg: pkv_common.h — host logic, shared with the C ABI's pkv_attn_decode_tiles and the sanitizer build)

template <int D>
void launch_combine(const DecodeArgs & da, int n_seq, hipStream_t st) {
    if constexpr (256 % D == 0) {
        if (da.out_image) { hipLaunchKernelGGL((attn_combine_kernel<D, true>), dim3(da.n_head * D / 256, n_seq), dim3(256 / D * 64), 0, st, da); return; }
    }
    hipLaunchKernelGGL((attn_combine_kernel<D>), dim3(da.n_head, n_seq), dim3(64), 0, st, da);
}

template <int KT, int VT, int D, int G>
int launch_one(const DecodeArgs & da0, int n_seq, hipStream_t st) {
    DecodeArgs da = da0;
    const bool cap = da.softcap != 0.0f, sp = da.slot_pos != nullptr || da.mask != nullptr;
    // multi-tile form: the plain variants with their own merge launch
    da.n_tile = 1; da.final_out = 0;
    if (!cap && !sp && !da.counters && !da.stamps) {
        da.n_tile    = decode_tiles_per_wg(da.n_splits, da.n_head_kv, n_seq, G);
        da.n_splits  = (da.n_splits + da.n_tile - 1) / da.n_tile;
        da.final_out = da.n_tile > 1 && da.n_splits == 1 && da.out_direct != nullptr && da.out_image == nullptr;   // (the image is made by the merge launch)
        if (da.final_out) da.out = da.out_direct;     // (also under PKV_ATTN_NO_COMBINE: this form has no partials to leave behind)
    }
    // quantized KV with workgroups to spare: 32 tokens per wave (4-wave workgroups)
    constexpr bool QKV = KT != PKV_TYPE_F16 || VT != PKV_TYPE_F16;
    const bool wide = QKV && (long) da.n_splits * da.n_head_kv * n_seq >= 1024;
#ifndef PKV_TPW_Q4
#define PKV_TPW_Q4 32
#endif
    constexpr int TQ = (KT == PKV_TYPE_Q4_0 && VT == PKV_TYPE_Q4_0) ? PKV_TPW_Q4 : 32;   // tokens per wave of the wide form
    const bool multi = da.n_tile > 1;
    dim3 grid(da.n_splits, da.n_head_kv, n_seq);
    const dim3 block(64 * Geo<KT, VT, D>::NW);
    const dim3 block32(64 * Geo<KT, VT, D, TQ>::NW);
    if (da.k_cur && da.counters) {   // the whole decode step of a layer in ONE launch
        launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, true, true>, grid, block, st, da);
    } else if (da.k_cur) {      // fused decode step: RoPE(Q), RoPE(K) + KV store and the split-KV attention in one launch
        if constexpr (KT == PKV_TYPE_F16 && VT == PKV_TYPE_F16 && D == 128 && (G == 4 || G == 8)) {
            if (da.stamps) {    // diagnostic instantiations (pkv_attn_decode_stamps)
                if (g_stamp_mode == 2) launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, true, 2>, grid, block, st, da);
                else                   launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, true, 1>, grid, block, st, da);
                PKV_HIP(hipGetLastError());
                return PKV_OK;
            }
        }
        if (da.stamps) { set_error("pkv_attn_decode_stamps: f16 KV, head_dim 128, gqa 4 or 8 only"); return PKV_ERR_UNSUPPORTED; }
        if constexpr (QKV) {
            if (wide && multi)  launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, true, 0, TQ, true>, grid, block32, st, da);
            else if (wide)      launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, true, 0, TQ>, grid, block32, st, da);
        }
        if (!wide && multi) launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, true, 0, 16, true>, grid, block, st, da);
        else if (!wide)     launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, true>, grid, block, st, da);
        if (da.out && !da.final_out) launch_combine<D>(da, n_seq, st);
    } else if (da.counters) {   // fused merge (no second launch); softcap / irregular sequences take the two-launch path
        launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, true, false>, grid, block, st, da);
    } else {
        const bool plain = !cap && !sp;
        if constexpr (QKV) {
            if (wide && plain && multi) launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, false, 0, TQ, true>, grid, block32, st, da);
            else if (wide && plain)     launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, false, 0, TQ>, grid, block32, st, da);
        }
        if (wide && plain) {}
        else if (plain && multi) launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, false, 0, 16, true>, grid, block, st, da);
        else if (plain)       launch_kernel(attn_decode_kernel<KT, VT, D, G, false, false, false, false>, grid, block, st, da);
        else if (!cap && sp)  launch_kernel(attn_decode_kernel<KT, VT, D, G, false, true, false, false>, grid, block, st, da);
        else if (cap && !sp)  launch_kernel(attn_decode_kernel<KT, VT, D, G, true, false, false, false>, grid, block, st, da);
        else                  launch_kernel(attn_decode_kernel<KT, VT, D, G, true, true, false, false>, grid, block, st, da);
        if (da.out && !da.final_out) launch_combine<D>(da, n_seq, st);
    }
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}

template <int KT, int VT, int D>
int launch_g(const DecodeArgs & da, int G, int n_seq, hipStream_t st) {
    switch (G) {
        case 1: return launch_one<KT, VT, D, 1>(da, n_seq, st);
        case 2: return launch_one<KT, VT, D, 2>(da, n_seq, st);
        case 4: return launch_one<KT, VT, D, 4>(da, n_seq, st);
        case 8: return launch_one<KT, VT, D, 8>(da, n_seq, st);
    }
    set_error("pkv_attn_decode: unsupported gqa ratio %d (supported 1,2,4,8)", G);
    return PKV_ERR_UNSUPPORTED;
}

template <int KT, int VT>
int launch_d(const DecodeArgs & da, int D, int G, int n_seq, hipStream_t st) {
    switch (D) {
        case 64:  return launch_g<KT, VT, 64>(da, G, n_seq, st);
        case 128: return launch_g<KT, VT, 128>(da, G, n_seq, st);
    }
    set_error("pkv_attn_decode: unsupported head_dim %d (supported 64,128)", D);
    return PKV_ERR_UNSUPPORTED;
}

template <int KT>
int launch_v(const DecodeArgs & da, int VT, int D, int G, int n_seq, hipStream_t st) {
    switch (VT) {
        case PKV_TYPE_F16:  return launch_d<KT, PKV_TYPE_F16>(da, D, G, n_seq, st);
        case PKV_TYPE_Q8_0: return launch_d<KT, PKV_TYPE_Q8_0>(da, D, G, n_seq, st);
        case PKV_TYPE_Q4_0: return launch_d<KT, PKV_TYPE_Q4_0>(da, D, G, n_seq, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

}  // namespace

// [part_o f32 n_tok*n_head*ns*D][part_ml f32 n_tok*n_head*ns*2][tickets u32 n_tok*n_head][rope cache f32 n_tok*D, 16-byte aligned]
size_t decode_workspace_bytes(int n_tok, int n_head, int D, int max_seq_len) {
    const size_t ns = (size_t) (max_seq_len + WG_TOK - 1) / WG_TOK;
    return (size_t) n_tok * n_head * (ns ? ns : 1) * (D + 2) * sizeof(float) + (size_t) n_tok * n_head * sizeof(unsigned)
         + 16 + (size_t) n_tok * D * sizeof(float);   // + the step's RoPE cache rows (pkv_attn_decode_step without rope_cache)
}

int launch_attn_decode(const pkv_attn_args & a, hipStream_t st, const pkv_step_args * step) {
    PKV_REQUIRE(a.q && (a.out || (a.flags & PKV_ATTN_NO_COMBINE)) && a.k_pool && a.v_pool && a.block_table && a.seq_lens, "pkv_attn_decode: null pointer");
    PKV_REQUIRE(a.q_start == nullptr && a.n_tok == a.n_seq, "pkv_attn_decode: one query token per sequence (use pkv_attn_prefill)");
    PKV_REQUIRE(a.n_seq > 0 && a.n_head > 0 && a.n_head_kv > 0 && a.n_head % a.n_head_kv == 0, "pkv_attn_decode: bad head counts");
    PKV_REQUIRE(a.block_size >= WG_TOK && (a.block_size & (a.block_size - 1)) == 0, "pkv_attn_decode: block_size must be a power of two >= %d", WG_TOK);
    PKV_REQUIRE(kv_type_ok(a.type_k) && kv_type_ok(a.type_v), "pkv_attn_decode: KV types must be f16/q8_0/q4_0");
    PKV_REQUIRE(a.max_seq_len > 0 && a.max_seq_len <= a.max_blocks * a.block_size, "pkv_attn_decode: max_seq_len out of range");
    PKV_REQUIRE((a.q_stride_head % 4) == 0 && (a.q_stride_tok % 4) == 0 && ((uintptr_t) a.q % 16) == 0, "pkv_attn_decode: q must be 16-byte aligned");
    PKV_REQUIRE(a.slot_pos == nullptr || a.q_pos != nullptr, "pkv_attn_decode: slot_pos needs q_pos");
    if (a.flags & (PKV_ATTN_Q_F16 | PKV_ATTN_OUT_F16)) { set_error("pkv_attn_decode: f16 q / out are a pkv_attn_prefill mode (a decode step moves 2 rows per head)"); return PKV_ERR_UNSUPPORTED; }
    if (a.max_bias != 0.0f || (a.flags & PKV_ATTN_NONCAUSAL) || (a.mask && (a.slot_pos || step))) {
        set_error("pkv_attn_decode: ALiBi / non-causal go through pkv_attn_prefill; a mask excludes slot_pos and the fused step");
        return PKV_ERR_UNSUPPORTED;
    }
    PKV_REQUIRE(!a.mask || a.mask_stride >= a.max_seq_len, "pkv_attn_decode: mask rows shorter than max_seq_len");
    const size_t need = decode_workspace_bytes(a.n_tok, a.n_head, a.head_dim, a.max_seq_len);
    PKV_REQUIRE(a.workspace && a.workspace_bytes >= need, "pkv_attn_decode: workspace too small (%zu < %zu)", a.workspace_bytes, need);

    // measurement hook: events bound to the split-KV launch itself (see pkv_attn_args.ev_start)
    struct ProfScope {
        ProfScope(void * e0, void * e1) { g_prof_start = (hipEvent_t) e0; g_prof_stop = (hipEvent_t) e1; }
        ~ProfScope() { g_prof_start = g_prof_stop = nullptr; }
    } prof_scope(a.ev_start && a.ev_stop ? a.ev_start : nullptr, a.ev_start && a.ev_stop ? a.ev_stop : nullptr);

    DecodeArgs da;
    da.q = a.q; da.q_stride_tok = a.q_stride_tok; da.q_stride_head = a.q_stride_head;
    da.k_pool = (const char *) a.k_pool; da.v_pool = (const char *) a.v_pool;
    da.block_table = a.block_table; da.seq_lens = a.seq_lens; da.slot_pos = a.slot_pos; da.q_pos = a.q_pos;
    da.mask = (const _Float16 *) a.mask; da.mask_stride = a.mask_stride;
    da.n_head = a.n_head; da.n_head_kv = a.n_head_kv; da.block_size = a.block_size; da.max_blocks = a.max_blocks;
    da.kv_div = 1; da.n_head_kv_pool = a.n_head_kv;
    {   // gqa ratios 16 and 32 (the reference's harness tests nr = 16): ratio 8 over kv_div times as many virtual kv heads — the K / V tiles are read kv_div times
        const int G0 = a.n_head / a.n_head_kv;
        if (G0 == 16 || G0 == 32) {
            if (step) { set_error("pkv_attn_decode_step: gqa ratio %d takes pkv_qkv_prepare + pkv_attn_decode", G0); return PKV_ERR_UNSUPPORTED; }
            da.kv_div = G0 / 8; da.n_head_kv = a.n_head_kv * da.kv_div;
        }
    }
    da.n_splits = (a.max_seq_len + WG_TOK - 1) / WG_TOK;
    da.bs_shift = 0;
    while ((WG_TOK << da.bs_shift) < a.block_size) ++da.bs_shift;
    da.part_o  = (float *) a.workspace;
    da.part_ml = da.part_o + (size_t) a.n_tok * a.n_head * da.n_splits * a.head_dim;
    da.out = (a.flags & PKV_ATTN_NO_COMBINE) ? nullptr : a.out;
    da.out_direct = a.out;
    da.out_image = (char *) a.out_image; da.image_bytes = 0;
    if (a.out_image) {
        const int64_t Krow = (int64_t) a.n_head * a.head_dim;
        PKV_REQUIRE(256 % a.head_dim == 0 && Krow % 256 == 0 && !(a.flags & (PKV_ATTN_NO_COMBINE | PKV_ATTN_FUSED_COMBINE)),
                    "pkv_attn_decode: out_image needs 256 %% head_dim == 0, whole 256-blocks per row and the merge launch");
        da.image_bytes = (int) act_image_bytes(Krow);
    }
    da.counters = nullptr;
    const size_t po_bytes = (size_t) a.n_tok * a.n_head * da.n_splits * a.head_dim * sizeof(float);
    da.part_o_bytes = (int) po_bytes; da.part_ml_bytes = (int) (po_bytes / a.head_dim * 2);
    if ((a.flags & PKV_ATTN_FUSED_COMBINE) && !(a.flags & PKV_ATTN_NO_COMBINE) && a.logit_softcap == 0.0f && !a.slot_pos && !a.mask &&
        po_bytes < ((size_t) 1 << 31))   // 32-bit buffer offsets; larger slabs take the two-launch path
        da.counters = reinterpret_cast<unsigned *>(da.part_ml + (size_t) a.n_tok * a.n_head * da.n_splits * 2);
    da.softcap = a.logit_softcap;
    da.scale   = a.logit_softcap != 0.0f ? a.scale / a.logit_softcap : a.scale;  // :12290-12292
    da.k_cur = da.v_cur = nullptr; da.rope_cache = nullptr; da.has_rope = 0; da.rope = RopeDev{};
    da.stamps = g_stamps;
    da.n_tile = 1; da.final_out = 0;
    if (step) {
        PKV_REQUIRE(step->k_cur && step->v_cur, "pkv_attn_decode_step: null pointer");
        PKV_REQUIRE(((uintptr_t) step->k_cur % 8) == 0 && ((uintptr_t) step->v_cur % 8) == 0 && (a.q_stride_head % 2) == 0, "pkv_attn_decode_step: rows must be 8-byte aligned");
        if (a.logit_softcap != 0.0f || a.slot_pos) {
            set_error("pkv_attn_decode_step: softcap / irregular sequences take pkv_qkv_prepare + pkv_attn_decode");
            return PKV_ERR_UNSUPPORTED;
        }
        da.k_cur = step->k_cur; da.v_cur = step->v_cur;
        if (step->rope) {
            const int rc = make_rope(step->rope, a.head_dim, &da.rope);
            if (rc != PKV_OK) return rc;
            if ((da.rope.mode & 2) && da.rope.n_dims != a.head_dim) { set_error("fused NEOX rope needs n_dims == head_dim"); return PKV_ERR_UNSUPPORTED; }
            da.has_rope = 1;
            if (step->rope_cache) {
                PKV_REQUIRE(((uintptr_t) step->rope_cache % 16) == 0, "pkv_attn_decode_step: rope_cache must be 16-byte aligned");
                da.rope_cache = reinterpret_cast<const float4 *>(step->rope_cache);
            } else {   // no row supplied by the caller: build this launch's rows in the workspace tail first
                PKV_REQUIRE(step->pos != nullptr, "pkv_attn_decode_step: pos or rope_cache is required with rope");
                char * tail = reinterpret_cast<char *>(da.part_ml + (size_t) a.n_tok * a.n_head * da.n_splits * 2) + (size_t) a.n_tok * a.n_head * sizeof(unsigned);
                tail = reinterpret_cast<char *>(((uintptr_t) tail + 15) & ~(uintptr_t) 15);
                const int rc2 = launch_rope_cache(step->pos, reinterpret_cast<float *>(tail), a.n_tok, a.head_dim, da.rope, st);
                if (rc2 != PKV_OK) return rc2;
                da.rope_cache = reinterpret_cast<const float4 *>(tail);
            }
        }
    }
    const int G = a.n_head / da.n_head_kv;
    switch (a.type_k) {
        case PKV_TYPE_F16:  return launch_v<PKV_TYPE_F16>(da, a.type_v, a.head_dim, G, a.n_seq, st);
        case PKV_TYPE_Q8_0: return launch_v<PKV_TYPE_Q8_0>(da, a.type_v, a.head_dim, G, a.n_seq, st);
        case PKV_TYPE_Q4_0: return launch_v<PKV_TYPE_Q4_0>(da, a.type_v, a.head_dim, G, a.n_seq, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

int launch_attn_decode_stamps(const pkv_attn_args & a, const pkv_step_args * step, unsigned long long * stamps, int mode, hipStream_t st) {
    PKV_REQUIRE(stamps && step, "pkv_attn_decode_stamps: stamps buffer and step arguments are required");
    g_stamps = stamps;
    g_stamp_mode = mode == 2 ? 2 : 1;
    const int rc = launch_attn_decode(a, st, step);
    g_stamps = nullptr;
    return rc;
}

}  // namespace pkv
