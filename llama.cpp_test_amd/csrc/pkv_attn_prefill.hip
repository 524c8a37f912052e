// pkv_attn_prefill.hip — batched-prefill paged attention for gfx950: flash-style, MFMA (v_mfma_f32_32x32x16_f16).
//
// Replaces, for many query tokens per sequence, the K/V gather views + GGML_OP_FLASH_ATTN_EXT + dense causal KQ_mask
// of the reference (/ref/src/llama-graph.cpp:1404-1432, :403-468; CPU kernel /ref/ggml/src/ggml-cpu/ggml-cpu.c:12219).
//
// Decomposition: one workgroup per (sequence, kv head, 32-query tile); wave g of the workgroup owns query head
// hk*G + g, so the K/V tile staged in LDS is read from HBM once for all G = n_head/n_head_kv query heads.
// Per 64-token KV tile (half a 128-token block — tiles never straddle blocks, the block table is consulted per tile):
//   S^T[kv][q] = K[kv][:] . Q^T       A = K rows from LDS (XOR-swizzled 16-B chunks, conflict-free ds_read_b128),
//                                     B = Q^T fragments held in registers for the whole kernel;
//                                     the accumulator has the QUERY on the lane, so the online-softmax state (m, l)
//                                     is per lane and the row max/sum needs one cross-half exchange only
//   O^T[d][q] += V^T[d][kv] . P^T     B = P^T straight from the S^T accumulators (cdna_hip_programming.md §3
//                                     "An accumulator tile as the next MFMA's operand", permuted k order),
//                                     A = V^T from the token-pair-interleaved V tile ([pair][d] dwords: the two halves of
//                                     a dword are two consecutive tokens of one d -> conflict-free ds_read_b32)
// Causal structure replaces the mask: KV tiles past the tile's last query position are never visited; the diagonal tile
// masks per element.  Quantized K/V (q8_0, q4_0) are dequantized to f16 while staging (Q stays f16: closer to exact
// arithmetic than the reference's q8_0-quantized Q; tolerance in tests/test_gpu_prefill.py).
#include "pkv_common.h"

#include <algorithm>

namespace pkv {
extern void * g_prefill_stamps;
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __fp16 half2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int QT  = 32;  // query rows per workgroup tile (one MFMA N dimension)
constexpr int KVT = 64;  // KV tokens per LDS tile

struct PrefillArgs {
    const float *   q;
    int64_t         q_stride_tok, q_stride_head;
    float *         out;
    const char *    k_pool;
    const char *    v_pool;
    const int32_t * block_table;
    const int32_t * seq_lens;
    const int32_t * q_start;
    int   n_head, n_head_kv, block_size, max_blocks;
    float scale, softcap;
    const _Float16 * mask;        // optional additive mask f16 [>= n_tok rows][mask_stride], row = token index, col = kv cell
    int64_t mask_stride;
    float   max_bias;             // ALiBi: slope(head) * mask (/ref/ggml/src/ggml-cpu/ggml-cpu.c:12294-12316)
    int     causal;               // 0: every query sees all seq_len cells (test-backend-ops style), 1: llama causal
    unsigned long long * stamps;  // diagnostic build only (PKV_PREFILL_STAMP): per-workgroup phase timestamps
    // split-KV (few query tiles over a long past, e.g. a 512-token ubatch at ctx 16384): the KV range of a query tile is cut
    // into n_split parts, one workgroup each; partial (m, l, O) go to scratch and prefill_combine_kernel merges them
    int     n_split, n_tok_total; // n_split == 1: no scratch, results go straight to `out`
    float * part_o;               // [n_tok][n_head][n_split][D], un-normalised
    float * part_ml;              // [n_tok][n_head][n_split][2] = (m * cexp (log2 domain), l)
};

__device__ __forceinline__ int crow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }  // C/D row of register i

// Staging of one 64-token K tile: HBM -> registers (load) now, registers -> LDS (write) one loop iteration later, so the
// loads of tile t+1 are in flight while tile t is being multiplied (cdna_hip_programming.md T14, async-STAGE split).
// LDS image: f16 rows of D halfs, 16-byte chunk c of row r stored at chunk c ^ (r & 15) (conflict-free ds_read_b128).
// Loads are unconditional: rows past the end of the sequence are clamped to the last valid row and removed later by
// the causal test.
template <int KT, int D, int NT>
struct KStage {
    static constexpr int CPR = D / 8;            // 16-byte chunks per f16 row
    static constexpr int NI  = KVT * CPR / NT;   // chunks per thread
    static_assert(KVT * CPR % NT == 0, "tile must divide evenly over the workgroup");
    uint2 raw[NI][2];
    float dsc[NI];

    __device__ __forceinline__ void load(const char * tile, int row0, int n_rows_valid, int block_size, int tid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * NT, r = c / CPR, cc = c % CPR;
            const int row = row0 + min(r, n_rows_valid - 1);
            dsc[i] = 0.0f;
            if constexpr (KT == PKV_TYPE_F16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(tile + ((size_t) row * D + cc * 8) * 2);
                raw[i][0] = make_uint2(v.x, v.y); raw[i][1] = make_uint2(v.z, v.w);
            } else {
                dsc[i] = (float) *reinterpret_cast<const _Float16 *>(
                    tile + tile_quant_bytes(KT, block_size, D) + ((size_t) row * (D / PKV_QK) + cc * 8 / PKV_QK) * 2);
                if constexpr (KT == PKV_TYPE_Q8_0) {
                    raw[i][0] = *reinterpret_cast<const uint2 *>(tile + (size_t) row * D + cc * 8);
                } else {  // q4_0: byte j of the 32-block = elem j | elem j+16 << 4
                    const int blk = cc * 8 / PKV_QK, e0 = cc * 8 % PKV_QK;
                    raw[i][0] = *reinterpret_cast<const uint2 *>(tile + (size_t) row * (D / 2) + blk * 16 + (e0 & 15));
                }
            }
        }
    }
    __device__ __forceinline__ void write(char * lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * NT, r = c / CPR, cc = c % CPR;
            uint4 v;
            if constexpr (KT == PKV_TYPE_F16) {
                v = make_uint4(raw[i][0].x, raw[i][0].y, raw[i][1].x, raw[i][1].y);
            } else {
                const uint32_t ww[2] = {raw[i][0].x, raw[i][0].y};
                _Float16 o[8];
                if constexpr (KT == PKV_TYPE_Q8_0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (_Float16) ((float) (int) (signed char) (ww[j >> 2] >> (8 * (j & 3))) * dsc[i]);
                } else {
                    const int sh = (cc * 8 % PKV_QK) < 16 ? 0 : 4;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (_Float16) ((float) ((int) ((ww[j >> 2] >> (8 * (j & 3) + sh)) & 0xF) - 8) * dsc[i]);
                }
                v = *reinterpret_cast<const uint4 *>(o);
            }
            *reinterpret_cast<uint4 *>(lds + (size_t) r * D * 2 + ((cc ^ (r & 15)) & (CPR - 1)) * 16) = v;
        }
    }
};

// V tile: LDS image [32 token pairs][D] dwords (half2 = tokens 2p, 2p+1 of one d).  Unconditional clamped loads; cells
// past the end of the sequence are zeroed by a select when written (0 * stale NaN must not happen).
template <int VT, int D, int NT>
struct VStage {
    static constexpr int CPP = D / 4;                    // 16-byte chunks per pair row (D dwords)
    static constexpr int NI  = (KVT / 2) * CPP / NT;     // chunks per thread
    static_assert((KVT / 2) * CPP % NT == 0, "tile must divide evenly over the workgroup");
    static constexpr int TL = VT == PKV_TYPE_F16 ? 2 : (VT == PKV_TYPE_Q8_0 ? 4 : 8);
    uint4 raw[NI];
    float d0[NI], d1[NI];
    int   row0_, nval_;

    __device__ __forceinline__ void load(const char * tile, int row0, int n_rows_valid, int block_size, int tid) {
        row0_ = row0; nval_ = n_rows_valid;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * NT, p = c / CPP, cc = c % CPP;
            const int row = row0 + min(2 * p, (n_rows_valid - 1) & ~1);   // even row of a pair that exists
            const int grp = row / TL, tl = row % TL;
            raw[i] = *reinterpret_cast<const uint4 *>(tile + ((size_t) grp * D + cc * 4) * 4);
            d0[i] = d1[i] = 0.0f;
            if constexpr (VT != PKV_TYPE_F16) {
                const _Float16 * sp = reinterpret_cast<const _Float16 *>(
                    tile + tile_quant_bytes(VT, block_size, D) + ((size_t) grp * (D / PKV_QK) + cc * 4 / PKV_QK) * TL * 2);
                d0[i] = (float) sp[tl];
                d1[i] = (float) sp[tl + 1];
            }
        }
    }
    __device__ __forceinline__ void write(char * lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * NT, p = c / CPP, cc = c % CPP;
            const int t0 = 2 * p;
            uint4 v = raw[i];
            if constexpr (VT != PKV_TYPE_F16) {
                const int tl = (row0_ + min(t0, (nval_ - 1) & ~1)) % TL;
                const uint32_t ww[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x, y;
                    if constexpr (VT == PKV_TYPE_Q8_0) {
                        x = (float) (int) (signed char) (ww[j] >> (8 * tl)) * d0[i];
                        y = (float) (int) (signed char) (ww[j] >> (8 * (tl + 1))) * d1[i];
                    } else {
                        x = (float) ((int) ((ww[j] >> (4 * tl)) & 0xF) - 8) * d0[i];
                        y = (float) ((int) ((ww[j] >> (4 * (tl + 1))) & 0xF) - 8) * d1[i];
                    }
                    const _Float16 hx = (_Float16) x, hy = (_Float16) y;
                    o[j] = (uint32_t) *reinterpret_cast<const uint16_t *>(&hx) | ((uint32_t) *reinterpret_cast<const uint16_t *>(&hy) << 16);
                }
                v = make_uint4(o[0], o[1], o[2], o[3]);
            }
            const uint32_t keep = t0 >= nval_ ? 0u : (t0 + 1 >= nval_ ? 0xFFFFu : 0xFFFFFFFFu);
            v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
            *reinterpret_cast<uint4 *>(lds + ((size_t) p * D + cc * 4) * 4) = v;
        }
    }
};

template <int KT, int VT, int D, int G, bool SOFTCAP, bool STAMP = false, bool MASKED = false>
__global__ void __launch_bounds__(64 * G, 2) attn_prefill_kernel(const PrefillArgs a) {
    unsigned long long t_begin = 0, t_q = 0, t_loop = 0, t_stage = 0, t_qk = 0, t_sm = 0, t_pv = 0, tt = 0;
    if constexpr (STAMP) t_begin = __builtin_amdgcn_s_memtime();
    constexpr int NT = 64 * G, KS = D / 16, DT = D / 32;  // k-steps of QK^T, 32-row d tiles of O^T
    // two LDS images per operand: tile t+1 is written while slower waves still multiply tile t -> ONE barrier per tile
    constexpr int IMG = KVT * D * 2;                                  // one f16 tile image
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];       // K0, V0, K1, V1; the epilogue stages O through all of it
    auto sKb = [&](int i) -> char * { return smem + (size_t) i * 2 * IMG; };
    auto sVb = [&](int i) -> char * { return smem + (size_t) i * 2 * IMG + IMG; };

    // Causal work grows with the query tile index (tile t visits t/2 + 1 KV tiles): dispatch order = longest first, over
    // ALL sequences and heads (workgroups are handed out in linear block-id order), so the short tiles fill the tail.
    const int lin  = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nhs = gridDim.y * gridDim.z;
    const int xr   = gridDim.x - 1 - lin / nhs;                     // x = tile * n_split + split, longest tiles first
    const int tile = xr / a.n_split, sp = xr % a.n_split, hk = (lin % nhs) % gridDim.y, s = (lin % nhs) / gridDim.y;
    const int tok0 = a.q_start[s], n_q = a.q_start[s + 1] - tok0;
    if (tile * QT >= n_q) return;
    const int seq_len = a.seq_lens[s];
    const int tid = threadIdx.x, g = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int head = hk * G + g;
    const int jr = tile * QT + r;            // this lane's query inside the sequence's new tokens
    const bool q_ok = jr < n_q;
    // index of the last KV cell this query sees, and the workgroup-uniform bound of the KV loop
    const int qpos   = a.causal ? seq_len - n_q + jr : seq_len - 1;
    const int kv_all = a.causal ? min(seq_len, seq_len - n_q + tile * QT + QT) : seq_len;   // cells this query tile can see
    const int nkt    = (kv_all + KVT - 1) / KVT;                                             // ... in KV tiles
    const int kv_beg = (int) ((long long) nkt * sp / a.n_split) * KVT;                       // this workgroup's share
    const int kv_end = min(kv_all, (int) ((long long) nkt * (sp + 1) / a.n_split) * KVT);

    floatx16 o[DT + 1];   // o[DT] = row sums (every row of that tile holds l of the lane's query)
#pragma unroll
    for (int dt = 0; dt <= DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.0f;
    float m = -INFINITY;
    const half8_t ones = {(_Float16) 1.0f, (_Float16) 1.0f, (_Float16) 1.0f, (_Float16) 1.0f,
                          (_Float16) 1.0f, (_Float16) 1.0f, (_Float16) 1.0f, (_Float16) 1.0f};
    const float cexp = ((SOFTCAP || MASKED) ? 1.0f : a.scale) * 1.44269504088896340736f;   // exp(x * scale) = exp2(x * cexp)
    const int qpos_min = a.causal ? seq_len - n_q + tile * QT : seq_len - 1;                  // smallest last-visible cell in the tile
    float slope = 1.0f;
    if constexpr (MASKED) {
        if (a.max_bias > 0.0f) {
            const uint32_t n_head_log2 = 1u << (31 - __clz(a.n_head));
            const float m0 = exp2f(-a.max_bias / n_head_log2), m1 = exp2f(-(a.max_bias / 2.0f) / n_head_log2);
            slope = (uint32_t) head < n_head_log2 ? powf(m0, head + 1) : powf(m1, 2 * (head - n_head_log2) + 1);
        }
    }
    const _Float16 * mrow = MASKED ? a.mask + (size_t) (tok0 + (q_ok ? jr : 0)) * a.mask_stride : nullptr;

    const size_t tb_k = tile_bytes(KT, a.block_size, D), tb_v = tile_bytes(VT, a.block_size, D);
    KStage<KT, D, NT> kst;
    VStage<VT, D, NT> vst;
    auto prefetch = [&](int kv0) {
        const int phys = a.block_table[(size_t) s * a.max_blocks + kv0 / a.block_size];
        const int row0 = kv0 % a.block_size, nval = min(KVT, seq_len - kv0);
        kst.load(a.k_pool + ((size_t) phys * a.n_head_kv + hk) * tb_k, row0, nval, a.block_size, tid);
        vst.load(a.v_pool + ((size_t) phys * a.n_head_kv + hk) * tb_v, row0, nval, a.block_size, tid);
    };
    // the first KV tile's loads go out BEFORE the Q loads: the block-table lookup and the tile's HBM latency overlap the
    // (row-strided, therefore slow) Q fetch instead of following it
    prefetch(min(kv_beg, max(kv_all - 1, 0) / KVT * KVT));   // (an empty share still loads a valid, aligned tile; its loop does not run)
    // ---- Q^T fragments, rounded to f16 (GGML_FP32_TO_FP16 of the reference's q_to_vec_dot for f16 K) ---------------
    half8_t qf[KS];
    {
        const float * qp = a.q + (size_t) (tok0 + (q_ok ? jr : 0)) * a.q_stride_tok + (size_t) head * a.q_stride_head;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // rows past the end of the tile read query 0 of the sequence (in bounds) and are never written back
            const float4 x0 = *reinterpret_cast<const float4 *>(qp + ks * 16 + h * 8);
            const float4 x1 = *reinterpret_cast<const float4 *>(qp + ks * 16 + h * 8 + 4);
            qf[ks] = half8_t{(_Float16) x0.x, (_Float16) x0.y, (_Float16) x0.z, (_Float16) x0.w,
                             (_Float16) x1.x, (_Float16) x1.y, (_Float16) x1.z, (_Float16) x1.w};
        }
    }

    if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_q = __builtin_amdgcn_s_memtime(); }
    kst.write(sKb(0), tid);
    vst.write(sVb(0), tid);
    for (int kv0 = kv_beg, it = 0; kv0 < kv_end; kv0 += KVT, ++it) {
        if constexpr (STAMP) tt = __builtin_amdgcn_s_memtime();
        const char * sK = sKb(it & 1);
        const char * sV = sVb(it & 1);
        const bool more = kv0 + KVT < kv_end;
        __syncthreads();                       // tile `it` is complete in LDS; everyone is done with tile it-1
        if (more) prefetch(kv0 + KVT);         // HBM -> registers, in flight while this tile is multiplied
        if constexpr (STAMP) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_stage += t2 - tt; tt = t2; }

        // ---- S^T = K . Q^T : two 32-row kv tiles ---------------------------------------------------------------
        floatx16 sacc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[mt][i] = 0.0f;
            const int row = mt * 32 + r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8_t kf = *reinterpret_cast<const half8_t *>(sK + (size_t) row * D * 2 + (((2 * ks + h) ^ (row & 15)) & (D / 8 - 1)) * 16);
                sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[mt], 0, 0, 0);
            }
        }
        if constexpr (STAMP) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory"); const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_qk += t2 - tt; tt = t2; }
        // ---- causal mask (only tiles that reach the diagonal), running max; VALU kept to ~1 op per logit:
        //      the softmax scale rides in the exponent, p = exp2(c*s - c*m) = one v_fma + one v_exp -----------------------
        if constexpr (SOFTCAP) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[mt][i] = a.softcap * tanhf(sacc[mt][i] * a.scale);
        } else if constexpr (MASKED) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[mt][i] *= a.scale;
        }
        if constexpr (MASKED) {  // s += slope * mask (:12349-12365); -inf entries skip the cell
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kv = min(kv0 + mt * 32 + crow(i, h), seq_len - 1);
                    sacc[mt][i] += slope * (float) mrow[kv];
                }
        }
        if (kv0 + KVT - 1 > qpos_min) {  // workgroup-uniform
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sacc[mt][i] = kv0 + mt * 32 + crow(i, h) <= qpos ? sacc[mt][i] : -INFINITY;
        }
        float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, fmaxf(sacc[0][i], sacc[1][i]));   // v_max3_f32
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m, tmax);                       // in units of the raw (or soft-capped) logit
        const float alpha = m_new == -INFINITY ? 1.0f : exp2f((m - m_new) * cexp);
        m = m_new;
        if (__any(alpha != 1.0f)) {  // wave-uniform: after the first tiles the running max rarely moves
#pragma unroll
            for (int dt = 0; dt <= DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        const float mc = m == -INFINITY ? 0.0f : m * cexp;
        if constexpr (STAMP) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_sm += t2 - tt; tt = t2; }
        // ---- O^T += V^T . P^T ; the row sums l come from one extra MFMA against an all-ones A fragment, i.e. from the
        //      same f16-rounded weights PV applies -------------------------------------------------------------------
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half2_t pp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[mt][8 * s2 + 2 * j], cexp, -mc));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[mt][8 * s2 + 2 * j + 1], cexp, -mc));
                    pp[j] = __builtin_amdgcn_cvt_pkrtz(p0, p1);
                }
                const half8_t pf = *reinterpret_cast<const half8_t *>(pp);
                const int p0 = mt * 16 + 8 * s2 + 2 * h;  // token pairs p0, p0+1, p0+4, p0+5 = this half's k order
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const uint32_t * vb = reinterpret_cast<const uint32_t *>(sV) + dt * 32 + r;
                    const uint32_t w[4] = {vb[(size_t) p0 * D], vb[(size_t) (p0 + 1) * D], vb[(size_t) (p0 + 4) * D], vb[(size_t) (p0 + 5) * D]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8_t *>(w), pf, o[dt], 0, 0, 0);
                }
                o[DT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, o[DT], 0, 0, 0);
            }
        if (more) {                            // registers -> the other LDS image (last read two barriers ago)
            kst.write(sKb((it + 1) & 1), tid);
            vst.write(sVb((it + 1) & 1), tid);
        }
        if constexpr (STAMP) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory"); const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_pv += t2 - tt; tt = t2; }
    }
    if constexpr (STAMP) t_loop = __builtin_amdgcn_s_memtime();
    // ---- normalise and write out[tok][head][d] -----------------------------------------------------------------------------
    // O^T sits with the query on the lane: stored directly, every lane would write 16-byte pieces of ITS row (32 rows per
    // instruction, 32-64 cache lines touched each).  Instead each wave transposes its tile through LDS (the K/V images are
    // free now) and stores whole rows: 32 lanes x 16 B = one 512-byte head row per half-wave.
    {
        constexpr int DH = G <= 4 ? D : D / 2;                 // d columns staged per pass (wave strip = 32 rows x DH f32 <= 4*IMG / G)
        constexpr int CPRW = DH / 4;                           // 16-byte chunks per staged row
        static_assert(32 * DH * 4 * G <= 4 * IMG, "per-wave output strips must fit the tile images");
        float * strip = reinterpret_cast<float *>(smem) + (size_t) g * 32 * DH;
        const bool  partial = a.n_split > 1;
        const float inv = partial ? 1.0f : 1.0f / o[DT][0];
        if (partial && q_ok && h == 0) {                       // (m, l) of this share; an empty share leaves (-inf, 0)
            float * ml = a.part_ml + (((size_t) (tok0 + jr) * a.n_head + head) * a.n_split + sp) * 2;
            ml[0] = m == -INFINITY ? -INFINITY : m * cexp;
            ml[1] = o[DT][0];
        }
        __syncthreads();                                       // every wave is done reading the last K/V tile
#pragma unroll
        for (int pass = 0; pass < D / DH; ++pass) {
#pragma unroll
            for (int dt = pass * (DH / 32); dt < (pass + 1) * (DH / 32); ++dt)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const int c = ((dt * 32 - pass * DH) + crow(i, h)) / 4;   // registers i..i+3 are 4 consecutive d
                    *reinterpret_cast<float4 *>(strip + (size_t) r * DH + ((c ^ (r & 7)) * 4)) =
                        make_float4(o[dt][i] * inv, o[dt][i + 1] * inv, o[dt][i + 2] * inv, o[dt][i + 3] * inv);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip is private to this wave
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it2 = 0; it2 < 32 * CPRW / 64; ++it2) {
                const int e = it2 * 64 + lane, row = e / CPRW, c = e % CPRW;
                const float4 v = *reinterpret_cast<const float4 *>(strip + (size_t) row * DH + ((c ^ (row & 7)) * 4));
                if (tile * QT + row < n_q) {
                    const size_t th = (size_t) (tok0 + tile * QT + row) * a.n_head + head;
                    float * dst = partial ? a.part_o + (th * a.n_split + sp) * D : a.out + th * D;
                    *reinterpret_cast<float4 *>(dst + pass * DH + c * 4) = v;
                }
            }
            if (pass + 1 < D / DH) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long * o8 = a.stamps + ((size_t) (s * gridDim.y + hk) * gridDim.x + xr) * 8;
            o8[0] = t_q - t_begin; o8[1] = t_loop - t_q; o8[2] = t_end - t_loop; o8[3] = t_stage; o8[4] = t_qk; o8[5] = t_sm; o8[6] = t_pv; o8[7] = t_begin;
        }
    }
}

// merge of the split-KV shares: out = sum_j O_j 2^(m_j - M) / sum_j l_j 2^(m_j - M); one workgroup per (token, head).
// Shares are selected, never multiplied, when their weight is zero (an empty share's O is not written).
template <int D>
__global__ void __launch_bounds__(D / 4) prefill_combine_kernel(const PrefillArgs a) {
    const int head = blockIdx.x, tok = blockIdx.y, c = threadIdx.x;
    const size_t th = (size_t) tok * a.n_head + head;
    const float * ml = a.part_ml + th * a.n_split * 2;
    float M = -INFINITY;
    for (int j = 0; j < a.n_split; ++j) M = fmaxf(M, ml[2 * j]);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float  L = 0.0f;
    for (int j = 0; j < a.n_split; ++j) {
        const float mj = ml[2 * j];
        if (mj == -INFINITY) continue;
        const float w = exp2f(mj - M);
        const float4 v = *reinterpret_cast<const float4 *>(a.part_o + (th * a.n_split + j) * D + c * 4);
        acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y); acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
        L = fmaf(ml[2 * j + 1], w, L);
    }
    const float inv = 1.0f / L;   // every cell masked -> 0 * inf = NaN, as the single-share kernel's 1 / 0
    *reinterpret_cast<float4 *>(a.out + th * D + c * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

template <int KT, int VT, int D, int G>
int launch_one(const PrefillArgs & pa, int n_tiles, int n_seq, hipStream_t st) {
    if constexpr (KT == PKV_TYPE_F16 && VT == PKV_TYPE_F16 && D == 128 && G == 4) {
        if (pa.stamps) {
            hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false, true>), dim3(n_tiles * pa.n_split, pa.n_head_kv, n_seq), dim3(64 * G), 0, st, pa);
            PKV_HIP(hipGetLastError());
            return PKV_OK;
        }
    }
    const dim3 grid(n_tiles * pa.n_split, pa.n_head_kv, n_seq), block(64 * G);
    if (pa.mask) {
        if (pa.softcap != 0.0f) hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, true, false, true>), grid, block, 0, st, pa);
        else                    hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false, false, true>), grid, block, 0, st, pa);
    } else {
        if (pa.softcap != 0.0f) hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, true>), grid, block, 0, st, pa);
        else                    hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false>), grid, block, 0, st, pa);
    }
    if (pa.n_split > 1) hipLaunchKernelGGL((prefill_combine_kernel<D>), dim3(pa.n_head, pa.n_tok_total), dim3(D / 4), 0, st, pa);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
template <int KT, int VT, int D>
int launch_g(const PrefillArgs & pa, int G, int n_tiles, int n_seq, hipStream_t st) {
    switch (G) {
        case 1: return launch_one<KT, VT, D, 1>(pa, n_tiles, n_seq, st);
        case 2: return launch_one<KT, VT, D, 2>(pa, n_tiles, n_seq, st);
        case 4: return launch_one<KT, VT, D, 4>(pa, n_tiles, n_seq, st);
        case 8: return launch_one<KT, VT, D, 8>(pa, n_tiles, n_seq, st);
    }
    set_error("pkv_attn_prefill: unsupported gqa ratio %d (supported 1,2,4,8)", G);
    return PKV_ERR_UNSUPPORTED;
}
template <int KT, int VT>
int launch_d(const PrefillArgs & pa, int D, int G, int n_tiles, int n_seq, hipStream_t st) {
    switch (D) {
        case 64:  return launch_g<KT, VT, 64>(pa, G, n_tiles, n_seq, st);
        case 128: return launch_g<KT, VT, 128>(pa, G, n_tiles, n_seq, st);
    }
    set_error("pkv_attn_prefill: unsupported head_dim %d (supported 64,128)", D);
    return PKV_ERR_UNSUPPORTED;
}
template <int KT>
int launch_v(const PrefillArgs & pa, int VT, int D, int G, int n_tiles, int n_seq, hipStream_t st) {
    switch (VT) {
        case PKV_TYPE_F16:  return launch_d<KT, PKV_TYPE_F16>(pa, D, G, n_tiles, n_seq, st);
        case PKV_TYPE_Q8_0: return launch_d<KT, PKV_TYPE_Q8_0>(pa, D, G, n_tiles, n_seq, st);
        case PKV_TYPE_Q4_0: return launch_d<KT, PKV_TYPE_Q4_0>(pa, D, G, n_tiles, n_seq, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

}  // namespace

void * g_prefill_stamps = nullptr;  // set by pkv_debug_set_prefill_stamps (diagnostic)
void set_prefill_stamps(void * p) { g_prefill_stamps = p; }
// scratch of the split-KV path (opportunistic: a call whose workspace is smaller simply runs unsplit).  Splitting only
// pays when the grid is small, i.e. for at most a few thousand query tokens per launch.
constexpr int PREFILL_MAX_SPLIT = 8, PREFILL_SPLIT_MAX_TOK = 2048;
size_t prefill_workspace_bytes(int n_tok, int n_head, int D) {
    if (n_tok > PREFILL_SPLIT_MAX_TOK) return 0;
    return (size_t) n_tok * n_head * PREFILL_MAX_SPLIT * (D + 2) * sizeof(float);
}

int launch_attn_prefill(const pkv_attn_args & a, hipStream_t st) {
    PKV_REQUIRE(a.q && a.out && a.k_pool && a.v_pool && a.block_table && a.seq_lens && a.q_start, "pkv_attn_prefill: null pointer");
    PKV_REQUIRE(a.n_seq > 0 && a.n_tok > 0 && a.max_q_per_seq > 0, "pkv_attn_prefill: bad token counts");
    PKV_REQUIRE(a.n_head > 0 && a.n_head_kv > 0 && a.n_head % a.n_head_kv == 0, "pkv_attn_prefill: bad head counts");
    PKV_REQUIRE(a.block_size >= KVT && a.block_size % KVT == 0, "pkv_attn_prefill: block_size must be a multiple of %d", KVT);
    PKV_REQUIRE(kv_type_ok(a.type_k) && kv_type_ok(a.type_v), "pkv_attn_prefill: KV types must be f16/q8_0/q4_0");
    PKV_REQUIRE(a.max_seq_len > 0 && a.max_seq_len <= a.max_blocks * a.block_size, "pkv_attn_prefill: max_seq_len out of range");
    PKV_REQUIRE((a.q_stride_head % 4) == 0 && (a.q_stride_tok % 4) == 0 && ((uintptr_t) a.q % 16) == 0 && ((uintptr_t) a.out % 16) == 0,
                "pkv_attn_prefill: q/out must be 16-byte aligned");
    if (a.slot_pos) { set_error("pkv_attn_prefill: irregular sequences (slot_pos) are not supported yet; defrag first"); return PKV_ERR_UNSUPPORTED; }
    PrefillArgs pa;
    pa.q = a.q; pa.q_stride_tok = a.q_stride_tok; pa.q_stride_head = a.q_stride_head; pa.out = a.out;
    pa.k_pool = (const char *) a.k_pool; pa.v_pool = (const char *) a.v_pool;
    pa.block_table = a.block_table; pa.seq_lens = a.seq_lens; pa.q_start = a.q_start;
    pa.n_head = a.n_head; pa.n_head_kv = a.n_head_kv; pa.block_size = a.block_size; pa.max_blocks = a.max_blocks;
    pa.softcap = a.logit_softcap;
    pa.stamps = (unsigned long long *) g_prefill_stamps;
    pa.mask = (const _Float16 *) a.mask; pa.mask_stride = a.mask_stride; pa.max_bias = a.max_bias;
    pa.causal = (a.flags & PKV_ATTN_NONCAUSAL) ? 0 : 1;
    PKV_REQUIRE(a.max_bias == 0.0f || a.mask, "pkv_attn_prefill: max_bias (ALiBi) needs a mask");
    pa.scale   = a.logit_softcap != 0.0f ? a.scale / a.logit_softcap : a.scale;
    const int n_tiles = (a.max_q_per_seq + QT - 1) / QT, G = a.n_head / a.n_head_kv;
    // split-KV when the launch would leave most of the 256 CUs (2 workgroups each) idle and the past is long enough to share
    pa.n_split = 1; pa.n_tok_total = a.n_tok; pa.part_o = nullptr; pa.part_ml = nullptr;
    const long long n_wg = (long long) n_tiles * a.n_head_kv * a.n_seq;
    const int nkt_max = (a.max_seq_len + KVT - 1) / KVT;
    if (n_wg < 384 && a.n_tok <= PREFILL_SPLIT_MAX_TOK && !pa.stamps) {
        int ns = (int) std::min<long long>(PREFILL_MAX_SPLIT, (512 + n_wg - 1) / n_wg);
        ns = std::min(ns, std::max(1, nkt_max / 8));               // at least ~8 KV tiles (512 cells) per share: below that the merge launch costs more than it saves
        const size_t need = (size_t) a.n_tok * a.n_head * ns * (a.head_dim + 2) * sizeof(float);
        if (ns > 1 && a.workspace && a.workspace_bytes >= need) {
            pa.n_split = ns;
            pa.part_o  = (float *) a.workspace;
            pa.part_ml = pa.part_o + (size_t) a.n_tok * a.n_head * ns * a.head_dim;
        }
    }
    switch (a.type_k) {
        case PKV_TYPE_F16:  return launch_v<PKV_TYPE_F16>(pa, a.type_v, a.head_dim, G, n_tiles, a.n_seq, st);
        case PKV_TYPE_Q8_0: return launch_v<PKV_TYPE_Q8_0>(pa, a.type_v, a.head_dim, G, n_tiles, a.n_seq, st);
        case PKV_TYPE_Q4_0: return launch_v<PKV_TYPE_Q4_0>(pa, a.type_v, a.head_dim, G, n_tiles, a.n_seq, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

}  // namespace pkv
