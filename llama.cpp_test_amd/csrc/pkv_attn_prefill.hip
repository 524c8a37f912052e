// pkv_attn_prefill.hip — batched-prefill paged attention for gfx950: flash-style, MFMA (v_mfma_f32_32x32x16_f16).
//
// Replaces, for many query tokens per sequence, the K/V gather views + GGML_OP_FLASH_ATTN_EXT + dense causal KQ_mask
// of the reference (/ref/src/llama-graph.cpp:1404-1432, :403-468; CPU kernel /ref/ggml/src/ggml-cpu/ggml-cpu.c:12219).
//
// Decomposition: one workgroup per (sequence, kv head, 32-query tile); wave g of the workgroup owns query head
// hk*G + g, so the K/V tile staged in LDS is read from HBM once for all G = n_head/n_head_kv query heads; the kv head is
// the fastest-varying workgroup index, so with 8 kv heads every XCD (its own L2) works on ONE head's K/V.
// Per 64-token KV tile (half a 128-token block — tiles never straddle blocks; the block table is consulted per tile, by a
// scalar load issued one tile ahead):
//   S^T[kv][q] = K[kv][:] . Q^T       A = K rows from LDS (XOR-swizzled 16-B chunks, conflict-free ds_read_b128),
//                                     B = Q^T fragments held in registers for the whole kernel;
//                                     the accumulator has the QUERY on the lane, so the online-softmax state (m, l)
//                                     is per lane and the row max/sum needs one cross-half exchange only
//   O^T[d][q] += V^T[d][kv] . P^T     B = P^T straight from the S^T accumulators (cdna_hip_programming.md §3
//                                     "An accumulator tile as the next MFMA's operand", permuted k order),
//                                     A = V^T from an LDS image laid out in exactly that k order: ONE ds_read_b128 per MFMA
// Round 2 (profiles/r02_prefill_*): operand reads issued two groups AHEAD of their MFMAs by hand (hipcc put every ds_read next
// to its MFMA behind lgkmcnt(0): one LDS latency per MFMA); table / length lookups as scalar loads (through the by-value
// argument struct they were vector loads + vmcnt(0), which drained the tile's prefetch every time); all staging addresses =
// one per-thread 32-bit offset + constants through buffer descriptors that end with the tile's last valid row; f16 K tiles by
// LDS-DMA; Q fetched as whole rows and turned through LDS; O stored as whole rows through a descriptor; row sums on the VALU.
// Causal structure replaces the mask: KV tiles past the tile's last query position are never visited; the diagonal tile
// masks per element.  Quantized K/V (q8_0, q4_0) are dequantized to f16 while staging; for quantized K the Q rows are rounded to the
// reference's q8_0 values first (q8_round below), so the output meets the 1e-3 bar against the reference (tests/test_gpu_prefill.py).
#include "pkv_common.h"
#include "pkv_rowcodec.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace pkv {
extern void * g_prefill_stamps;
namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __fp16 half2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const void * p, long long bytes) {   // raw buffer: reads past `bytes` return 0, stores are dropped
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int) (bytes < 0 ? 0 : bytes), 0x00020000);
}

#ifndef PKV_PREFILL_ABLATE
#define PKV_PREFILL_ABLATE 0   // timing-only experiments (wrong results; profiles/r02_prefill_ablation.txt): 1 no tile barrier, 2 no softmax arithmetic,
                               // 4 no prefetch and no LDS stores, 8 no QK^T MFMAs, 16 no PV MFMAs, 64 no prefetch loads, 256 loads waited for but not stored
#endif
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
    int   kv_div, n_head_kv_pool; // gqa ratios 16 / 32 as ratio 8 over kv_div times as many virtual kv heads (see pkv_attn_decode.hip)
    int   bs_shift;               // log2(block_size), or -1 when it is not a power of two
    int   mask_align8;            // mask rows start on 8-byte boundaries (base and row stride)
    const float * rope_cache;     // pkv_attn_prefill_step: (cos, sin) rows of the ubatch's tokens [n_tok][D/2][2] (pkv_rope_cache), NORM pairs: Q arrives
                                  // un-rotated and is rotated on its way into the QK^T operand; NULL = q is used as given
    int   q_f16, out_f16;         // PKV_ATTN_Q_F16 / PKV_ATTN_OUT_F16: q / out are f16 (strides in elements): half the boundary traffic of a prompt ubatch
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

// Dequantization while staging: 4 quants -> 4 halfs times their scales with packed f16 arithmetic.  A byte b under 0x64 is the half 1024 + b
// (exact), so one v_perm_b32 places two quants as two halfs, one packed add removes the bias (1024 + 128 for int8 ^ 0x80, 1024 + 8 for a q4_0
// nibble) and one packed multiply applies the f16 scales: q * d is exact in 19 bits and rounded once, the value (f16) ((float) q * (float) d)
// that the scalar conversion produced with 3 instructions per element.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2_t as_h2(uint32_t u) { return *reinterpret_cast<const h2_t *>(&u); }
__device__ __forceinline__ uint32_t as_u(h2_t h) { return *reinterpret_cast<const uint32_t *>(&h); }
// b = 4 biased quants in 4 bytes (element j in byte j); d01 / d23 = the scales of elements (0, 1) / (2, 3)
__device__ __forceinline__ void dq4(uint32_t b, float bias, h2_t d01, h2_t d23, uint32_t & o01, uint32_t & o23) {
    const h2_t nb = {(_Float16) -bias, (_Float16) -bias};
    o01 = as_u((as_h2(__builtin_amdgcn_perm(0x64646464u, b, 0x04010400u)) + nb) * d01);
    o23 = as_u((as_h2(__builtin_amdgcn_perm(0x64646464u, b, 0x04030402u)) + nb) * d23);
}

// Q rows for quantized K.  The reference quantizes every Q row to q8_0 before QK^T (quantize_row_q8_0, AVX2 branch, /ref/ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862:
// per 32-element block d = amax / 127 stored as f16, id = 127 / amax, round-half-even) and multiplies integers; r02 kept Q in f16 here, closer to exact arithmetic
// but up to 3e-3 away from the reference's output.  Now the Q row takes the reference's rounding: a lane holds 4 consecutive elements of a row, a block is 8 lanes
// of one DPP row; the operand value q_i * f16(d) (19 significant bits) is then rounded to f16 — relative 2^-12, against the 2^-8 of the quantization it follows.
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void q8_round(float & x0, float & x1, float & x2, float & x3) {
#pragma clang fp contract(off)
    float am = fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(x2), fabsf(x3)));
    am = fmaxf(am, dpp_f<0xB1>(am));    // quad_perm [1,0,3,2]
    am = fmaxf(am, dpp_f<0x4E>(am));    // quad_perm [2,3,0,1]
    am = fmaxf(am, dpp_f<0x141>(am));   // row_half_mirror: the block's 8 lanes
    const float d  = am / 127.0f;
    const float id = am != 0.0f ? 127.0f / am : 0.0f;
    const float dh = (float) (_Float16) d;
    x0 = __builtin_rintf(x0 * id) * dh; x1 = __builtin_rintf(x1 * id) * dh;
    x2 = __builtin_rintf(x2 * id) * dh; x3 = __builtin_rintf(x3 * id) * dh;
}

// Staging of one 64-token K tile: HBM -> registers (load) now, registers -> LDS (write) one loop iteration later, so the
// loads of tile t+1 are in flight while tile t is being multiplied (cdna_hip_programming.md T14, async-STAGE split).
// LDS image: f16 rows of D halfs, 16-byte chunk c of row r stored at chunk c ^ (r & 15) (conflict-free ds_read_b128).
// Loads go through a buffer descriptor that covers exactly the tile's valid rows: every address is a per-thread 32-bit offset
// computed once + a compile-time constant, and rows past the end of the sequence read as 0 (the causal test removes them).
template <int KT, int D, int NT>
struct KStage {
    static constexpr int CPR  = D / 8;            // 16-byte chunks per f16 row
    static constexpr int NI   = KVT * CPR / NT;   // chunks per thread
    static constexpr int RSTP = NT / CPR;         // rows between a thread's chunks
    static constexpr int ROWB = KT == PKV_TYPE_F16 ? D * 2 : (KT == PKV_TYPE_Q8_0 ? D : D / 2);
    static constexpr int NR   = KT == PKV_TYPE_F16 ? 4 : 2;
    static_assert(KVT * CPR % NT == 0 && NT % CPR == 0, "tile must divide evenly over the workgroup");
    uint32_t raw[NI][NR];
    uint32_t dsc[KT == PKV_TYPE_F16 ? 1 : NI];    // f16 block scale bits

    static __device__ __forceinline__ int goff(int tid) {          // byte offset of the thread's first chunk inside the tile rows
        const int r = tid / CPR, cc = tid % CPR;
        if constexpr (KT == PKV_TYPE_F16)       return r * ROWB + cc * 16;
        else if constexpr (KT == PKV_TYPE_Q8_0) return r * ROWB + cc * 8;
        else                                    return r * ROWB + (cc / 4) * 16 + ((cc & 1) * 8);   // q4_0: byte j of a block = elem j | elem j+16 << 4
    }
    static __device__ __forceinline__ int soff(int tid) { return ((tid / CPR) * (D / PKV_QK) + (tid % CPR) / 4) * 2; }

    __device__ __forceinline__ void load(rsrc_t rs, rsrc_t rs_s, int go, int so) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (KT == PKV_TYPE_F16) {
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, go + i * RSTP * ROWB, 0, 0);
                raw[i][0] = v.x; raw[i][1] = v.y; raw[i][2] = v.z; raw[i][3] = v.w;
            } else {
                const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, go + i * RSTP * ROWB, 0, 0);
                raw[i][0] = v.x; raw[i][1] = v.y;
                dsc[i] = __builtin_amdgcn_raw_buffer_load_b16(rs_s, so + i * RSTP * (D / PKV_QK) * 2, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void write(char * lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = tid / CPR + i * RSTP, cc = tid % CPR;
            uint4 v;
            if constexpr (KT == PKV_TYPE_F16) {
                v = make_uint4(raw[i][0], raw[i][1], raw[i][2], raw[i][3]);
            } else {
                const h2_t dd = as_h2((dsc[i] & 0xFFFFu) * 0x10001u);          // the block scale in both halves
                uint32_t o[4];
                if constexpr (KT == PKV_TYPE_Q8_0) {
                    dq4(raw[i][0] ^ 0x80808080u, 1152.0f, dd, dd, o[0], o[1]);
                    dq4(raw[i][1] ^ 0x80808080u, 1152.0f, dd, dd, o[2], o[3]);
                } else {
                    const int sh = (cc & 2) ? 4 : 0;
                    dq4((raw[i][0] >> sh) & 0x0F0F0F0Fu, 1032.0f, dd, dd, o[0], o[1]);
                    dq4((raw[i][1] >> sh) & 0x0F0F0F0Fu, 1032.0f, dd, dd, o[2], o[3]);
                }
                v = make_uint4(o[0], o[1], o[2], o[3]);
            }
#if PKV_PREFILL_ABLATE & 256   // timing experiment: wait for the loads, skip the LDS store
            asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
            *reinterpret_cast<uint4 *>(lds + r * D * 2 + ((cc ^ (r & 15)) & (CPR - 1)) * 16) = v;
#endif
        }
    }
};

// f16 K: no conversion on the way, so the tile goes HBM -> LDS directly (LDS-DMA: buffer_load_dwordx4 ... lds), past the registers and
// past the VGPR -> LDS store path (13 cycles per ds_write_b128 and wave; timing ablation: the LDS stores of the register-staged tile cost
// 17 % of the loop).  One instruction fills 1 KiB of the image = RPI whole rows, lane l -> row l / CPR, chunk POSITION l % CPR; the image's
// XOR swizzle is applied on the source side: position p of row r holds chunk p ^ (r & 15).
template <int D, int NT>
struct KStage<PKV_TYPE_F16, D, NT> {
    static constexpr int CPR = D / 8, ROWB = D * 2, RPI = 1024 / ROWB, NIN = KVT / RPI, NW = NT / 64, NI = NIN / NW;
    static_assert(NIN % NW == 0 && NI >= 1, "tile must divide evenly over the waves");
    int vo[NI];                                   // source byte offsets of this lane's chunks, fixed for the kernel
    __device__ __forceinline__ void init(int tid) {
        const int l = tid & 63, w = tid >> 6;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = (w * NI + i) * RPI + l / CPR;
            vo[i] = row * ROWB + (((l % CPR) ^ (row & 15)) & (CPR - 1)) * 16;
        }
    }
    __device__ __forceinline__ void dma(rsrc_t rs, char * img, int wave) const {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass has no LDS address space: with the cast in sight it drops the kernel's launch stub)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *) (img + (wave * NI + i) * 1024), 16, vo[i], 0, 0, 0);
#endif
    }
};

// V tile: LDS image = V^T in MFMA-operand order: row d holds the tile's 64 tokens as 8 chunks of 16 bytes, chunk kg = mt*4 + s2*2 + h
// = the 8 f16 values {tokens u0..u0+3, u0+8..u0+11}, u0 = 32 mt + 16 s2 + 4 h, that lane (d, h) feeds to the PV MFMA of (mt, s2) as its
// A fragment (the k order of the P^T fragment it multiplies: accumulator rows crow(8 s2 + j, h)) -> ONE ds_read_b128 per MFMA where a
// token-pair image needs four ds_read_b32 (128 B/clk, and far below that at two waves per SIMD: MI355X_MICROARCH.md LDS table).
// The pool's V tile is [token group][d] dwords (2 / 4 / 8 tokens per dword for f16 / q8_0 / q4_0), so a thread that owns (d, kg) loads
// 4 (f16) or 2 (quantized) dwords at a row stride — consecutive lanes = consecutive d: every load instruction is one contiguous run —
// and writes ONE 16-byte chunk.  Chunk kg of row d sits at chunk kg ^ vswz(d): conflict-free for the b128 row reads (lanes = d) AND
// for these writes.  Groups past the end of the sequence read as 0 through the buffer descriptor; the tokens of the last, partly
// valid group are zeroed when written (0 * stale NaN must not happen).
__device__ __forceinline__ int vswz(int d) { return (d & 7) ^ ((d >> 4) & 1); }

template <int VT, int D, int NT>
struct VStage {
    static constexpr int NCH = D * 8;                    // 16-byte chunks of the image
    static constexpr int NI  = NCH / NT;                 // chunks per thread
    static_assert(NCH % NT == 0 && NT % 64 == 0 && (NT % D == 0 || D % NT == 0), "tile must divide evenly over the workgroup");
    static constexpr int TL = VT == PKV_TYPE_F16 ? 2 : (VT == PKV_TYPE_Q8_0 ? 4 : 8);
    static constexpr int NR = VT == PKV_TYPE_F16 ? 4 : 2;
    uint32_t raw[NI][NR];
    u2_t     sc[VT == PKV_TYPE_F16 ? 1 : NI][2];         // quantized: the 4 f16 token scales of each loaded dword
    int      nval_;

    // chunk i of thread tid = (d, kg): (tid + i NT) % D, (tid + i NT) / D, split into a per-thread part (computed once) and a per-i
    // constant: NT >= D: d = tid % D, kg = tid / D + i (NT / D) (disjoint bits); NT < D: d = tid + NT (i % (D / NT)), kg = i / (D / NT)
    static constexpr int DPT = NT >= D ? 1 : D / NT;
    static __device__ __forceinline__ int d_t(int tid) { return NT >= D ? tid % D : tid; }
    static __device__ __forceinline__ int kg_t(int tid) { return NT >= D ? tid / D : 0; }
    static constexpr int d_c(int i) { return NT >= D ? 0 : NT * (i % DPT); }
    static constexpr int kg_c(int i) { return NT >= D ? i * (NT / D) : i / DPT; }
    static __host__ __device__ constexpr int u0_of(int kg) { return (kg >> 2) * 32 + ((kg >> 1) & 1) * 16 + (kg & 1) * 4; }   // linear over kg's bits
    static __device__ __forceinline__ int u0t(int tid) { return u0_of(kg_t(tid)); }
    // byte offsets of token group `grp` (a dword row of the tile) and of its scales, for column d
    static __device__ __forceinline__ int goff(int tid) { return (u0_of(kg_t(tid)) / TL * D + d_t(tid)) * 4; }
    static __device__ __forceinline__ int soff(int tid) {
        const int u0 = u0_of(kg_t(tid));
        return ((u0 / TL * (D / PKV_QK) + d_t(tid) / PKV_QK) * TL + u0 % TL) * 2;
    }
    static __device__ __forceinline__ int woff(int tid) {          // LDS byte offset of chunk 0; chunk i: ^ (kg_c(i) * 16), + d_c(i) rows
        const int d = d_t(tid);
        return d * (KVT * 2) + ((kg_t(tid) ^ vswz(d)) & 7) * 16;
    }

    __device__ __forceinline__ void load(rsrc_t rs, rsrc_t rs_s, int n_rows_valid, int go, int so) {
        nval_ = n_rows_valid;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u0c = u0_of(kg_c(i));              // compile-time: the thread's own u0 part is a multiple of TL or below 8
            if constexpr (VT == PKV_TYPE_F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    raw[i][e] = __builtin_amdgcn_raw_buffer_load_b32(rs, go, ((u0c / 2 + (e & 1) + (e >> 1) * 4) * D + d_c(i)) * 4, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int gc = u0c / TL + e * (8 / TL);
                    raw[i][e] = __builtin_amdgcn_raw_buffer_load_b32(rs, go, (gc * D + d_c(i)) * 4, 0);
                    sc[i][e]  = __builtin_amdgcn_raw_buffer_load_b64(rs_s, so, ((gc * (D / PKV_QK) + d_c(i) / PKV_QK) * TL + u0c % TL) * 2, 0);
                }
            }
        }
    }
    __device__ __forceinline__ void write(char * lds, int wo, int u0t) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u0 = u0t + u0_of(kg_c(i));
            uint32_t o[4];
            if constexpr (VT == PKV_TYPE_F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = raw[i][e];
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const u2_t s2 = sc[i][e];                          // the 4 tokens' f16 scales: (s0, s1) | (s2, s3)
                    if constexpr (VT == PKV_TYPE_Q8_0) {
                        dq4(raw[i][e] ^ 0x80808080u, 1152.0f, as_h2(s2.x), as_h2(s2.y), o[2 * e], o[2 * e + 1]);
                    } else {                                               // nibble j of w = token j: (n1 n0) (n3 n2) -> bytes n0, n2 | n1, n3
                        const uint32_t w = raw[i][e] >> (4 * (u0 & 7));
                        const uint32_t b = (w & 0x0F0Fu) | (((w >> 4) & 0x0F0Fu) << 16);
                        const h2_t nb = {(_Float16) -1032.0f, (_Float16) -1032.0f};
                        o[2 * e]     = as_u((as_h2(__builtin_amdgcn_perm(0x64646464u, b, 0x04020400u)) + nb) * as_h2(s2.x));
                        o[2 * e + 1] = as_u((as_h2(__builtin_amdgcn_perm(0x64646464u, b, 0x04030401u)) + nb) * as_h2(s2.y));
                    }
                }
            }
            if (nval_ < KVT) {                           // workgroup-uniform: only a sequence's last tile
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t0 = u0 + (e >> 1) * 8 + (e & 1) * 2;
                    o[e] &= t0 >= nval_ ? 0u : (t0 + 1 >= nval_ ? 0xFFFFu : 0xFFFFFFFFu);
                }
            }
#if PKV_PREFILL_ABLATE & 256
            asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
#else
            *reinterpret_cast<uint4 *>(lds + ((wo ^ (kg_c(i) * 16)) + d_c(i) * (KVT * 2))) = make_uint4(o[0], o[1], o[2], o[3]);
#endif
        }
    }
};

// f16 V: the pool tile is [token pair][d] dwords.  A thread that owns (d, kg) needs four 4-byte loads per chunk — 16 load instructions
// per wave and tile, and the vector memory pipe is paid per instruction.  Here a thread loads 16 bytes = one pair x 4 columns (4 load
// instructions per wave and tile) and scatters the 4 dwords into 4 rows of the V^T image (ds_write_b32, as cheap on the LDS store path
// as the ds_write_b128 they replace: 2 cycles per source dword).  A wave instruction covers 16 pairs x 4 chunks of columns: lanes 4a..4a+3
// read 64 contiguous bytes, and the scattered dwords of a 32-lane half fall on every bank at most twice (free for ds_write_b32;
// brute-forced over all units).  Unit q = wave * NI + i: pair block q & 1, column block q >> 1.
template <int D, int NT>
struct VStage<PKV_TYPE_F16, D, NT> {
    static constexpr int TL = 2;
    static constexpr int NW = NT / 64, NU = (KVT / 2) * (D / 4) / 64, NI = NU / NW;
    static_assert(NU % NW == 0 && (NI == 1 || NI % 2 == 0), "units must divide evenly over the waves");
    uint32_t raw[NI][4];
    int      nval_;
    static __device__ __forceinline__ int pb_t(int tid) { return NI == 1 ? ((tid >> 6) & 1) : 0; }
    static __device__ __forceinline__ int cb_t(int tid) { return NI == 1 ? (tid >> 7) : (tid >> 6) * (NI / 2); }
    static constexpr int pb_c(int i) { return NI == 1 ? 0 : (i & 1); }
    static constexpr int cb_c(int i) { return NI == 1 ? 0 : (i >> 1); }
    static __device__ __forceinline__ int goff(int tid) {
        const int l = tid & 63;
        return (((l >> 2) + 16 * pb_t(tid)) * D + ((l & 3) + 4 * cb_t(tid)) * 4) * 4;
    }
    static __device__ __forceinline__ int soff(int) { return 0; }
    static __device__ __forceinline__ int u0t(int tid) { return 2 * (((tid & 63) >> 2) + 16 * pb_t(tid)); }   // first token of the thread's pair
    // LDS byte offset of dword e = 0 of unit part i = 0: row d = 16 cb + 4 (l & 3) + e, chunk field (kg ^ vswz(d)) & 7, dword w of the chunk
    static __device__ __forceinline__ int woff(int tid) {
        const int l = tid & 63, cb = cb_t(tid);
        const int kgl = (l >> 5) * 2 + ((l >> 3) & 1);                  // s2, h of token 2 (l >> 2)
        const int f0  = (kgl ^ (4 * (l & 1)) ^ (4 * pb_t(tid)) ^ (cb & 1)) & 7;
        const int wd  = ((l >> 4) & 1) * 2 + ((l >> 2) & 1);
        return (4 * (l & 3)) * (KVT * 2) + cb * 16 * (KVT * 2) + f0 * 16 + wd * 4;
    }
    __device__ __forceinline__ void load(rsrc_t rs, rsrc_t, int n_rows_valid, int go, int) {
        nval_ = n_rows_valid;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, go, (16 * pb_c(i) * D + 16 * cb_c(i)) * 4, 0);
            raw[i][0] = v.x; raw[i][1] = v.y; raw[i][2] = v.z; raw[i][3] = v.w;
        }
    }
    __device__ __forceinline__ void write(char * lds, int wo, int u0) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            uint32_t keep = 0xFFFFFFFFu;
            if (nval_ < KVT) {                           // workgroup-uniform: only a sequence's last tile
                const int t0 = u0 + 32 * pb_c(i);
                keep = t0 >= nval_ ? 0u : (t0 + 1 >= nval_ ? 0xFFFFu : 0xFFFFFFFFu);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = ((4 * pb_c(i)) ^ e ^ (cb_c(i) & 1)) * 16, y = e * (KVT * 2) + cb_c(i) * 16 * (KVT * 2);
#if PKV_PREFILL_ABLATE & 256
                asm volatile("" :: "v"(raw[i][e] & keep));
#else
                *reinterpret_cast<uint32_t *>(lds + ((wo ^ x) + y)) = raw[i][e] & keep;
#endif
            }
        }
    }
};

// q8_0 V, wide form: the pool tile is [group of 4 tokens][d] dwords (+ 4 f16 token scales per group and 32 columns).  A thread loads 16 bytes = one group x
// 4 columns and the group's 4 scales (2 + 2 load instructions per wave and tile where the per-(d, kg) form issues 16), dequantizes each dword (4 tokens of
// one column) with packed f16 arithmetic and stores it as one 8-byte half chunk: lanes 0-15 of a store take the 16 groups of ONE column = the row's 128
// bytes = every bank once.  Unit q = wave * NI + i: columns 16 q .. 16 q + 15.
template <int D, int NT>
struct VStageQ8W {
    static constexpr int TL = 4;
    static constexpr int NW = NT / 64, NU = (KVT / 4) * (D / 4) / 64, NI = NU / NW;
    static_assert(NU % NW == 0 && (NI == 1 || NI % 2 == 0), "units must divide evenly over the waves");
    u4_t raw[NI];
    u2_t sc[NI];
    int  nval_;
    static __device__ __forceinline__ int u_t(int tid) { return (tid >> 6) * NI; }              // first unit of the thread's wave
    static __device__ __forceinline__ int goff(int tid) { const int l = tid & 63; return ((l & 15) * D + ((l >> 4) + 4 * u_t(tid)) * 4) * 4; }
    static __device__ __forceinline__ int soff(int tid) { const int l = tid & 63; return ((l & 15) * (D / PKV_QK) + (NI == 1 ? u_t(tid) >> 1 : u_t(tid) / 2)) * 8; }
    static __device__ __forceinline__ int u0t(int tid) { return 4 * (tid & 15); }                // first token of the thread's group
    static __device__ __forceinline__ int woff(int tid) {
        const int l = tid & 63, g = l & 15, u = u_t(tid);
        const int kg = (g >> 3) * 4 + ((g >> 2) & 1) * 2 + (g & 1), jh = (g >> 1) & 1;
        return (4 * (l >> 4)) * (KVT * 2) + u * 16 * (KVT * 2) + ((kg ^ (4 * ((l >> 4) & 1)) ^ (u & 1)) & 7) * 16 + jh * 8;
    }
    __device__ __forceinline__ void load(rsrc_t rs, rsrc_t rs_s, int n_rows_valid, int go, int so) {
        nval_ = n_rows_valid;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, go, 64 * i, 0);
            sc[i]  = __builtin_amdgcn_raw_buffer_load_b64(rs_s, so, (NI == 1 ? 0 : i >> 1) * 8, 0);
        }
    }
    __device__ __forceinline__ void write(char * lds, int wo, int u0) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
            uint32_t k01 = 0xFFFFFFFFu, k23 = 0xFFFFFFFFu;
            if (nval_ < KVT) {                           // workgroup-uniform: only a sequence's last tile
                k01 = u0 >= nval_ ? 0u : (u0 + 1 >= nval_ ? 0xFFFFu : 0xFFFFFFFFu);
                k23 = u0 + 2 >= nval_ ? 0u : (u0 + 3 >= nval_ ? 0xFFFFu : 0xFFFFFFFFu);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t o0, o1;
                dq4(w[e] ^ 0x80808080u, 1152.0f, as_h2(sc[i].x), as_h2(sc[i].y), o0, o1);
                const int x = ((e ^ (i & 1)) & 7) * 16, y = e * (KVT * 2) + i * 16 * (KVT * 2);
                *reinterpret_cast<uint2 *>(lds + ((wo ^ x) + y)) = make_uint2(o0 & k01, o1 & k23);
            }
        }
    }
};
template <int VT, int D, int NT> struct VPick { using type = VStage<VT, D, NT>; };
template <int D, int NT> struct VPick<PKV_TYPE_Q8_0, D, NT> {
    static constexpr int NU = (KVT / 4) * (D / 4) / 64, NW = NT / 64;
    static constexpr bool wide = NU % NW == 0 && (NU / NW == 1 || (NU / NW) % 2 == 0);
    using type = typename std::conditional<wide, VStageQ8W<D, NT>, VStage<PKV_TYPE_Q8_0, D, NT>>::type;
};

// QB = 32-query blocks per wave.  QB = 2 (one wave per SIMD, the whole 512-entry register file; built with -amdgpu-mfma-vgpr-form, see the
// Makefile): every K / V^T fragment read from LDS feeds TWO MFMAs, and a staged KV tile, a barrier and a table lookup serve 64 queries per head
// instead of 32.  Measured slower than QB = 1 at two waves per SIMD (launch_attn_prefill): opt-in experiment.
template <int KT, int VT, int D, int G, bool SOFTCAP, bool STAMP = false, bool MASKED = false, int QB = 1>
__global__ void __launch_bounds__(64 * G, QB == 2 ? 1 : 2) attn_prefill_kernel(const int32_t * __restrict__ p_table, const int32_t * __restrict__ p_lens,
                                                                 const int32_t * __restrict__ p_qstart, const PrefillArgs a) {
    // the three leading arguments (= a.block_table, a.seq_lens, a.q_start) as explicit restrict-qualified parameters: hipcc reads them with
    // SCALAR loads.  Through the by-value struct it used a vector load + s_waitcnt vmcnt(0) + v_readfirstlane for every table entry, i.e.
    // the wave stood still until all of the tile's prefetch loads had returned (the largest single stall of the r01 kernel).
    unsigned long long t_begin = 0, t_q = 0, t_loop = 0, t_stage = 0, t_qk = 0, t_sm = 0, t_pv = 0, tt = 0;
    if constexpr (STAMP) t_begin = __builtin_amdgcn_s_memtime();
    constexpr int NT = 64 * G, KS = D / 16, DT = D / 32;  // k-steps of QK^T, 32-row d tiles of O^T
    constexpr int QTW = QT * QB;                          // query rows per wave = per workgroup tile
    constexpr bool PIN = !SOFTCAP && !MASKED;             // hand-placed operand reads (see the loop); the rare variants leave it to hipcc
    // two LDS images per operand: tile t+1 is written while slower waves still multiply tile t -> ONE barrier per tile
    constexpr int IMG = KVT * D * 2;                                  // one f16 tile image
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];       // K0, V0, K1, V1; Q strips (prologue) and O strips (epilogue) go through all of it
    auto sKb = [&](int i) -> char * { return smem + i * 2 * IMG; };
    auto sVb = [&](int i) -> char * { return smem + i * 2 * IMG + IMG; };

    // Causal work grows with the query tile index (tile t visits t/2 + 1 KV tiles): dispatch order = longest first, over
    // ALL sequences and heads (workgroups are handed out in linear block-id order), so the short tiles fill the tail.
    const int lin  = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nhs = gridDim.y * gridDim.z;
    const int xr   = gridDim.x - 1 - lin / nhs;                     // x = tile * n_split + split, longest tiles first
    const int tile = xr / a.n_split, sp = xr % a.n_split, hk = (lin % nhs) % gridDim.y, s = (lin % nhs) / gridDim.y;
    const int tok0 = p_qstart[s], n_q = p_qstart[s + 1] - tok0;
    if (tile * QTW >= n_q) return;
    const int seq_len = p_lens[s];
    const int tid = threadIdx.x, g = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;   // g: wave-uniform, and known to be
    const int head = hk * G + g;
    int jr[QB], qpos[QB];                    // this lane's queries (one per block) inside the sequence's new tokens; the last KV cell each sees
    bool q_ok[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        jr[b]   = tile * QTW + b * QT + r;
        q_ok[b] = jr[b] < n_q;
        qpos[b] = a.causal ? seq_len - n_q + jr[b] : seq_len - 1;
    }
    const int n_rows = min(QTW, n_q - tile * QTW);   // query rows of this tile that exist
    // the workgroup-uniform bound of the KV loop
    const int kv_all = a.causal ? min(seq_len, seq_len - n_q + tile * QTW + QTW) : seq_len;   // cells this query tile can see
    const int nkt    = (kv_all + KVT - 1) / KVT;                                             // ... in KV tiles
    const int kv_beg = (int) ((long long) nkt * sp / a.n_split) * KVT;                       // this workgroup's share
    const int kv_end = min(kv_all, (int) ((long long) nkt * (sp + 1) / a.n_split) * KVT);

    floatx16 o[QB][DT];
    float m[QB], lsum[QB];              // lsum: this lane's share of l (its 32 of the tile's 64 cells), summed from the f16-rounded weights PV applies
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        m[b] = -INFINITY; lsum[b] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[b][dt][i] = 0.0f;
    }
    const half2_t one2 = {(__fp16) 1.0f, (__fp16) 1.0f};
    const float cexp = ((SOFTCAP || MASKED) ? 1.0f : a.scale) * 1.44269504088896340736f;   // exp(x * scale) = exp2(x * cexp)
    const int qpos_min = a.causal ? seq_len - n_q + tile * QTW : seq_len - 1;                  // smallest last-visible cell in the tile
    float slope = 1.0f;
    if constexpr (MASKED) {
        if (a.max_bias > 0.0f) {
            const uint32_t n_head_log2 = 1u << (31 - __clz(a.n_head));
            const float m0 = exp2f(-a.max_bias / n_head_log2), m1 = exp2f(-(a.max_bias / 2.0f) / n_head_log2);
            slope = (uint32_t) head < n_head_log2 ? powf(m0, head + 1) : powf(m1, 2 * (head - n_head_log2) + 1);
        }
    }
    // mask rows of this tile's queries: row = token index, column = kv cell; this lane reads row r (row 0 for a query that does not exist)
    [[maybe_unused]] const rsrc_t rs_mask = mk_rsrc(MASKED ? a.mask + (size_t) (tok0 + tile * QTW) * a.mask_stride : nullptr,
                                                    MASKED ? ((long long) (n_rows - 1) * a.mask_stride + seq_len) * 2 : 0);
    [[maybe_unused]] int mask_vo[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) mask_vo[b] = (int) ((q_ok[b] ? b * QT + r : 0) * a.mask_stride * 2) + 8 * h;

    const int vsw = vswz(r) ^ h;             // swizzle of this lane's V^T rows (d = 32 dt + r: the same for every dt), its k half folded in
    const size_t tb_k = tile_bytes(KT, a.block_size, D), tb_v = tile_bytes(VT, a.block_size, D);
    using KSt = KStage<KT, D, NT>;
    using VSt = typename VPick<VT, D, NT>::type;
    KSt kst;
    VSt vst;
    int k_go = 0, k_so = 0;
    if constexpr (KT != PKV_TYPE_F16) { k_go = KSt::goff(tid); k_so = KSt::soff(tid); }
    const int v_go = VSt::goff(tid), v_so = VSt::soff(tid), v_wo = VSt::woff(tid), v_u0 = VSt::u0t(tid);
    auto blk_of = [&](int kv) { return a.bs_shift >= 0 ? kv >> a.bs_shift : kv / a.block_size; };
    // the table entry of a tile is looked up one tile ahead (a scalar load that returns under the previous tile's arithmetic, not
    // between the barrier and the tile's loads)
    const int32_t * trow = p_table + (size_t) s * a.max_blocks;
    const int last_blk = blk_of(max(seq_len - 1, 0));
    constexpr bool KDMA = KT == PKV_TYPE_F16;    // K tile by LDS-DMA (no staging registers, no LDS stores)
    if constexpr (KDMA) kst.init(tid);
    auto prefetch_k = [&](int kv0, int phys, char * img) {
        const int row0 = kv0 - blk_of(kv0) * a.block_size, nval = min(KVT, seq_len - kv0);
        const char * kt = a.k_pool + ((size_t) phys * a.n_head_kv_pool + (a.kv_div > 1 ? hk / a.kv_div : hk)) * tb_k;
        const rsrc_t rs = mk_rsrc(kt + (size_t) row0 * KSt::ROWB, nval * KSt::ROWB);
        if constexpr (KDMA) kst.dma(rs, img, g);
        else kst.load(rs, mk_rsrc(kt + tile_quant_bytes(KT, a.block_size, D) + (size_t) row0 * (D / PKV_QK) * 2, nval * (D / PKV_QK) * 2), k_go, k_so);
    };
    auto prefetch_v = [&](int kv0, int phys) {
        const int row0 = kv0 - blk_of(kv0) * a.block_size, nval = min(KVT, seq_len - kv0);
        const char * vt = a.v_pool + ((size_t) phys * a.n_head_kv_pool + (a.kv_div > 1 ? hk / a.kv_div : hk)) * tb_v;
        constexpr int VTL = VSt::TL;
        const int ngrp = (nval + VTL - 1) / VTL;
        vst.load(mk_rsrc(vt + (size_t) (row0 / VTL) * D * 4, ngrp * D * 4),
                 mk_rsrc(vt + tile_quant_bytes(VT, a.block_size, D) + (size_t) (row0 / VTL) * (D / PKV_QK) * VTL * 2, ngrp * (D / PKV_QK) * VTL * 2), nval, v_go, v_so);
    };
    // the first KV tile's loads go out BEFORE the Q loads: the block-table lookup and the tile's HBM latency overlap the Q fetch
    // (the K tile by LDS-DMA only where the Q strips below leave image K0 alone)
    const int kv_first = min(kv_beg, max(kv_all - 1, 0) / KVT * KVT);   // (an empty share still loads a valid, aligned tile; its loop does not run)
    constexpr bool QFREE = G * QT * D * 2 <= 2 * IMG;                   // the Q strips fit images V0 + K1
    const int phys0 = trow[blk_of(kv_first)];
    if constexpr (!KDMA || QFREE) prefetch_k(kv_first, phys0, sKb(0));
    prefetch_v(kv_first, phys0);
    int phys_nxt = trow[min(blk_of(kv_first + KVT), last_blk)];
    // ---- Q^T fragments, rounded to f16 (GGML_FP32_TO_FP16 of the reference's q_to_vec_dot for f16 K).  Lane (r, h) needs 32 bytes of
    //      row r per k-step: fetched that way every load instruction touches 64 cache lines.  Instead the wave reads its 32 rows as
    //      whole rows (two 512-byte rows per instruction), converts, and turns them through an LDS strip (the tile images are free).
    half8_t qf[QB][KS];
    {
        char * strip = smem + (QFREE ? IMG : 0) + g * (QT * D * 2);   // [32 rows][D halfs], 16-byte chunk c of row q at c ^ (q & 15); one block at a time
        static_assert(G * QT * D * 2 <= 4 * IMG, "Q strips must fit the tile images");
        const int esz = a.q_f16 ? 2 : 4;                              // bytes per element of q
        const long long qsb = (long long) a.q_stride_tok * esz;
        constexpr int LPR = D / 4, RPI = 64 / LPR, NQ = QT / RPI;   // lanes per row, rows per instruction, instructions
        const int lr = lane / LPR, lc = lane % LPR;                   // row inside the instruction's group, 4-element column
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const rsrc_t rq = mk_rsrc(reinterpret_cast<const char *>(a.q) + ((size_t) (tok0 + tile * QTW + qb * QT) * a.q_stride_tok + (size_t) head * a.q_stride_head) * esz,
                                      (long long) (n_rows - qb * QT - 1) * qsb + D * esz);   // (a block past the end of the tile: no bytes, reads as 0)
            // fused RoPE(Q): the rows' (cos, sin) pairs, same shape as the rows themselves at half the bytes (float4 = the two pairs of the lane's 4 elements)
            const rsrc_t rc = mk_rsrc(a.rope_cache ? a.rope_cache + (size_t) (tok0 + tile * QTW + qb * QT) * D : nullptr,
                                      a.rope_cache ? (long long) (n_rows - qb * QT) * D * 4 : 0);
#pragma unroll
            for (int b = 0; b < NQ; b += 8) {
                u4_t x[8], cs[8];
                if (a.q_f16) {   // (workgroup-uniform) 8 bytes = the lane's 4 halfs
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(rq, (int) (((b + i) * RPI + lr) * qsb) + lc * 8, 0, 0);
                        x[i].x = v.x; x[i].y = v.y; x[i].z = 0; x[i].w = 0;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, (int) (((b + i) * RPI + lr) * qsb) + lc * 16, 0, 0);
                }
                if (a.rope_cache) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) cs[i] = __builtin_amdgcn_raw_buffer_load_b128(rc, (((b + i) * RPI + lr) * D + lc * 4) * 4, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (b + i) * RPI + lr;
                    float x0, x1, x2, x3;
                    if (a.q_f16) {
                        const h2_t lo = as_h2(x[i].x), hi = as_h2(x[i].y);
                        x0 = (float) lo[0]; x1 = (float) lo[1]; x2 = (float) hi[0]; x3 = (float) hi[1];
                    } else {
                        x0 = __uint_as_float(x[i].x); x1 = __uint_as_float(x[i].y); x2 = __uint_as_float(x[i].z); x3 = __uint_as_float(x[i].w);
                    }
                    if (a.rope_cache) {   // the rotation of kv_store_kernel's Q plane (rot0 / rot1: every product and the sum rounded on its own)
                        const float c0 = __uint_as_float(cs[i].x), s0 = __uint_as_float(cs[i].y), c1 = __uint_as_float(cs[i].z), s1 = __uint_as_float(cs[i].w);
                        const float y0 = rot0(x0, x1, c0, s0), y1 = rot1(x0, x1, c0, s0), y2 = rot0(x2, x3, c1, s1), y3 = rot1(x2, x3, c1, s1);
                        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
                    }
                    if constexpr (KT != PKV_TYPE_F16) {   // quantized K: the reference's q_to_vec_dot is quantize_row_q8_0 (/ref/ggml/src/ggml-cpu/ggml-cpu.c:12343)
                        q8_round(x0, x1, x2, x3);         // the values its dot multiplies, q_i * f16(amax / 127); then rounded once more, to the f16 operand
                    }
                    _Float16 hv[4] = {(_Float16) x0, (_Float16) x1, (_Float16) x2, (_Float16) x3};
                    *reinterpret_cast<uint2 *>(strip + row * D * 2 + (((lc >> 1) ^ (row & 15)) & (D / 8 - 1)) * 16 + (lc & 1) * 8) = *reinterpret_cast<const uint2 *>(hv);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip is private to this wave
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[qb][ks] = *reinterpret_cast<const half8_t *>(strip + r * D * 2 + (((2 * ks + h) ^ (r & 15)) & (D / 8 - 1)) * 16);
            if (qb + 1 < QB) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // (LDS serves a wave in order)
        }
    }
    __syncthreads();                           // every wave has its Q fragments: the strips become tile images
    if constexpr (KDMA && !QFREE) prefetch_k(kv_first, phys0, sKb(0));

    if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_q = __builtin_amdgcn_s_memtime(); }
    if constexpr (!KDMA) kst.write(sKb(0), tid);
    vst.write(sVb(0), v_wo, v_u0);
    // LDS operand reads run AHEAD of the MFMAs that consume them, by hand: left to itself hipcc (256 registers at two waves per
    // SIMD) sinks every ds_read_b128 to just before its MFMA and waits lgkmcnt(0) — one LDS latency per MFMA, 32 per tile.
    // Groups of 4 (QK^T) / DT (PV) fragments, two groups in flight, pinned by sched_barriers.
    constexpr int KG = KS / 4, NGK = 2 * KG;          // QK^T: groups of 4 k-steps; group gi = (mt = gi / KG, k-steps 4 (gi % KG) ..)
    static_assert(KS % 4 == 0, "head_dim must be a multiple of 64");
    const int k_ro = r * D * 2, krow_sw = r & 15;     // this lane's K row inside a 32-row half tile
    for (int kv0 = kv_beg, it = 0; kv0 < kv_end; kv0 += KVT, ++it) {
        if constexpr (STAMP) tt = __builtin_amdgcn_s_memtime();
        const char * sK = sKb(it & 1);
        const char * sV = sVb(it & 1);
        const bool more = kv0 + KVT < kv_end;
        if constexpr (KDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the K tile has landed (LDS-DMA is counted in vmcnt)
        if (!(PKV_PREFILL_ABLATE & 1)) __syncthreads();                       // tile `it` is complete in LDS; everyone is done with tile it-1
        half8_t fb[2][4];                      // operand fragments in flight (K rows, then V^T rows)
        auto kread = [&](half8_t * dst, int gi) {
            const int mt = gi / KG, ks0 = (gi % KG) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst[j] = *reinterpret_cast<const half8_t *>(sK + mt * 32 * D * 2 + k_ro + (((2 * (ks0 + j) + h) ^ krow_sw) & (D / 8 - 1)) * 16);
        };
        auto vread = [&](half8_t * dst, int gi) {     // gi = mt * 2 + s2: chunk 2 gi + h of row d
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                dst[dt] = *reinterpret_cast<const half8_t *>(sV + (dt * 32 + r) * (KVT * 2) + (((2 * gi) ^ vsw) & 7) * 16);
        };
        kread(fb[0], 0);
        kread(fb[1], 1);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        if (more && !(PKV_PREFILL_ABLATE & (4 | 64))) {                            // HBM -> registers, in flight while this tile is multiplied
            prefetch_k(kv0 + KVT, phys_nxt, sKb((it + 1) & 1));
            prefetch_v(kv0 + KVT, phys_nxt);
        }
        if constexpr (STAMP) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_stage += t2 - tt; tt = t2; }

        // ---- S^T = K . Q^T : two 32-row kv tiles ---------------------------------------------------------------
        floatx16 sacc[QB][2];
        const floatx16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // the chain's first MFMA takes C = 0 as an inline constant
#pragma unroll
        for (int gi = 0; gi < NGK; ++gi) {
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    if (PKV_PREFILL_ABLATE & 8) { if (gi % KG == 0 && j == 0) sacc[b][gi / KG] = zero16; sacc[b][gi / KG][j] += (float) fb[gi & 1][j][0] * (float) qf[b][(gi % KG) * 4 + j][0]; }
                    else
                    sacc[b][gi / KG] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[gi & 1][j], qf[b][(gi % KG) * 4 + j], (gi % KG == 0 && j == 0) ? zero16 : sacc[b][gi / KG], 0, 0, 0);
                }
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < NGK) kread(fb[gi & 1], gi + 2);
            else              vread(fb[gi & 1], gi + 2 - NGK);   // the first two V^T groups land under the softmax arithmetic
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        // registers -> the other LDS image (last read two barriers ago), EARLY: the K rows as soon as QK^T is issued (the stores go out
        // under the MFMAs still in the pipe), the V^T chunks after the softmax arithmetic — at the end of the tile nothing stands between
        // the last PV MFMA and the barrier.  The K loads were issued first: their wait leaves the V loads in flight.
        if constexpr (!KDMA) if (more && !(PKV_PREFILL_ABLATE & (4 | 128))) kst.write(sKb((it + 1) & 1), tid);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAMP) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory"); const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_qk += t2 - tt; tt = t2; }
        // ---- causal mask (only tiles that reach the diagonal), running max; VALU kept to ~1 op per logit:
        //      the softmax scale rides in the exponent, p = exp2(c*s - c*m) = one v_fma + one v_exp -----------------------
        float mc[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
        if constexpr (SOFTCAP) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[b][mt][i] = a.softcap * tanhf(sacc[b][mt][i] * a.scale);
        } else if constexpr (MASKED) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[b][mt][i] *= a.scale;
        }
        if constexpr (MASKED) {  // s += slope * mask (:12349-12365); -inf entries skip the cell
            // accumulator registers 4a .. 4a+3 are 4 consecutive cells: 8 eight-byte loads through a descriptor over the tile's mask rows
            // (cells past the end of the last row read as 0; the cell test below removes them)
            if (a.mask_align8) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int a4 = 0; a4 < 4; ++a4) {
                        const u2_t w = __builtin_amdgcn_raw_buffer_load_b64(rs_mask, mask_vo[b], (kv0 + mt * 32 + 8 * a4) * 2, 0);
                        const _Float16 * mh = reinterpret_cast<const _Float16 *>(&w);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sacc[b][mt][4 * a4 + e] += slope * (float) mh[e];
                    }
            } else {                           // rows that are not 8-byte aligned: one cell per load, one half tile at a time
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    unsigned short mv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) mv[i] = __builtin_amdgcn_raw_buffer_load_b16(rs_mask, mask_vo[b], (kv0 + mt * 32 + (i & 3) + 8 * (i >> 2)) * 2, 0);
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[b][mt][i] += slope * (float) *reinterpret_cast<const _Float16 *>(&mv[i]);
                }
            }
        }
        if (kv0 + KVT - 1 > qpos_min) {  // workgroup-uniform
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sacc[b][mt][i] = kv0 + mt * 32 + crow(i, h) <= qpos[b] ? sacc[b][mt][i] : -INFINITY;
        }
        // max(a, b) as med3(a, b, +inf): fmaxf on an MFMA result makes hipcc quiet it first (one v_max x, x per logit)
        float tmx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!(PKV_PREFILL_ABLATE & 2))
#pragma unroll
        for (int c = 0; c < 4; ++c) {          // four independent chains
            tmx[c] = __builtin_amdgcn_fmed3f(sacc[b][0][c], sacc[b][1][c], INFINITY);
#pragma unroll
            for (int i = c + 4; i < 16; i += 4) tmx[c] = __builtin_amdgcn_fmed3f(tmx[c], __builtin_amdgcn_fmed3f(sacc[b][0][i], sacc[b][1][i], INFINITY), INFINITY);
        }
        float tmax = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(tmx[0], tmx[1], INFINITY), __builtin_amdgcn_fmed3f(tmx[2], tmx[3], INFINITY), INFINITY);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = __builtin_amdgcn_fmed3f(__uint_as_float(sw[0]), __uint_as_float(sw[1]), INFINITY);   // both halves of the row: lanes l and l ^ 32
        }
        const float m_new = __builtin_amdgcn_fmed3f(m[b], tmax, INFINITY);   // in units of the raw (or soft-capped) logit
        const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m[b] - m_new) * cexp);
        m[b] = m_new;
        if (__any(alpha != 1.0f)) {  // wave-uniform: after the first tiles the running max rarely moves
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[b][dt][i] *= alpha;
            lsum[b] *= alpha;
        }
        mc[b] = m[b] == -INFINITY ? 0.0f : m[b] * cexp;
        }
        if (more && !(PKV_PREFILL_ABLATE & (4 | 128))) vst.write(sVb((it + 1) & 1), v_wo, v_u0);
        if constexpr (STAMP) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_sm += t2 - tt; tt = t2; }
        // ---- O^T += V^T . P^T ; the row sums l are v_dot2 adds of the same f16-rounded weights PV applies.  Group gi = (mt, s2):
        //      its weights are exponentiated while group gi-1's MFMAs run -------------------------------------------------------
        auto pgroup = [&](int b, int gi) -> half8_t {
            const int mt = gi >> 1, s2 = gi & 1;
            half2_t pp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = (PKV_PREFILL_ABLATE & 2) ? sacc[b][mt][8 * s2 + 2 * j] : __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][mt][8 * s2 + 2 * j], cexp, -mc[b]));
                const float p1 = (PKV_PREFILL_ABLATE & 2) ? sacc[b][mt][8 * s2 + 2 * j + 1] : __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][mt][8 * s2 + 2 * j + 1], cexp, -mc[b]));
                pp[j] = __builtin_amdgcn_cvt_pkrtz(p0, p1);
                if (!(PKV_PREFILL_ABLATE & 2)) lsum[b] = __builtin_amdgcn_fdot2(pp[j], one2, lsum[b], false);
            }
            return *reinterpret_cast<const half8_t *>(pp);
        };
        half8_t pf[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b) pf[b] = pgroup(b, 0);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    if (PKV_PREFILL_ABLATE & 16) o[b][dt][gi] += (float) fb[gi & 1][dt][0] * (float) pf[b][0];
                    else o[b][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[gi & 1][dt], pf[b], o[b][dt], 0, 0, 0);
                }
            if (gi + 1 < 4) {
#pragma unroll
                for (int b = 0; b < QB; ++b) pf[b] = pgroup(b, gi + 1);
            }
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < 4) vread(fb[gi & 1], gi + 2);
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        // table entry of the tile after next: a scalar load issued here returns before the barrier (a wait for any LDS read with a scalar
        // load outstanding is a wait for everything — lgkmcnt(0) — so it is not issued among the operand reads)
        phys_nxt = trow[min(blk_of(kv0 + 2 * KVT), last_blk)];
        if constexpr (STAMP) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory"); const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_pv += t2 - tt; tt = t2; }
    }
    if constexpr (STAMP) t_loop = __builtin_amdgcn_s_memtime();
    // ---- normalise and write out[tok][head][d] -----------------------------------------------------------------------------
    // O^T sits with the query on the lane: stored directly, every lane would write 16-byte pieces of ITS row (32 rows per
    // instruction, 32-64 cache lines touched each).  Instead each wave transposes its tile through LDS (the K/V images are
    // free now) and stores whole rows: 32 lanes x 16 B = one 512-byte head row per half-wave.  Stores go through a buffer
    // descriptor that ends with the tile's last existing row: no per-row branches.
    {
        constexpr int DH = G <= 4 ? D : D / 2;                 // d columns staged per pass (wave strip = 32 rows x DH f32 <= 4*IMG / G)
        constexpr int CPRW = DH / 4;                           // 16-byte chunks per staged row
        static_assert(32 * DH * 4 * G <= 4 * IMG, "per-wave output strips must fit the tile images");
        float * strip = reinterpret_cast<float *>(smem) + g * 32 * DH;
        const bool  partial = a.n_split > 1;
        const bool  o16 = a.out_f16 && !partial;                // the tile's result leaves as f16 rows (the split-KV partials stay f32; their merge converts)
        const int   osz = o16 ? 2 : 4;
        const long long rowb = (long long) a.n_head * (partial ? a.n_split : 1) * D * osz;   // bytes between two tokens' rows
        __syncthreads();                                       // every wave is done reading the last K/V tile
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float l_all;
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum[qb]), __float_as_uint(lsum[qb]), false, false);
                l_all = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);   // the two lane halves hold the two halves of every tile's cells
            }
            const float inv = partial ? 1.0f : 1.0f / l_all;
            if (partial && q_ok[qb] && h == 0) {               // (m, l) of this share; an empty share leaves (-inf, 0)
                float * ml = a.part_ml + (((size_t) (tok0 + jr[qb]) * a.n_head + head) * a.n_split + sp) * 2;
                ml[0] = m[qb] == -INFINITY ? -INFINITY : m[qb] * cexp;
                ml[1] = l_all;
            }
            const size_t th0 = (size_t) (tok0 + tile * QTW + qb * QT) * a.n_head + head;   // (token, head) row of the block's first query
            const rsrc_t ro = mk_rsrc(partial ? reinterpret_cast<char *>(a.part_o + (th0 * a.n_split + sp) * D) : reinterpret_cast<char *>(a.out) + th0 * D * osz,
                                      (n_rows - qb * QT - 1) * rowb + D * osz);
#pragma unroll
            for (int pass = 0; pass < D / DH; ++pass) {
#pragma unroll
                for (int dt = pass * (DH / 32); dt < (pass + 1) * (DH / 32); ++dt)
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        const int c = ((dt * 32 - pass * DH) + crow(i, h)) / 4;   // registers i..i+3 are 4 consecutive d
                        *reinterpret_cast<float4 *>(strip + r * DH + ((c ^ (r & 7)) * 4)) =
                            make_float4(o[qb][dt][i] * inv, o[qb][dt][i + 1] * inv, o[qb][dt][i + 2] * inv, o[qb][dt][i + 3] * inv);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip is private to this wave
                __builtin_amdgcn_wave_barrier();
                constexpr int NS = 32 * CPRW / 64;             // store instructions per pass
#pragma unroll
                for (int b = 0; b < NS; b += 8) {
                    float4 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = (b + i) * 64 + lane, row = e / CPRW, c = e % CPRW;
                        v[i] = *reinterpret_cast<const float4 *>(strip + row * DH + ((c ^ (row & 7)) * 4));
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = (b + i) * 64 + lane, row = e / CPRW, c = e % CPRW;
                        if (o16) {
                            const _Float16 hv[4] = {(_Float16) v[i].x, (_Float16) v[i].y, (_Float16) v[i].z, (_Float16) v[i].w};
                            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2_t *>(hv), ro, (int) (row * rowb) + (pass * DH + c * 4) * 2, 0, 0);
                        } else {
                            u4_t w; w.x = __float_as_uint(v[i].x); w.y = __float_as_uint(v[i].y); w.z = __float_as_uint(v[i].z); w.w = __float_as_uint(v[i].w);
                            __builtin_amdgcn_raw_buffer_store_b128(w, ro, (int) (row * rowb) + (pass * DH + c * 4) * 4, 0, 0);
                        }
                    }
                }
                if (pass + 1 < D / DH || qb + 1 < QB) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
            }
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
    if (a.out_f16) {
        const _Float16 hv[4] = {(_Float16) (acc.x * inv), (_Float16) (acc.y * inv), (_Float16) (acc.z * inv), (_Float16) (acc.w * inv)};
        *reinterpret_cast<uint2 *>(reinterpret_cast<_Float16 *>(a.out) + th * D + c * 4) = *reinterpret_cast<const uint2 *>(hv);
    } else
    *reinterpret_cast<float4 *>(a.out + th * D + c * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

template <int KT, int VT, int D, int G>
int launch_one(const PrefillArgs & pa, int n_tiles, int n_seq, int qb, hipStream_t st) {
    if constexpr (KT == PKV_TYPE_F16 && VT == PKV_TYPE_F16 && D == 128 && G == 4) {
        if (pa.stamps) {
            hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false, true>), dim3(n_tiles * pa.n_split, pa.n_head_kv, n_seq), dim3(64 * G), 0, st, pa.block_table, pa.seq_lens, pa.q_start, pa);
            PKV_HIP(hipGetLastError());
            return PKV_OK;
        }
    }
    const dim3 grid(n_tiles * pa.n_split, pa.n_head_kv, n_seq), block(64 * G);
    if constexpr (G == 4 && D == 128 && KT == PKV_TYPE_F16 && VT == PKV_TYPE_F16) {   // the one instantiation kept of the 64-queries-per-wave experiment
        if (qb == 2) {   // (launch_attn_prefill asks for it only without mask / soft-cap)
            hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false, false, false, 2>), grid, block, 0, st, pa.block_table, pa.seq_lens, pa.q_start, pa);
            if (pa.n_split > 1) hipLaunchKernelGGL((prefill_combine_kernel<D>), dim3(pa.n_head, pa.n_tok_total), dim3(D / 4), 0, st, pa);
            PKV_HIP(hipGetLastError());
            return PKV_OK;
        }
    }
    if (pa.mask) {
        if (pa.softcap != 0.0f) hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, true, false, true>), grid, block, 0, st, pa.block_table, pa.seq_lens, pa.q_start, pa);
        else                    hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false, false, true>), grid, block, 0, st, pa.block_table, pa.seq_lens, pa.q_start, pa);
    } else {
        if (pa.softcap != 0.0f) hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, true>), grid, block, 0, st, pa.block_table, pa.seq_lens, pa.q_start, pa);
        else                    hipLaunchKernelGGL((attn_prefill_kernel<KT, VT, D, G, false>), grid, block, 0, st, pa.block_table, pa.seq_lens, pa.q_start, pa);
    }
    if (pa.n_split > 1) hipLaunchKernelGGL((prefill_combine_kernel<D>), dim3(pa.n_head, pa.n_tok_total), dim3(D / 4), 0, st, pa);
    PKV_HIP(hipGetLastError());
    return PKV_OK;
}
template <int KT, int VT, int D>
int launch_g(const PrefillArgs & pa, int G, int n_tiles, int n_seq, int qb, hipStream_t st) {
    switch (G) {
        case 1: return launch_one<KT, VT, D, 1>(pa, n_tiles, n_seq, qb, st);
        case 2: return launch_one<KT, VT, D, 2>(pa, n_tiles, n_seq, qb, st);
        case 4: return launch_one<KT, VT, D, 4>(pa, n_tiles, n_seq, qb, st);
        case 8: return launch_one<KT, VT, D, 8>(pa, n_tiles, n_seq, qb, st);
    }
    set_error("pkv_attn_prefill: unsupported gqa ratio %d (supported 1,2,4,8)", G);
    return PKV_ERR_UNSUPPORTED;
}
template <int KT, int VT>
int launch_d(const PrefillArgs & pa, int D, int G, int n_tiles, int n_seq, int qb, hipStream_t st) {
    switch (D) {
        case 64:  return launch_g<KT, VT, 64>(pa, G, n_tiles, n_seq, qb, st);
        case 128: return launch_g<KT, VT, 128>(pa, G, n_tiles, n_seq, qb, st);
    }
    set_error("pkv_attn_prefill: unsupported head_dim %d (supported 64,128)", D);
    return PKV_ERR_UNSUPPORTED;
}
template <int KT>
int launch_v(const PrefillArgs & pa, int VT, int D, int G, int n_tiles, int n_seq, int qb, hipStream_t st) {
    switch (VT) {
        case PKV_TYPE_F16:  return launch_d<KT, PKV_TYPE_F16>(pa, D, G, n_tiles, n_seq, qb, st);
        case PKV_TYPE_Q8_0: return launch_d<KT, PKV_TYPE_Q8_0>(pa, D, G, n_tiles, n_seq, qb, st);
        case PKV_TYPE_Q4_0: return launch_d<KT, PKV_TYPE_Q4_0>(pa, D, G, n_tiles, n_seq, qb, st);
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

int launch_attn_prefill(const pkv_attn_args & a, hipStream_t st, const float * rope_cache) {
    PKV_REQUIRE(a.q && a.out && a.k_pool && a.v_pool && a.block_table && a.seq_lens && a.q_start, "pkv_attn_prefill: null pointer");
    PKV_REQUIRE(a.n_seq > 0 && a.n_tok > 0 && a.max_q_per_seq > 0, "pkv_attn_prefill: bad token counts");
    PKV_REQUIRE(a.n_head > 0 && a.n_head_kv > 0 && a.n_head % a.n_head_kv == 0, "pkv_attn_prefill: bad head counts");
    PKV_REQUIRE(a.block_size >= KVT && a.block_size % KVT == 0, "pkv_attn_prefill: block_size must be a multiple of %d", KVT);
    PKV_REQUIRE(kv_type_ok(a.type_k) && kv_type_ok(a.type_v), "pkv_attn_prefill: KV types must be f16/q8_0/q4_0");
    PKV_REQUIRE(a.max_seq_len > 0 && a.max_seq_len <= a.max_blocks * a.block_size, "pkv_attn_prefill: max_seq_len out of range");
    PKV_REQUIRE((a.q_stride_head % 4) == 0 && (a.q_stride_tok % 4) == 0 && ((uintptr_t) a.q % 16) == 0 && ((uintptr_t) a.out % 16) == 0,
                "pkv_attn_prefill: q/out must be 16-byte aligned");
    PKV_REQUIRE(!(a.flags & (PKV_ATTN_Q_F16 | PKV_ATTN_OUT_F16)) || !g_prefill_stamps, "pkv_attn_prefill: the stamp build takes f32 q / out");
    if (a.slot_pos) { set_error("pkv_attn_prefill: irregular sequences (slot_pos) are not supported yet; defrag first"); return PKV_ERR_UNSUPPORTED; }
    PrefillArgs pa;
    pa.q = a.q; pa.q_stride_tok = a.q_stride_tok; pa.q_stride_head = a.q_stride_head; pa.out = a.out;
    pa.k_pool = (const char *) a.k_pool; pa.v_pool = (const char *) a.v_pool;
    pa.block_table = a.block_table; pa.seq_lens = a.seq_lens; pa.q_start = a.q_start;
    pa.n_head = a.n_head; pa.n_head_kv = a.n_head_kv; pa.block_size = a.block_size; pa.max_blocks = a.max_blocks;
    pa.kv_div = 1; pa.n_head_kv_pool = a.n_head_kv;
    if (a.n_head / a.n_head_kv == 16 || a.n_head / a.n_head_kv == 32) { pa.kv_div = a.n_head / a.n_head_kv / 8; pa.n_head_kv = a.n_head_kv * pa.kv_div; }
    pa.bs_shift = -1;
    for (int sh = 0; sh < 31; ++sh) if ((1 << sh) == a.block_size) pa.bs_shift = sh;
    pa.softcap = a.logit_softcap;
    pa.stamps = (unsigned long long *) g_prefill_stamps;
    pa.mask = (const _Float16 *) a.mask; pa.mask_stride = a.mask_stride; pa.max_bias = a.max_bias;
    pa.mask_align8 = ((uintptr_t) a.mask % 8) == 0 && (a.mask_stride % 4) == 0;
    pa.rope_cache = rope_cache;
    PKV_REQUIRE(((uintptr_t) rope_cache % 16) == 0, "pkv_attn_prefill_step: rope_cache must be 16-byte aligned");
    pa.causal = (a.flags & PKV_ATTN_NONCAUSAL) ? 0 : 1;
    pa.q_f16 = (a.flags & PKV_ATTN_Q_F16) ? 1 : 0; pa.out_f16 = (a.flags & PKV_ATTN_OUT_F16) ? 1 : 0;
    PKV_REQUIRE(a.max_bias == 0.0f || a.mask, "pkv_attn_prefill: max_bias (ALiBi) needs a mask");
    pa.scale   = a.logit_softcap != 0.0f ? a.scale / a.logit_softcap : a.scale;
    const int G = a.n_head / pa.n_head_kv;
    // 64 queries per wave (QB = 2: one workgroup per CU, one wave per SIMD) is built, bit-compatible and SLOWER than two 32-query waves per
    // SIMD (617 vs 644 TFLOP/s at pp4096, 295 vs 334 at 32 x pp512): a single in-order wave per SIMD does not overlap its
    // softmax arithmetic with its own MFMAs.  Kept as an opt-in experiment for the headline instantiation (PKV_PREFILL_QB=2).
    int qb = 1;
    if (const char * e = getenv("PKV_PREFILL_QB"))
        if (atoi(e) == 2 && G == 4 && a.head_dim == 128 && a.type_k == PKV_TYPE_F16 && a.type_v == PKV_TYPE_F16 && !a.mask && a.logit_softcap == 0.0f && !g_prefill_stamps) qb = 2;
    const int n_tiles = (a.max_q_per_seq + QT * qb - 1) / (QT * qb);
    // split-KV when the launch would leave most of the 256 CUs (2 workgroups each) idle and the past is long enough to share
    pa.n_split = 1; pa.n_tok_total = a.n_tok; pa.part_o = nullptr; pa.part_ml = nullptr;
    const long long n_wg = (long long) n_tiles * pa.n_head_kv * a.n_seq;
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
#ifdef PKV_PREFILL_DEV   // development builds: the headline instantiation only (the full set takes minutes to compile)
    if (a.type_k == PKV_TYPE_F16 && a.type_v == PKV_TYPE_F16 && a.head_dim == 128 && G == 4) return launch_one<PKV_TYPE_F16, PKV_TYPE_F16, 128, 4>(pa, n_tiles, a.n_seq, qb, st);
    set_error("pkv_attn_prefill: development build"); return PKV_ERR_UNSUPPORTED;
#else
    switch (a.type_k) {
        case PKV_TYPE_F16:  return launch_v<PKV_TYPE_F16>(pa, a.type_v, a.head_dim, G, n_tiles, a.n_seq, qb, st);
        case PKV_TYPE_Q8_0: return launch_v<PKV_TYPE_Q8_0>(pa, a.type_v, a.head_dim, G, n_tiles, a.n_seq, qb, st);
        case PKV_TYPE_Q4_0: return launch_v<PKV_TYPE_Q4_0>(pa, a.type_v, a.head_dim, G, n_tiles, a.n_seq, qb, st);
    }
    return PKV_ERR_UNSUPPORTED;
#endif
}

}  // namespace pkv
