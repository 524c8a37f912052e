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

namespace pkv {
namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int WG_TOK = 128;  // KV tokens per workgroup (== one block at block_size 128)
constexpr int NWAVE  = 4;
constexpr int TPW    = WG_TOK / NWAVE;  // tokens per wave

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
    float *         part_o;   // [n_seq][n_head][n_splits][D]
    float *         part_ml;  // [n_seq][n_head][n_splits][2]
    float *         out;      // [n_seq][n_head][D]
    int   n_head, n_head_kv, block_size, max_blocks, n_splits;
    float scale, softcap;
};

// Q fragment for one lane: f16 path = E halfs per head; quantized path = E int8 per head + f16-rounded block scale
template <int KT, int D, int G> struct QFrag;
template <int D, int G> struct QFrag<PKV_TYPE_F16, D, G> {
    half2_t h[G][4];
    __device__ __forceinline__ void load(const float * q, int64_t stride_head, int kc) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 a = *reinterpret_cast<const float4 *>(q + g * stride_head + kc * 8);
            const float4 b = *reinterpret_cast<const float4 *>(q + g * stride_head + kc * 8 + 4);
            h[g][0] = half2_t{(_Float16) a.x, (_Float16) a.y};  // RN-even == GGML_FP32_TO_FP16
            h[g][1] = half2_t{(_Float16) a.z, (_Float16) a.w};
            h[g][2] = half2_t{(_Float16) b.x, (_Float16) b.y};
            h[g][3] = half2_t{(_Float16) b.z, (_Float16) b.w};
        }
    }
};
// quantize_row_q8_0 as the reference's CPU backend does it for Q (AVX2 branch,
// /ref/ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862): d = amax/127 (stored as f16), id = 127/amax, round-half-even.
template <int KT, int D, int G> struct QFragQ {
    static constexpr int E  = KTr<KT, D>::E;   // 16 (q8_0 K) or 32 (q4_0 K) elements per lane
    static constexpr int NW = E / 4;
    int   qs[G][NW];
    float d[G];
    int   qsum[G];  // sum of the block's int8 (q4_0 K needs it for the -8 offset)
    __device__ __forceinline__ void load(const float * q, int64_t stride_head, int kc) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float x[E];
#pragma unroll
            for (int j = 0; j < E / 4; ++j) {
                const float4 a = *reinterpret_cast<const float4 *>(q + g * stride_head + kc * E + j * 4);
                x[j * 4 + 0] = a.x; x[j * 4 + 1] = a.y; x[j * 4 + 2] = a.z; x[j * 4 + 3] = a.w;
            }
            float amax = 0.0f;
#pragma unroll
            for (int j = 0; j < E; ++j) amax = fmaxf(amax, fabsf(x[j]));
            if (E == 16) amax = fmaxf(amax, dpp_mov<0xB1>(amax));  // the 32-block spans two lanes
            const float dd = amax / 127.0f;
            const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
            d[g] = (float) (_Float16) dd;
            int sum = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                int pk = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int v = (int) __builtin_rintf(x[w * 4 + b] * id);
                    sum += v;
                    pk |= (v & 0xFF) << (8 * b);
                }
                qs[g][w] = pk;
            }
            qsum[g] = sum;  // only read by the q4_0-K dot, where the lane holds the whole 32-block
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

// V side: a lane owns 8 consecutive d of TL tokens per 16-byte load; 16 lanes (D=128) cover the d range.
//   f16 : TL = 1, tile = half[tok][D]
//   q8_0: TL = 2, tile = int8[tok/2][D/8][2][8]  then half d[tok/2][D/32][2]
//   q4_0: TL = 4, tile = nib [tok/4][D/8][4][8]  then half d[tok/4][D/32][4]   (byte m = d 2m | d 2m+1 << 4)
template <int T> struct VTr;
template <> struct VTr<PKV_TYPE_F16>  { static constexpr int TL = 1; };
template <> struct VTr<PKV_TYPE_Q8_0> { static constexpr int TL = 2; };
template <> struct VTr<PKV_TYPE_Q4_0> { static constexpr int TL = 4; };

template <int KT, int VT, int D, int G>
__global__ void __launch_bounds__(256) attn_decode_kernel(const DecodeArgs a) {
    using KT_ = KTr<KT, D>;
    constexpr int K_LPR = KT_::LPR, K_RPL = 64 / K_LPR, K_NL = TPW / K_RPL;
    constexpr int V_LPR = D / 8, V_TL = VTr<VT>::TL, V_RPL = (64 / V_LPR) * V_TL, V_NL = TPW / V_RPL;
    static_assert(K_NL >= 1 && V_NL >= 1, "tile too small for this type / head size");

    __shared__ float sS[WG_TOK][G];        // logits, then probabilities
    __shared__ float sRed[2][NWAVE][G];    // per-wave max / sum
    __shared__ float sO[NWAVE][G][D];      // per-wave partial outputs

    const int split = blockIdx.x, hk = blockIdx.y, s = blockIdx.z;
    const int seq_len = a.seq_lens[s];
    const int base    = split * WG_TOK;
    if (base >= seq_len) return;  // workgroup-uniform

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lb = base / a.block_size, boff = base % a.block_size;
    const int phys = a.block_table[(size_t) s * a.max_blocks + lb];
    const size_t tb_k = tile_bytes(KT, a.block_size, D), tb_v = tile_bytes(VT, a.block_size, D);
    const char * ktile = a.k_pool + ((size_t) phys * a.n_head_kv + hk) * tb_k;
    const char * vtile = a.v_pool + ((size_t) phys * a.n_head_kv + hk) * tb_v;
    const int n_valid = min(WG_TOK, seq_len - base);

    // ---- issue every K and V load of this lane before anything else ----------------------------------
    const int kr = lane / K_LPR, kc = lane % K_LPR;
    uint4 kreg[K_NL];
    float ksc[K_NL];
#pragma unroll
    for (int i = 0; i < K_NL; ++i) {
        const int tw = wave * TPW + i * K_RPL + kr;
        kreg[i] = make_uint4(0, 0, 0, 0);
        ksc[i]  = 0.0f;
        if (tw < n_valid) {
            const int row = boff + tw;
            kreg[i] = *reinterpret_cast<const uint4 *>(ktile + ((size_t) row * K_LPR + kc) * 16);
            if constexpr (KT != PKV_TYPE_F16) {
                const int blk = kc * KT_::E / PKV_QK;
                ksc[i] = (float) *reinterpret_cast<const _Float16 *>(
                    ktile + tile_quant_bytes(KT, a.block_size, D) + ((size_t) row * (D / PKV_QK) + blk) * 2);
            }
        }
    }
    const int vr = lane / V_LPR, vc = lane % V_LPR;
    uint4 vreg[V_NL];
    float vsc[V_NL][V_TL];
#pragma unroll
    for (int i = 0; i < V_NL; ++i) {
        const int tw = wave * TPW + i * V_RPL + vr * V_TL;  // first of the TL tokens of this lane-load
        vreg[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < V_TL; ++t) vsc[i][t] = 0.0f;
        if (tw < n_valid) {
            const int grp = (boff + tw) / V_TL;  // token group index inside the tile
            vreg[i] = *reinterpret_cast<const uint4 *>(vtile + ((size_t) grp * V_LPR + vc) * 16);
            if constexpr (VT != PKV_TYPE_F16) {
                const _Float16 * sp = reinterpret_cast<const _Float16 *>(
                    vtile + tile_quant_bytes(VT, a.block_size, D) + ((size_t) grp * (D / PKV_QK) + vc / 4) * V_TL * 2);
#pragma unroll
                for (int t = 0; t < V_TL; ++t) vsc[i][t] = tw + t < n_valid ? (float) sp[t] : 0.0f;
            }
        }
    }

    // ---- Q fragment (L2-resident, 16 KiB for the whole launch) ----------------------------------------
    QFrag<KT, D, G> qf;
    qf.load(a.q + (size_t) s * a.q_stride_tok + (size_t) hk * G * a.q_stride_head, a.q_stride_head, kc);

    // ---- QK^T: lane partials -> DPP all-reduce over the K_LPR lanes of a row -> logits in LDS -----------
    const int32_t * spos = a.slot_pos ? a.slot_pos + (size_t) s * a.max_blocks * a.block_size + base : nullptr;
    const int qpos = a.slot_pos ? a.q_pos[s] : 0;
#pragma unroll
    for (int i = 0; i < K_NL; ++i) {
        float sv[G];
        k_dot<KT, D, G>(kreg[i], ksc[i], qf, sv);
#pragma unroll
        for (int g = 0; g < G; ++g) sv[g] = group_sum<K_LPR>(sv[g]);
        if (kc == 0) {
            const int tw = wave * TPW + i * K_RPL + kr;
            bool ok = tw < n_valid;
            if (spos && ok) { const int p = spos[tw]; ok = p >= 0 && p <= qpos; }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float v = sv[g] * a.scale;                                  // :12359
                if (a.softcap != 0.0f) v = a.softcap * tanhf(v);            // :12361-12363 (scale pre-divided on host)
                sS[tw][g] = ok ? v : -INFINITY;                             // mask == -inf -> skipped cell (:12349)
            }
        }
    }
    __syncthreads();

    // ---- softmax over the workgroup's 128 tokens: thread (tok, head pair) ---------------------------------
    {
        constexpr int GP = (G + 1) / 2;         // heads per thread
        const int tok = tid & (WG_TOK - 1), gh = tid >> 7;
        float sv[GP], m[GP];
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            const int g = gh * GP + j;
            sv[j] = g < G ? sS[tok][g] : -INFINITY;
            m[j]  = wave_max(sv[j]);
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < GP; ++j) if (gh * GP + j < G) sRed[0][wave][gh * GP + j] = m[j];
        }
        __syncthreads();
        // waves 0,1 hold head group 0, waves 2,3 head group 1 (tok spans two waves)
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            const int g = gh * GP + j;
            if (g < G) {
                const float mm = fmaxf(sRed[0][gh * 2][g], sRed[0][gh * 2 + 1][g]);
                const float p  = mm == -INFINITY ? 0.0f : __expf(sv[j] - mm);
                sS[tok][g]     = p;
                const float l  = wave_sum(p);
                if (lane == 0) { sRed[1][wave][g] = l; }
                m[j] = mm;
            }
        }
        __syncthreads();
        if (tid < G) {  // publish (m, l) of this split
            const int g = tid, ghh = g / GP;
            const float mm = fmaxf(sRed[0][ghh * 2][g], sRed[0][ghh * 2 + 1][g]);
            const float l  = sRed[1][ghh * 2][g] + sRed[1][ghh * 2 + 1][g];
            float * ml = a.part_ml + (((size_t) s * a.n_head + hk * G + g) * a.n_splits + split) * 2;
            ml[0] = mm;
            ml[1] = l;
        }
    }

    // ---- PV: f32 accumulators, lane owns 8 d of its tokens -------------------------------------------------
    float acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.0f;

#pragma unroll
    for (int i = 0; i < V_NL; ++i) {
        const int tw = wave * TPW + i * V_RPL + vr * V_TL;
#pragma unroll
        for (int t = 0; t < V_TL; ++t) {
            float p[G];
#pragma unroll
            for (int g = 0; g < G; ++g) p[g] = sS[tw + t][g];
            float vf[8];
            if constexpr (VT == PKV_TYPE_F16) {
                const _Float16 * vh = reinterpret_cast<const _Float16 *>(&vreg[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = (float) vh[j];
            } else if constexpr (VT == PKV_TYPE_Q8_0) {
                const int * vw = reinterpret_cast<const int *>(&vreg[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = (float) (int) (signed char) (vw[t * 2 + j / 4] >> (8 * (j & 3)));
#pragma unroll
                for (int g = 0; g < G; ++g) p[g] *= vsc[i][t];
            } else {
                const uint32_t vw = reinterpret_cast<const uint32_t *>(&vreg[i])[t];
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = (float) ((int) ((vw >> (4 * j)) & 0xF) - 8);
#pragma unroll
                for (int g = 0; g < G; ++g) p[g] *= vsc[i][t];
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[g][j] = fmaf(p[g], vf[j], acc[g][j]);
        }
    }
    // sum the lane-rows of the wave (lanes with equal vc), then the waves through LDS
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[g][j];
            if (V_LPR <= 8)  v += __shfl_xor(v, 8);
            if (V_LPR <= 16) v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[g][j] = v;
        }
    if (vr == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            *reinterpret_cast<float4 *>(&sO[wave][g][vc * 8])     = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
            *reinterpret_cast<float4 *>(&sO[wave][g][vc * 8 + 4]) = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
        }
    }
    __syncthreads();
    for (int e = tid; e < G * D; e += 256) {
        const int g = e / D, d = e % D;
        const float v = sO[0][g][d] + sO[1][g][d] + sO[2][g][d] + sO[3][g][d];
        a.part_o[(((size_t) s * a.n_head + hk * G + g) * a.n_splits + split) * D + d] = v;
    }
}

// merge the per-block partials: out = sum_j O_j e^{m_j - M} / sum_j l_j e^{m_j - M}
template <int D>
__global__ void __launch_bounds__(D) attn_combine_kernel(const DecodeArgs a) {
    const int h = blockIdx.x, s = blockIdx.y, d = threadIdx.x;
    const int seq_len = a.seq_lens[s];
    const int ns = (seq_len + WG_TOK - 1) / WG_TOK;
    float * o = a.out + ((size_t) s * a.n_head + h) * D;
    if (ns <= 0) { o[d] = 0.0f; return; }
    const float * ml = a.part_ml + ((size_t) s * a.n_head + h) * a.n_splits * 2;
    const float * po = a.part_o + ((size_t) s * a.n_head + h) * a.n_splits * D;
    float M = -INFINITY;
    for (int j = 0; j < ns; ++j) M = fmaxf(M, ml[j * 2]);
    float L = 0.0f, acc = 0.0f;
    for (int j = 0; j < ns; ++j) {
        const float w = M == -INFINITY ? 0.0f : __expf(ml[j * 2] - M);
        L += ml[j * 2 + 1] * w;
        acc = fmaf(po[(size_t) j * D + d], w, acc);
    }
    o[d] = acc / L;  // all cells masked -> 0/0 = NaN, like the reference's 1/S with S == 0
}

template <int KT, int VT, int D, int G>
int launch_one(const DecodeArgs & da, int n_seq, hipStream_t st) {
    dim3 grid(da.n_splits, da.n_head_kv, n_seq);
    hipLaunchKernelGGL((attn_decode_kernel<KT, VT, D, G>), grid, dim3(256), 0, st, da);
    if (da.out) hipLaunchKernelGGL((attn_combine_kernel<D>), dim3(da.n_head, n_seq), dim3(D), 0, st, da);
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

size_t decode_workspace_bytes(int n_tok, int n_head, int D, int max_seq_len) {
    const size_t ns = (size_t) (max_seq_len + WG_TOK - 1) / WG_TOK;
    return (size_t) n_tok * n_head * (ns ? ns : 1) * (D + 2) * sizeof(float);
}

int launch_attn_decode(const pkv_attn_args & a, hipStream_t st) {
    PKV_REQUIRE(a.q && (a.out || (a.flags & PKV_ATTN_NO_COMBINE)) && a.k_pool && a.v_pool && a.block_table && a.seq_lens, "pkv_attn_decode: null pointer");
    PKV_REQUIRE(a.q_start == nullptr && a.n_tok == a.n_seq, "pkv_attn_decode: one query token per sequence (use pkv_attn_prefill)");
    PKV_REQUIRE(a.n_seq > 0 && a.n_head > 0 && a.n_head_kv > 0 && a.n_head % a.n_head_kv == 0, "pkv_attn_decode: bad head counts");
    PKV_REQUIRE(a.block_size > 0 && a.block_size % WG_TOK == 0, "pkv_attn_decode: block_size must be a multiple of %d", WG_TOK);
    PKV_REQUIRE(kv_type_ok(a.type_k) && kv_type_ok(a.type_v), "pkv_attn_decode: KV types must be f16/q8_0/q4_0");
    PKV_REQUIRE(a.max_seq_len > 0 && a.max_seq_len <= a.max_blocks * a.block_size, "pkv_attn_decode: max_seq_len out of range");
    PKV_REQUIRE((a.q_stride_head % 4) == 0 && (a.q_stride_tok % 4) == 0 && ((uintptr_t) a.q % 16) == 0, "pkv_attn_decode: q must be 16-byte aligned");
    PKV_REQUIRE(a.slot_pos == nullptr || a.q_pos != nullptr, "pkv_attn_decode: slot_pos needs q_pos");
    const size_t need = decode_workspace_bytes(a.n_tok, a.n_head, a.head_dim, a.max_seq_len);
    PKV_REQUIRE(a.workspace && a.workspace_bytes >= need, "pkv_attn_decode: workspace too small (%zu < %zu)", a.workspace_bytes, need);

    DecodeArgs da;
    da.q = a.q; da.q_stride_tok = a.q_stride_tok; da.q_stride_head = a.q_stride_head;
    da.k_pool = (const char *) a.k_pool; da.v_pool = (const char *) a.v_pool;
    da.block_table = a.block_table; da.seq_lens = a.seq_lens; da.slot_pos = a.slot_pos; da.q_pos = a.q_pos;
    da.n_head = a.n_head; da.n_head_kv = a.n_head_kv; da.block_size = a.block_size; da.max_blocks = a.max_blocks;
    da.n_splits = (a.max_seq_len + WG_TOK - 1) / WG_TOK;
    da.part_o  = (float *) a.workspace;
    da.part_ml = da.part_o + (size_t) a.n_tok * a.n_head * da.n_splits * a.head_dim;
    da.out = (a.flags & PKV_ATTN_NO_COMBINE) ? nullptr : a.out;
    da.softcap = a.logit_softcap;
    da.scale   = a.logit_softcap != 0.0f ? a.scale / a.logit_softcap : a.scale;  // :12290-12292
    const int G = a.n_head / a.n_head_kv;
    switch (a.type_k) {
        case PKV_TYPE_F16:  return launch_v<PKV_TYPE_F16>(da, a.type_v, a.head_dim, G, a.n_seq, st);
        case PKV_TYPE_Q8_0: return launch_v<PKV_TYPE_Q8_0>(da, a.type_v, a.head_dim, G, a.n_seq, st);
        case PKV_TYPE_Q4_0: return launch_v<PKV_TYPE_Q4_0>(da, a.type_v, a.head_dim, G, a.n_seq, st);
    }
    return PKV_ERR_UNSUPPORTED;
}

}  // namespace pkv
