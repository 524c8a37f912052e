// pkv_rowcodec.h — device-side row helpers shared by the KV-store kernels (pkv_kv_ops.hip) and the fused decode step
// (pkv_attn_decode.hip): RoPE of one row held by a wave, the f16/q8_0/q4_0 row codecs and pool-tile row access.
// One 64-lane wave handles one row; lane i owns elements (2i, 2i+1).  gfx950 only.
//
// Every function that rounds spells its own `fp contract(off)`: the fused and the separate launches must produce the
// same bits whatever the including file's default is.
#pragma once
#include "pkv_common.h"

#include <math.h>

namespace pkv {
namespace {

template <int CTRL> __device__ __forceinline__ float dppf(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float x) {  // all-reduce max over the 16 lanes of a DPP row
    x = fmaxf(x, dppf<0xB1>(x));
    x = fmaxf(x, dppf<0x4E>(x));
    x = fmaxf(x, dppf<0x141>(x));
    x = fmaxf(x, dppf<0x140>(x));
    return x;
}

// ---- cosf / sinf as the host's libm computes them ------------------------------------------------------
// The reference's RoPE calls cosf / sinf of the C library (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10578-10593).  On the hosts this path
// is compared with (glibc >= 2.28, x86-64) those are the double-precision-internal routines of the ARM optimized-routines
// project: argument reduction by pi/2 in f64 (a 192-bit 4/pi table for |x| >= 120), a degree-7 / degree-8 polynomial in
// f64, ONE rounding to f32.  The device's own cosf / sinf differ from that by an ulp now and then, and so does the
// correctly rounded value (~1 % of RoPE angles: the polynomial is only good to 0.56 ulp) — either way a K element
// that sits on a rounding boundary of the cache type would land one quantum away from the reference's.  So the same
// published algorithm is restated here, operation by operation in IEEE f64 (no contraction): bit-identical to glibc on
// 4e7 sampled arguments of both signs (oracle/check_trig.c).  Constants: glibc's __sincosf_table / __inv_pio4.
struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };
__device__ __forceinline__ float sincos_poly(double x, double x2, bool flip, int n) {
#pragma clang fp contract(off)
    const double sg = flip ? -1.0 : 1.0;   // table 1 = table 0 with the cosine coefficients negated
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
        const double x7 = x3 * x2;
        const double s  = x + x3 * -0x1.555545995a603p-3;
        return (float) (s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = sg * -0x1.6c087e89a359dp-10 + x2 * (sg * 0x1.99343027bf8c3p-16);
    const double c1 = sg * 0x1p0 + x2 * (sg * -0x1.ffffffd0c621cp-2);
    const double x6 = x4 * x2;
    const double c  = c1 + x4 * (sg * 0x1.55553e1068f19p-5);
    return (float) (c + x6 * c2);
}
__device__ __forceinline__ uint32_t abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ff; }
// x reduced to [-pi/4, pi/4] and its quadrant n; *sign_add = the sign bit that the large-argument path folds into the quadrant
__device__ __forceinline__ double sincos_reduce(float y, int * np, int * sign_add) {
#pragma clang fp contract(off)
    const double x = (double) y;
    *sign_add = 0;
    if (abstop12(y) < abstop12(120.0f)) {
        const double r = x * 0x1.45F306DC9C883p+23;            // 2/pi * 2^24: the quadrant ends up in bits 24..31
        const int n = ((int32_t) r + 0x800000) >> 24;
        *np = n;
        return x - n * 0x1.921FB54442D18p0;
    }
    const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                   0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                   0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    uint32_t xi = __float_as_uint(y);
    *sign_add = (int) (xi >> 31);
    const uint32_t * arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = (uint32_t) (xi * arr[0]);
    const uint64_t res1 = (uint64_t) xi * arr[4], res2 = (uint64_t) xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int) n;
    return (double) (int64_t) res0 * 0x1.921FB54442D18p-62;
}
__device__ __forceinline__ float libm_sinf(float y) {
#pragma clang fp contract(off)
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        const double x = (double) y;
        return sincos_poly(x, x * x, false, 0);
    }
    if (!(abstop12(y) < abstop12(__builtin_inff()))) return y - y;
    int n, sg;
    const double x = sincos_reduce(y, &n, &sg);
    const int q = n + sg;
    const double s = (q & 1) != ((q >> 1) & 1) ? -1.0 : 1.0;      // sign[q & 3] = {1, -1, -1, 1}
    return sincos_poly(x * s, x * x, (q & 2) != 0, n);
}
__device__ __forceinline__ float libm_cosf(float y) {
#pragma clang fp contract(off)
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        const double x = (double) y;
        return sincos_poly(x, x * x, false, 1);
    }
    if (!(abstop12(y) < abstop12(__builtin_inff()))) return y - y;
    int n, sg;
    const double x = sincos_reduce(y, &n, &sg);
    const int q = n + sg;
    const double s = (q & 1) != ((q >> 1) & 1) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, (q & 2) != 0, n ^ 1);
}

// ---- RoPE --------------------------------------------------------------------------------------------
// (cos, sin) * mscale for rotation pair `ip` at position `pos`:
// ggml_rope_cache_init + rope_yarn (/ref/ggml/src/ggml-cpu/ggml-cpu.c:10578-10609).  theta is formed by the same
// sequence of f32 multiplications as the reference (theta *= theta_scale per pair), not by powf.
__device__ __forceinline__ float2 rope_cs(int pos, int ip, const RopeDev & r, float sin_sign) {
#pragma clang fp contract(off)
    float theta = (float) pos;
    for (int k = 0; k < ip; ++k) theta *= r.theta_scale;
    const float ff           = r.freq_factors ? r.freq_factors[ip] : 1.0f;
    const float theta_extrap = theta / ff;
    const float theta_interp = r.freq_scale * theta_extrap;
    float th = theta_interp, mscale = r.attn_factor;
    if (r.ext_factor != 0.0f) {
        const float y    = ((float) ip - r.corr0) / fmaxf(0.001f, r.corr1 - r.corr0);
        const float ramp = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * r.ext_factor;
        th = theta_interp * (1.0f - ramp) + theta_extrap * ramp;
        mscale *= r.yarn_mscale;   // 1.0f + 0.1f * logf(1.0f / freq_scale), formed on the host by the host's logf (make_rope)
    }
    return make_float2(libm_cosf(th) * mscale, libm_sinf(th) * mscale * sin_sign);
}

// one rotation, x0*cos - x1*sin / x0*sin + x1*cos (:10818-10819), every product and the sum rounded on its own: the pinned
// build of the reference (oracle/Makefile: ISO C mode, -std=c11, so GCC does NOT contract a*b + c*d into an FMA) rounds this
// way, and every kernel and call site here must round identically.  (r01 spelled an FMA here: 20 % of the rotated
// elements were one ulp away from the reference's.)
__device__ __forceinline__ float rot0(float x0, float x1, float c, float sn) {
#pragma clang fp contract(off)
    const float a = x0 * c, b = x1 * sn;
    return a - b;
}
__device__ __forceinline__ float rot1(float x0, float x1, float c, float sn) {
#pragma clang fp contract(off)
    const float a = x0 * sn, b = x1 * c;
    return a + b;
}

// rotate the wave's row in registers: lane i holds (x0, x1) = elements (2i, 2i+1) of a row of D elements
template <int D>
__device__ __forceinline__ float2 rope_row(float2 x, int lane, int pos, const RopeDev & r, float sin_sign) {
    const int e0 = 2 * (lane % (D / 2));   // D = 64: the idle upper half-wave aliases the lower one (in-bounds table reads)
    if ((r.mode & 2) == 0) {  // NORM: pairs (2i, 2i+1)
        if (e0 < r.n_dims) {
            const float2 cs = rope_cs(pos, e0 / 2, r, sin_sign);
            return make_float2(rot0(x.x, x.y, cs.x, cs.y), rot1(x.x, x.y, cs.x, cs.y));
        }
        return x;
    }
    // NEOX with n_dims == D: pairs (ic, ic + D/2); the partner elements live in lane ^ (D/4)
    constexpr int HL = D / 4;
    const float px = __shfl_xor(x.x, HL), py = __shfl_xor(x.y, HL);
    const bool  lo = (lane % (D / 2)) < HL;
    const int   ic = lo ? e0 : e0 - D / 2;
    const float2 c0 = rope_cs(pos, ic, r, sin_sign), c1 = rope_cs(pos, ic + 1, r, sin_sign);
    return lo ? make_float2(rot0(x.x, px, c0.x, c0.y), rot0(x.y, py, c1.x, c1.y))
              : make_float2(rot1(px, x.x, c0.x, c0.y), rot1(py, x.y, c1.x, c1.y));
}

// The same rotation from a precomputed (cos, sin) row (pkv_rope_cache = ggml_rope_cache_init, :10595: one row per token,
// shared by every head and every layer of the step).  `t` = the two consecutive table entries starting at
// rope_cache_base(): NORM pairs use one of them, NEOX lanes use both (pairs ic and ic + 1).
template <int D> __device__ __forceinline__ int rope_cache_base(int lane, const RopeDev & r) {
    const int l = lane % (D / 2);
    if ((r.mode & 2) == 0) return l & ~1;
    return l < D / 4 ? 2 * l : 2 * l - D / 2;
}
template <int D>
__device__ __forceinline__ float2 rope_row_cached(float2 x, int lane, float4 t, const RopeDev & r) {
    const int l = lane % (D / 2);
    if ((r.mode & 2) == 0) {
        if (2 * l >= r.n_dims) return x;
        const float c = (l & 1) ? t.z : t.x, sn = (l & 1) ? t.w : t.y;
        return make_float2(rot0(x.x, x.y, c, sn), rot1(x.x, x.y, c, sn));
    }
    constexpr int HL = D / 4;
    const float px = __shfl_xor(x.x, HL), py = __shfl_xor(x.y, HL);
    return l < HL ? make_float2(rot0(x.x, px, t.x, t.y), rot0(x.y, py, t.z, t.w))
                  : make_float2(rot1(px, x.x, t.x, t.y), rot1(py, x.y, t.z, t.w));
}

// T = float (ggml_compute_forward_rope_f32) or _Float16 (ggml_compute_forward_rope_f16, :10855: f16 in, f32 math, f16 out —
// what the reference's K-shift runs on an f16 cache, /ref/src/llama-context.cpp:483-492)
template <typename T>
__global__ void __launch_bounds__(64) rope_kernel(const T * x, const int32_t * pos, T * out, int n_head,
                                                  int D, const RopeDev r, float sin_sign, int64_t xs_tok, int64_t xs_head) {
    const int t = blockIdx.y, h = blockIdx.x;
    const T * src = x + (size_t) t * xs_tok + (size_t) h * xs_head;
    T *       dst = out + ((size_t) t * n_head + h) * D;
    const int p = pos[t];
    const bool neox = r.mode & 2;
    for (int ip = threadIdx.x; ip < D / 2; ip += 64) {
        if (2 * ip < r.n_dims) {
            const float2 cs = rope_cs(p, ip, r, sin_sign);
            const int i0 = neox ? ip : 2 * ip, i1 = neox ? ip + r.n_dims / 2 : 2 * ip + 1;
            const float x0 = (float) src[i0], x1 = (float) src[i1];
            dst[i0] = (T) rot0(x0, x1, cs.x, cs.y);
            dst[i1] = (T) rot1(x0, x1, cs.x, cs.y);
        } else {  // pass-through channels (:10841-10847)
            dst[2 * ip]     = src[2 * ip];
            dst[2 * ip + 1] = src[2 * ip + 1];
        }
    }
}

// ---- row <-> tile / reference-format codecs ----------------------------------------------------------------
// A row in flight: lane i holds the raw quants (q0, q1) of elements (2i, 2i+1) and the f16 bits of its block scale.
struct Q2 { int q0, q1; uint16_t d; };

__device__ __forceinline__ uint16_t h_bits(float f) { _Float16 h = (_Float16) f; return *reinterpret_cast<uint16_t *>(&h); }
__device__ __forceinline__ float    h_val(uint16_t b) { return (float) *reinterpret_cast<_Float16 *>(&b); }

// f32 pair -> Q2, following what ggml_cpy stores on the reference CPU backend:
//   f16 : GGML_FP32_TO_FP16 (RN-even)
//   q8_0: quantize_row_q8_0 AVX2 branch (/ref/ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862)
//   q4_0: quantize_row_q4_0_ref (/ref/ggml/src/ggml-quants.c:31-66), first-max-wins tie rule included
template <int T> __device__ __forceinline__ Q2 quantize(float2 x, int lane) {
#pragma clang fp contract(off)
    Q2 r;
    if constexpr (T == PKV_TYPE_F16) {
        r.q0 = h_bits(x.x); r.q1 = h_bits(x.y); r.d = 0;
    } else if constexpr (T == PKV_TYPE_Q8_0) {
        const float amax = row16_max(fmaxf(fabsf(x.x), fabsf(x.y)));
        const float d    = amax / 127.0f;
        const float id   = amax != 0.0f ? 127.0f / amax : 0.0f;
        r.d  = h_bits(d);
        r.q0 = (int) __builtin_rintf(x.x * id);
        r.q1 = (int) __builtin_rintf(x.y * id);
    } else {
        // (amax, max) with the reference's sequential "first strictly greater" rule == smallest index among ties
        float am = fabsf(x.x), mv = x.x;
        int   ix = 2 * lane;
        if (fabsf(x.y) > am) { am = fabsf(x.y); mv = x.y; ix = 2 * lane + 1; }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const float oa = __shfl_xor(am, m), ov = __shfl_xor(mv, m);
            const int   oi = __shfl_xor(ix, m);
            if (oa > am || (oa == am && oi < ix)) { am = oa; mv = ov; ix = oi; }
        }
        if (am == 0.0f) mv = 0.0f;              // an all-zero block: the reference's max stays its initial +0 also when the elements are -0 (d = +0 / -8 = -0)
        const float d  = mv / -8.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        r.d  = h_bits(d);
        const int a0 = (int) (signed char) (int) (x.x * id + 8.5f), a1 = (int) (signed char) (int) (x.y * id + 8.5f);
        r.q0 = min(15, a0) & 0xFF;
        r.q1 = min(15, a1) & 0xFF;
    }
    return r;
}
template <int T> __device__ __forceinline__ float2 dequantize(const Q2 & v) {
#pragma clang fp contract(off)
    if constexpr (T == PKV_TYPE_F16) return make_float2(h_val((uint16_t) v.q0), h_val((uint16_t) v.q1));
    else if constexpr (T == PKV_TYPE_Q8_0) { const float d = h_val(v.d); return make_float2(v.q0 * d, v.q1 * d); }
    else { const float d = h_val(v.d); return make_float2((v.q0 - 8) * d, (v.q1 - 8) * d); }
}

// K-layout / V-layout tile access ("Pool layout", include/mi355_paged_kv.h; V interleave in pkv_attn_decode.hip)
// V tiles: [tok/TL][D] dwords, each dword = the TL tokens of one d (f16 TL 2, q8_0 TL 4, q4_0 TL 8); scales
// half d[tok/TL][D/32][TL] after the quants.
template <int T, bool IS_V>
__device__ __forceinline__ void tile_write(char * tile, int row, int lane, const Q2 & v, int BS, int D) {
    const int e0 = 2 * lane;
    if constexpr (T == PKV_TYPE_Q4_0 && !IS_V) {  // byte j of the block = elem j | elem (j+16) << 4  (block_q4_0)
        const int p0 = __shfl_xor(v.q0, 8), p1 = __shfl_xor(v.q1, 8);
        if (e0 >= D) return;
        const int nb = D / PKV_QK, blk = e0 / PKV_QK;
        if ((lane & 15) < 8)
            *reinterpret_cast<uint16_t *>(tile + (size_t) row * (D / 2) + blk * 16 + 2 * (lane & 15)) =
                (uint16_t) ((v.q0 | (p0 << 4)) | ((v.q1 | (p1 << 4)) << 8));
        if ((lane & 15) == 0) *reinterpret_cast<uint16_t *>(tile + tile_quant_bytes(T, BS, D) + ((size_t) row * nb + blk) * 2) = v.d;
        return;
    }
    if (e0 >= D) return;
    const int nb = D / PKV_QK, blk = e0 / PKV_QK;
    char *    sc = tile + tile_quant_bytes(T, BS, D);
    if constexpr (T == PKV_TYPE_F16) {
        if (!IS_V) {
            *reinterpret_cast<uint32_t *>(tile + ((size_t) row * D + e0) * 2) = (uint32_t) (v.q0 & 0xFFFF) | ((uint32_t) v.q1 << 16);
        } else {
            char * p = tile + ((size_t) (row / 2) * D + e0) * 4 + (row & 1) * 2;
            *reinterpret_cast<uint16_t *>(p)     = (uint16_t) v.q0;
            *reinterpret_cast<uint16_t *>(p + 4) = (uint16_t) v.q1;
        }
    } else if constexpr (T == PKV_TYPE_Q8_0) {
        if (!IS_V) {
            *reinterpret_cast<uint16_t *>(tile + (size_t) row * D + e0) = (uint16_t) ((v.q0 & 0xFF) | ((v.q1 & 0xFF) << 8));
            if ((lane & 15) == 0) *reinterpret_cast<uint16_t *>(sc + ((size_t) row * nb + blk) * 2) = v.d;
        } else {
            char * p = tile + ((size_t) (row / 4) * D + e0) * 4 + (row & 3);
            p[0] = (char) v.q0;
            p[4] = (char) v.q1;
            if ((lane & 15) == 0) *reinterpret_cast<uint16_t *>(sc + (((size_t) (row / 4) * nb + blk) * 4 + (row & 3)) * 2) = v.d;
        }
    } else {  // q4_0 V: eight tokens share a dword -> update only this token's nibble, atomically
        uint32_t * w = reinterpret_cast<uint32_t *>(tile + ((size_t) (row / 8) * D + e0) * 4);
        const int  sh = 4 * (row & 7);
        atomicXor(w,     ((w[0] >> sh & 0xFu) ^ (uint32_t) v.q0) << sh);   // own nibble only changes through this lane
        atomicXor(w + 1, ((w[1] >> sh & 0xFu) ^ (uint32_t) v.q1) << sh);
        if ((lane & 15) == 0) *reinterpret_cast<uint16_t *>(sc + (((size_t) (row / 8) * nb + blk) * 8 + (row & 7)) * 2) = v.d;
    }
}
template <int T, bool IS_V>
__device__ __forceinline__ Q2 tile_read(const char * tile, int row, int lane, int BS, int D) {
    Q2 r{0, 0, 0};
    const int e0 = 2 * lane;
    if (e0 >= D) return r;
    const int    nb = D / PKV_QK, blk = e0 / PKV_QK;
    const char * sc = tile + tile_quant_bytes(T, BS, D);
    if constexpr (T == PKV_TYPE_F16) {
        if (!IS_V) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(tile + ((size_t) row * D + e0) * 2);
            r.q0 = w & 0xFFFF; r.q1 = w >> 16;
        } else {
            const char * p = tile + ((size_t) (row / 2) * D + e0) * 4 + (row & 1) * 2;
            r.q0 = *reinterpret_cast<const uint16_t *>(p);
            r.q1 = *reinterpret_cast<const uint16_t *>(p + 4);
        }
    } else if constexpr (T == PKV_TYPE_Q8_0) {
        if (!IS_V) {
            const uint16_t pk = *reinterpret_cast<const uint16_t *>(tile + (size_t) row * D + e0);
            r.q0 = (int) (signed char) (pk & 0xFF); r.q1 = (int) (signed char) (pk >> 8);
            r.d  = *reinterpret_cast<const uint16_t *>(sc + ((size_t) row * nb + blk) * 2);
        } else {
            const char * p = tile + ((size_t) (row / 4) * D + e0) * 4 + (row & 3);
            r.q0 = (int) (signed char) p[0]; r.q1 = (int) (signed char) p[4];
            r.d  = *reinterpret_cast<const uint16_t *>(sc + (((size_t) (row / 4) * nb + blk) * 4 + (row & 3)) * 2);
        }
    } else {
        if (!IS_V) {
            const int      j  = (lane & 15) & 7;  // lanes 0..7 read low nibbles, 8..15 the high nibbles of the same bytes
            const uint16_t pk = *reinterpret_cast<const uint16_t *>(tile + (size_t) row * (D / 2) + blk * 16 + 2 * j);
            const int      sh = (lane & 15) < 8 ? 0 : 4;
            r.q0 = (pk >> sh) & 0xF; r.q1 = (pk >> (8 + sh)) & 0xF;
            r.d  = *reinterpret_cast<const uint16_t *>(sc + ((size_t) row * nb + blk) * 2);
        } else {
            const uint32_t * w = reinterpret_cast<const uint32_t *>(tile + ((size_t) (row / 8) * D + e0) * 4);
            const int sh = 4 * (row & 7);
            r.q0 = (w[0] >> sh) & 0xF; r.q1 = (w[1] >> sh) & 0xF;
            r.d  = *reinterpret_cast<const uint16_t *>(sc + (((size_t) (row / 8) * nb + blk) * 8 + (row & 7)) * 2);
        }
    }
    return r;
}
}  // namespace
}  // namespace pkv
