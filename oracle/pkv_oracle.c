// oracle/pkv_oracle.c — CPU restatement of the reference's paged-KV hot path (see pkv_oracle.h).
//
// TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and bench.py's cpu_baseline leg.
// Parity status: PINNED against oracle/_ref (the reference compiled in place) and tests/golden/.
//
// This is our own scalar restatement; where the reference's result depends on its SIMD evaluation
// order (x86 AVX2 build: 4x8-lane f32 accumulators, hsum order) we reproduce that order so that the
// restatement agrees with oracle/_ref to the last bit wherever libm's expf/cosf/sinf allow.
#define _GNU_SOURCE
#include "pkv_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define QK 32  // QK8_0 == QK4_0 == 32 (ggml/src/ggml-common.h:155,197)

static inline uint16_t f2h(float f) { return _cvtss_sh(f, 0); }  // RN-even, as GGML_FP32_TO_FP16 w/ F16C
static inline float    h2f(uint16_t h) { return _cvtsh_ss(h); }

size_t pkv_o_row_size(int type, int64_t ne) {
    switch (type) {
        case PKV_O_F32:  return (size_t) ne * 4;
        case PKV_O_F16:  return (size_t) ne * 2;
        case PKV_O_BF16: return (size_t) ne * 2;
        case PKV_O_Q8_0: return (size_t) (ne / QK) * 34;  // block_q8_0 {half d; int8 qs[32]} ggml-common.h:203-208
        case PKV_O_Q4_0: return (size_t) (ne / QK) * 18;  // block_q4_0 {half d; uint8 qs[16]} ggml-common.h:161-166
        case PKV_O_Q4_1: return (size_t) (ne / QK) * 20;  // block_q4_1 {half d, m; uint8 qs[16]} ggml-common.h:168-180
        case PKV_O_Q5_0: return (size_t) (ne / QK) * 22;  // block_q5_0 {half d; uint8 qh[4]; uint8 qs[16]} :182-188
        case PKV_O_Q5_1: return (size_t) (ne / QK) * 24;  // block_q5_1 {half d, m; uint8 qh[4]; uint8 qs[16]} :190-203
        case PKV_O_IQ4_NL: return (size_t) (ne / QK) * 18; // block_iq4_nl {half d; uint8 qs[16]} :421-426
        default: return 0;
    }
}

void pkv_o_f32_to_f16(const float * x, uint16_t * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = f2h(x[i]); }
void pkv_o_f16_to_f32(const uint16_t * x, float * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = h2f(x[i]); }

// ggml_compute_fp32_to_bf16 / ggml_compute_bf16_to_fp32 (ggml/src/ggml-impl.h:519-551): round to nearest even, NaNs forced quiet
static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t) ((u >> 16) | 64);
    return (uint16_t) ((u + (0x7fffu + ((u >> 16) & 1))) >> 16);
}
static inline float bf2f(uint16_t h) { const uint32_t u = (uint32_t) h << 16; float f; memcpy(&f, &u, 4); return f; }
void pkv_o_f32_to_bf16(const float * x, uint16_t * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = f2bf(x[i]); }
void pkv_o_bf16_to_f32(const uint16_t * x, float * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = bf2f(x[i]); }

// ggml/src/ggml-quants.c:194-215
void pkv_o_quantize_q8_0_ref(const float * x, void * vy, int64_t k) {
    uint8_t * y = (uint8_t *) vy;
    for (int64_t i = 0; i < k / QK; ++i, y += 34) {
        float amax = 0.0f;
        for (int j = 0; j < QK; ++j) amax = fmaxf(amax, fabsf(x[i * QK + j]));
        const float d  = amax / 127;
        const float id = d ? 1.0f / d : 0.0f;
        const uint16_t dh = f2h(d);
        memcpy(y, &dh, 2);
        for (int j = 0; j < QK; ++j) ((int8_t *) (y + 2))[j] = (int8_t) roundf(x[i * QK + j] * id);
    }
}

// ggml/src/ggml-cpu/ggml-cpu-quants.c:808-862 (AVX2 branch): d = max/127, id = 127/max, _MM_ROUND_NEAREST
void pkv_o_quantize_q8_0_cpu(const float * x, void * vy, int64_t k) {
    uint8_t * y = (uint8_t *) vy;
    for (int64_t i = 0; i < k / QK; ++i, y += 34) {
        float amax = 0.0f;
        for (int j = 0; j < QK; ++j) amax = fmaxf(amax, fabsf(x[i * QK + j]));
        const float d  = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        const uint16_t dh = f2h(d);
        memcpy(y, &dh, 2);
        for (int j = 0; j < QK; ++j) ((int8_t *) (y + 2))[j] = (int8_t) nearbyintf(x[i * QK + j] * id);  // ties-to-even
    }
}

// ggml/src/ggml-quants.c:31-66
void pkv_o_quantize_q4_0(const float * x, void * vy, int64_t k) {
    uint8_t * y = (uint8_t *) vy;
    for (int64_t i = 0; i < k / QK; ++i, y += 18) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < QK; ++j) {
            const float v = x[i * QK + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d  = max / -8;
        const float id = d ? 1.0f / d : 0.0f;
        const uint16_t dh = f2h(d);
        memcpy(y, &dh, 2);
        for (int j = 0; j < QK / 2; ++j) {
            const float x0 = x[i * QK + j] * id;
            const float x1 = x[i * QK + QK / 2 + j] * id;
            const int a0 = (int8_t) (x0 + 8.5f), a1 = (int8_t) (x1 + 8.5f);
            const uint8_t q0 = (uint8_t) (a0 < 15 ? a0 : 15), q1 = (uint8_t) (a1 < 15 ? a1 : 15);
            y[2 + j] = (uint8_t) (q0 | (q1 << 4));
        }
    }
}

// ---- the remaining CLI-whitelisted cache types (common/arg.cpp:172-182): q4_1, q5_0, q5_1, iq4_nl.  The CPU backend's from_float of each forwards to
//      the *_ref quantizer (ggml-cpu-quants.c:727-734, :11760-11763).  Compiled -std=c11 like oracle/_ref: nothing below is contracted into an FMA.
static const int8_t kvalues_iq4nl[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};   // ggml-quants.c:2434
static inline void put_h(uint8_t * p, float f) { const uint16_t h = f2h(f); memcpy(p, &h, 2); }
static inline float get_h(const uint8_t * p) { uint16_t h; memcpy(&h, p, 2); return h2f(h); }
// quantize_row_q4_1_ref, ggml-quants.c:68-103
static void quantize_q4_1(const float * x, uint8_t * y, int64_t k) {
    for (int64_t i = 0; i < k / QK; ++i, y += 20) {
        float min = FLT_MAX, max = -FLT_MAX;
        for (int j = 0; j < QK; ++j) { const float v = x[i * QK + j]; if (v < min) min = v; if (v > max) max = v; }
        const float d = (max - min) / 15, id = d ? 1.0f / d : 0.0f;
        put_h(y, d); put_h(y + 2, min);
        for (int j = 0; j < 16; ++j) {
            const float x0 = (x[i * QK + j] - min) * id, x1 = (x[i * QK + 16 + j] - min) * id;
            const int a0 = (int8_t) (x0 + 0.5f), a1 = (int8_t) (x1 + 0.5f);
            y[4 + j] = (uint8_t) ((uint8_t) (a0 < 15 ? a0 : 15) | ((uint8_t) (a1 < 15 ? a1 : 15) << 4));
        }
    }
}
// quantize_row_q5_0_ref, :105-147
static void quantize_q5_0(const float * x, uint8_t * y, int64_t k) {
    for (int64_t i = 0; i < k / QK; ++i, y += 22) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < QK; ++j) { const float v = x[i * QK + j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
        const float d = max / -16, id = d ? 1.0f / d : 0.0f;
        put_h(y, d);
        uint32_t qh = 0;
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i * QK + j] * id, x1 = x[i * QK + 16 + j] * id;
            const int a0 = (int8_t) (x0 + 16.5f), a1 = (int8_t) (x1 + 16.5f);
            const uint8_t xi0 = (uint8_t) (a0 < 31 ? a0 : 31), xi1 = (uint8_t) (a1 < 31 ? a1 : 31);
            y[6 + j] = (uint8_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(y + 2, &qh, 4);
    }
}
// quantize_row_q5_1_ref, :149-190
static void quantize_q5_1(const float * x, uint8_t * y, int64_t k) {
    for (int64_t i = 0; i < k / QK; ++i, y += 24) {
        float min = FLT_MAX, max = -FLT_MAX;
        for (int j = 0; j < QK; ++j) { const float v = x[i * QK + j]; if (v < min) min = v; if (v > max) max = v; }
        const float d = (max - min) / 31, id = d ? 1.0f / d : 0.0f;
        put_h(y, d); put_h(y + 2, min);
        uint32_t qh = 0;
        for (int j = 0; j < 16; ++j) {
            const float x0 = (x[i * QK + j] - min) * id, x1 = (x[i * QK + 16 + j] - min) * id;
            const uint8_t xi0 = (uint8_t) (x0 + 0.5f), xi1 = (uint8_t) (x1 + 0.5f);
            y[8 + j] = (uint8_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(y + 4, &qh, 4);
    }
}
// best_index_int8, ggml-quants.c:4562-4571
static inline int best_index_int8(int n, const int8_t * val, float x) {
    if (x <= val[0]) return 0;
    if (x >= val[n - 1]) return n - 1;
    int ml = 0, mu = n - 1;
    while (mu - ml > 1) { const int mav = (ml + mu) / 2; if (x < val[mav]) mu = mav; else ml = mav; }
    return x - val[mu - 1] < val[mu] - x ? mu - 1 : mu;
}
// quantize_row_iq4_nl_ref = quantize_row_iq4_nl_impl(32, 32, ..., ntry = -1), :4573-4660, :4707-4720: one block of 32, weights x^2, the scale of the first pass
// re-fitted as sum(w q x) / sum(w q q), no search, the indices of the FIRST pass kept.  (A block whose amax < 1e-15 keeps d = 0 and — in the reference — whatever
// its index scratch held before: such blocks are not compared byte for byte.)
static void quantize_iq4_nl(const float * x, uint8_t * y, int64_t k) {
    uint8_t L[QK];
    memset(L, 0, sizeof(L));
    for (int64_t i = 0; i < k / QK; ++i, y += 18) {
        const float * xb = x + i * QK;
        memset(y + 2, 0, 16);
        put_h(y, 0.0f);
        float amax = 0, max = 0;
        for (int j = 0; j < QK; ++j) { const float ax = fabsf(xb[j]); if (ax > amax) { amax = ax; max = xb[j]; } }
        if (amax >= 1e-15f) {
            float d = max / kvalues_iq4nl[0];
            const float id = 1 / d;
            float sumqx = 0, sumq2 = 0;
            for (int j = 0; j < QK; ++j) {
                const float al = id * xb[j];
                const int l = best_index_int8(16, kvalues_iq4nl, al);
                L[j] = (uint8_t) l;
                const float q = kvalues_iq4nl[l], w = xb[j] * xb[j];
                sumqx += w * q * xb[j];
                sumq2 += w * q * q;
            }
            d = sumqx / sumq2;
            put_h(y, d);
        }
        for (int j = 0; j < 16; ++j) y[2 + j] = (uint8_t) (L[j] | (L[16 + j] << 4));
    }
}
// dequantize_row_q4_1 / _q5_0 / _q5_1 / _iq4_nl, ggml-quants.c:275-346, :2436-2452
static void dequantize_new(int type, const uint8_t * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k / QK; ++i, y += QK) {
        if (type == PKV_O_Q4_1) {
            const uint8_t * b = x + i * 20;
            const float d = get_h(b), m = get_h(b + 2);
            for (int j = 0; j < 16; ++j) { y[j] = (b[4 + j] & 0x0F) * d + m; y[j + 16] = (b[4 + j] >> 4) * d + m; }
        } else if (type == PKV_O_Q5_0) {
            const uint8_t * b = x + i * 22;
            const float d = get_h(b);
            uint32_t qh; memcpy(&qh, b + 2, 4);
            for (int j = 0; j < 16; ++j) {
                const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10, xh_1 = ((qh >> (j + 12))) & 0x10;
                y[j] = (((b[6 + j] & 0x0F) | xh_0) - 16) * d; y[j + 16] = (((b[6 + j] >> 4) | xh_1) - 16) * d;
            }
        } else if (type == PKV_O_Q5_1) {
            const uint8_t * b = x + i * 24;
            const float d = get_h(b), m = get_h(b + 2);
            uint32_t qh; memcpy(&qh, b + 4, 4);
            for (int j = 0; j < 16; ++j) {
                const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10, xh_1 = ((qh >> (j + 12))) & 0x10;
                y[j] = ((b[8 + j] & 0x0F) | xh_0) * d + m; y[j + 16] = ((b[8 + j] >> 4) | xh_1) * d + m;
            }
        } else {
            const uint8_t * b = x + i * 18;
            const float d = get_h(b);
            for (int j = 0; j < 16; ++j) { y[j] = d * kvalues_iq4nl[b[2 + j] & 0xf]; y[j + 16] = d * kvalues_iq4nl[b[2 + j] >> 4]; }
        }
    }
}
// quantize_row_q8_1, AVX2 branch (ggml-cpu-quants.c:1138-1180): the q8_0 integers plus s = f16(d * sum of the quants); block {half d; half s; int8 qs[32]} 36 B
static void quantize_q8_1_cpu(const float * x, uint8_t * y, int64_t k) {
    for (int64_t i = 0; i < k / QK; ++i, y += 36) {
        float amax = 0.0f;
        for (int j = 0; j < QK; ++j) amax = fmaxf(amax, fabsf(x[i * QK + j]));
        const float d = amax / 127.f, id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        int sum = 0;
        for (int j = 0; j < QK; ++j) { const int q = (int) nearbyintf(x[i * QK + j] * id); ((int8_t *) (y + 4))[j] = (int8_t) q; sum += q; }
        put_h(y, d); put_h(y + 2, d * sum);
    }
}

// ggml/src/ggml-quants.c:349-362 (q8_0), :255-273 (q4_0)
void pkv_o_to_float(int type, const void * vx, float * y, int64_t k) {
    const uint8_t * x = (const uint8_t *) vx;
    if (type == PKV_O_F32) { memcpy(y, vx, (size_t) k * 4); return; }
    if (type == PKV_O_F16) { pkv_o_f16_to_f32((const uint16_t *) vx, y, k); return; }
    if (type == PKV_O_BF16) { pkv_o_bf16_to_f32((const uint16_t *) vx, y, k); return; }
    if (type == PKV_O_Q4_1 || type == PKV_O_Q5_0 || type == PKV_O_Q5_1 || type == PKV_O_IQ4_NL) { dequantize_new(type, x, y, k); return; }
    for (int64_t i = 0; i < k / QK; ++i) {
        uint16_t dh;
        if (type == PKV_O_Q8_0) {
            memcpy(&dh, x + i * 34, 2);
            const float d = h2f(dh);
            const int8_t * qs = (const int8_t *) (x + i * 34 + 2);
            for (int j = 0; j < QK; ++j) y[i * QK + j] = qs[j] * d;
        } else {
            memcpy(&dh, x + i * 18, 2);
            const float d = h2f(dh);
            const uint8_t * qs = x + i * 18 + 2;
            for (int j = 0; j < QK / 2; ++j) {
                y[i * QK + j]          = ((qs[j] & 0x0F) - 8) * d;
                y[i * QK + j + QK / 2] = ((qs[j] >> 4) - 8) * d;
            }
        }
    }
}

void pkv_o_from_float(int type, const float * x, void * y, int64_t k) {
    switch (type) {
        case PKV_O_F32:  memcpy(y, x, (size_t) k * 4); break;
        case PKV_O_F16:  pkv_o_f32_to_f16(x, (uint16_t *) y, k); break;
        case PKV_O_BF16: pkv_o_f32_to_bf16(x, (uint16_t *) y, k); break;
        case PKV_O_Q8_0: pkv_o_quantize_q8_0_cpu(x, y, k); break;
        case PKV_O_Q4_0: pkv_o_quantize_q4_0(x, y, k); break;
        case PKV_O_Q4_1: quantize_q4_1(x, (uint8_t *) y, k); break;
        case PKV_O_Q5_0: quantize_q5_0(x, (uint8_t *) y, k); break;
        case PKV_O_Q5_1: quantize_q5_1(x, (uint8_t *) y, k); break;
        case PKV_O_IQ4_NL: quantize_iq4_nl(x, (uint8_t *) y, k); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// RoPE — ggml-cpu.c:10571-10609 (yarn + cache init), :10669-10852 (apply); corr dims ggml.c:3735-3747
static float yarn_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}

void pkv_o_rope(const float * x, const int32_t * pos, const float * ff, float * out,
                int ne0, int n_head, int n_tok, int n_dims, int mode, int n_ctx_orig,
                float freq_base, float freq_scale, float ext_factor, float attn_factor,
                float beta_fast, float beta_slow, int inverse) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    float corr[2];
    {
        const float start = floorf(yarn_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base));
        const float end   = ceilf(yarn_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base));
        corr[0] = start > 0 ? start : 0;
        corr[1] = end < n_dims - 1 ? end : n_dims - 1;
    }
    const int   is_neox  = mode & 2;
    const float sin_sign = inverse ? -1.0f : 1.0f;
    float * cache = (float *) malloc(sizeof(float) * (size_t) ne0);

    for (int t = 0; t < n_tok; ++t) {
        float theta = (float) pos[t];
        for (int i0 = 0; i0 < ne0; i0 += 2) {
            const float f            = ff ? ff[i0 / 2] : 1.0f;
            const float theta_extrap = theta / f;
            const float theta_interp = freq_scale * theta_extrap;
            float th = theta_interp, mscale = attn_factor;
            if (ext_factor != 0.0f) {
                const float hi_lo = corr[1] - corr[0];
                const float yv    = (i0 / 2 - corr[0]) / (hi_lo > 0.001f ? hi_lo : 0.001f);
                const float ramp  = (1 - fminf(1, fmaxf(0, yv))) * ext_factor;
                th = theta_interp * (1 - ramp) + theta_extrap * ramp;
                mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
            }
            cache[i0]     = cosf(th) * mscale;
            cache[i0 + 1] = sinf(th) * mscale * sin_sign;
            theta *= theta_scale;
        }
        for (int h = 0; h < n_head; ++h) {
            const float * s = x + ((size_t) t * n_head + h) * ne0;
            float *       d = out + ((size_t) t * n_head + h) * ne0;
            if (is_neox) {
                for (int i0 = 0; i0 < n_dims; i0 += 2) {
                    const int ic = i0 / 2;
                    const float c = cache[i0], sn = cache[i0 + 1], x0 = s[ic], x1 = s[ic + n_dims / 2];
                    d[ic]              = x0 * c - x1 * sn;
                    d[ic + n_dims / 2] = x0 * sn + x1 * c;
                }
            } else {
                for (int i0 = 0; i0 < n_dims; i0 += 2) {
                    const float c = cache[i0], sn = cache[i0 + 1], x0 = s[i0], x1 = s[i0 + 1];
                    d[i0]     = x0 * c - x1 * sn;
                    d[i0 + 1] = x0 * sn + x1 * c;
                }
            }
            for (int i0 = n_dims; i0 < ne0; ++i0) d[i0] = s[i0];
        }
    }
    free(cache);
}

// ---------------------------------------------------------------------------------------------------
// vec dots in the reference's AVX2 evaluation order

// ggml_vec_dot_f16 (ggml-cpu.c:1565-1603), GGML_F16_STEP 32 / EPR 8 / ARR 4, FMA; reduce :671-689
static float vec_dot_f16(int n, const uint16_t * x, const uint16_t * y) {
    float sum[4][8];
    memset(sum, 0, sizeof(sum));
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l)
                sum[j][l] = fmaf(h2f(x[i + j * 8 + l]), h2f(y[i + j * 8 + l]), sum[j][l]);
    float t[4];
    for (int l = 0; l < 8; ++l) { sum[0][l] += sum[2][l]; sum[1][l] += sum[3][l]; }
    for (int l = 0; l < 8; ++l) sum[0][l] += sum[1][l];
    for (int l = 0; l < 4; ++l) t[l] = sum[0][l] + sum[0][l + 4];
    double sumf = (double) ((t[0] + t[1]) + (t[2] + t[3]));
    for (int i = np; i < n; ++i) sumf += (double) (h2f(x[i]) * h2f(y[i]));
    return (float) sumf;
}

// ggml_vec_dot_bf16, AVX2 branch (ggml-cpu.c:1537-1562): four 8-lane accumulators over steps of 32 (product then add; BF16_FMA: what the
// compiler made of it in oracle/_ref), c1 + c3 and c2 + c4, the halves, movehl, movehdup; leftovers in double
#ifndef BF16_FMA
#define BF16_FMA 1
#endif
static float vec_dot_bf16(int n, const uint16_t * x, const uint16_t * y) {
    float c[4][8];
    memset(c, 0, sizeof(c));
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) {
                const float a = bf2f(x[i + j * 8 + l]), b = bf2f(y[i + j * 8 + l]);
#if BF16_FMA
                c[j][l] = fmaf(a, b, c[j][l]);
#else
                volatile float pr = a * b;
                c[j][l] = pr + c[j][l];
#endif
            }
    float t[8], g[4];
    for (int l = 0; l < 8; ++l) t[l] = (c[0][l] + c[2][l]) + (c[1][l] + c[3][l]);
    for (int l = 0; l < 4; ++l) g[l] = t[l + 4] + t[l];
    g[0] += g[2]; g[1] += g[3];
    double sumf = (double) (g[0] + g[1]);
    for (int i = np; i < n; ++i) sumf += (double) (bf2f(x[i]) * bf2f(y[i]));
    return (float) sumf;
}

static inline float hsum8(const float * a) {  // hsum_float_8, ggml-cpu-quants.c:49-55
    float r[4];
    for (int l = 0; l < 4; ++l) r[l] = a[l + 4] + a[l];
    r[0] += r[2];
    r[1] += r[3];
    return r[0] + r[1];
}

// ggml_vec_dot_q8_0_q8_0 AVX2 (ggml-cpu-quants.c:3935-3952): per block 8 int32 lanes of 4 products each
static float vec_dot_q8_0_q8_0(int n, const uint8_t * x, const uint8_t * y) {
    float acc[8] = {0};
    for (int ib = 0; ib < n / QK; ++ib) {
        uint16_t dx, dy;
        memcpy(&dx, x + ib * 34, 2);
        memcpy(&dy, y + ib * 34, 2);
        const float d = h2f(dx) * h2f(dy);
        const int8_t * qx = (const int8_t *) (x + ib * 34 + 2);
        const int8_t * qy = (const int8_t *) (y + ib * 34 + 2);
        for (int l = 0; l < 8; ++l) {
            int s = 0;
            for (int j = 0; j < 4; ++j) s += (int) qx[l * 4 + j] * (int) qy[l * 4 + j];
            acc[l] = fmaf(d, (float) s, acc[l]);
        }
    }
    return hsum8(acc);
}

// ggml_vec_dot_q4_0_q8_0 AVX2 (ggml-cpu-quants.c:2273-2296): low nibbles -> elems 0..15, high -> 16..31, minus 8
static float vec_dot_q4_0_q8_0(int n, const uint8_t * x, const uint8_t * y) {
    float acc[8] = {0};
    for (int ib = 0; ib < n / QK; ++ib) {
        uint16_t dx, dy;
        memcpy(&dx, x + ib * 18, 2);
        memcpy(&dy, y + ib * 34, 2);
        const float d = h2f(dx) * h2f(dy);
        const uint8_t * qx = x + ib * 18 + 2;
        const int8_t *  qy = (const int8_t *) (y + ib * 34 + 2);
        int8_t e[QK];
        for (int j = 0; j < 16; ++j) { e[j] = (int8_t) ((qx[j] & 0x0F) - 8); e[j + 16] = (int8_t) ((qx[j] >> 4) - 8); }
        for (int l = 0; l < 8; ++l) {
            int s = 0;
            for (int j = 0; j < 4; ++j) s += (int) e[l * 4 + j] * (int) qy[l * 4 + j];
            acc[l] = fmaf(d, (float) s, acc[l]);
        }
    }
    return hsum8(acc);
}

// the element e (0..31) of a block as the AVX2 dots see it: low nibbles -> 0..15, high -> 16..31; q5: the fifth bit from qh
static inline int blk_q(int type, const uint8_t * b, int e) {
    if (type == PKV_O_Q4_1) return e < 16 ? b[4 + e] & 0xF : b[4 + e - 16] >> 4;
    if (type == PKV_O_IQ4_NL) return kvalues_iq4nl[e < 16 ? b[2 + e] & 0xF : b[2 + e - 16] >> 4];
    const int o = type == PKV_O_Q5_0 ? 2 : 4;
    uint32_t qh; memcpy(&qh, b + o, 4);
    const int nib = e < 16 ? b[o + 4 + e] & 0xF : b[o + 4 + e - 16] >> 4;
    const int q = nib | (int) (((qh >> e) & 1u) << 4);
    return type == PKV_O_Q5_0 ? q - 16 : q;
}
// ggml_vec_dot_q4_1_q8_1 (ggml-cpu-quants.c:2751-2786), _q5_0_q8_0 (:3080-3103), _q5_1_q8_1 (:3445-3471), _iq4_nl_q8_0 (:11999-12027), AVX2 branches: per
// block 8 lanes of 4 integer products, acc[l] = fma(d_x * d_y, lane sum, acc[l]); the *_1 types add m_x * s_y block by block into a scalar; iq4_nl keeps
// two accumulators (even / odd blocks) and adds them before the horizontal sum
static float vec_dot_new(int type, int n, const uint8_t * x, const uint8_t * y) {
    const int xs = type == PKV_O_Q4_1 ? 20 : type == PKV_O_Q5_0 ? 22 : type == PKV_O_Q5_1 ? 24 : 18;
    const int q81 = type == PKV_O_Q4_1 || type == PKV_O_Q5_1, ys = q81 ? 36 : 34, yq = q81 ? 4 : 2;
    float acc[2][8];
    memset(acc, 0, sizeof(acc));
    float summs = 0.0f;
    const int nbv = type == PKV_O_IQ4_NL ? (n / QK) & ~1 : n / QK;      // iq4_nl: blocks in pairs, an odd last one in the scalar tail (:12153-12161)
    for (int ib = 0; ib < nbv; ++ib) {
        const uint8_t * bx = x + ib * xs, * by = y + ib * ys;
        const float d = get_h(bx) * get_h(by);
        if (q81) summs += get_h(bx + 2) * get_h(by + 2);
        float * a = acc[type == PKV_O_IQ4_NL ? (ib & 1) : 0];
        for (int l = 0; l < 8; ++l) {
            int sm = 0;
            for (int j = 0; j < 4; ++j) sm += blk_q(type, bx, l * 4 + j) * (int) ((const int8_t *) (by + yq))[l * 4 + j];
            a[l] = fmaf(d, (float) sm, a[l]);
        }
    }
    for (int l = 0; l < 8; ++l) acc[0][l] += acc[1][l];
    float sumf = q81 ? hsum8(acc[0]) + summs : hsum8(acc[0]);
    for (int ib = nbv; ib < n / QK; ++ib) {
        const uint8_t * bx = x + ib * xs, * by = y + ib * ys;
        const float d = get_h(by) * get_h(bx);
        int sumi = 0;
        for (int e = 0; e < QK; ++e) sumi += blk_q(type, bx, e) * (int) ((const int8_t *) (by + yq))[e];
        sumf += d * sumi;
    }
    return sumf;
}

// ---------------------------------------------------------------------------------------------------
static int kv_blocked(int t) { return t == PKV_O_Q8_0 || t == PKV_O_Q4_0 || t == PKV_O_Q4_1 || t == PKV_O_Q5_0 || t == PKV_O_Q5_1 || t == PKV_O_IQ4_NL; }
static int kv_ok(int t) { return t == PKV_O_F16 || t == PKV_O_BF16 || kv_blocked(t); }
// ggml_compute_forward_flash_attn_ext_f16, ggml-cpu.c:12219-12432
int pkv_o_flash_attn_ext(const float * q, const void * k, const void * v, const uint16_t * mask,
                         float * out, float * logits,
                         int D, int n_q, int n_head, int n_kv, int n_head_kv, int mask_stride,
                         int type_k, int type_v, float scale, float max_bias, float logit_softcap,
                         int n_threads) {
    if (!kv_ok(type_k) || !kv_ok(type_v)) return -1;
    if (D > 512 || ((kv_blocked(type_k) || kv_blocked(type_v)) && D % QK != 0)) return -1;
    if (logit_softcap != 0) scale /= logit_softcap;                                   // :12290-12292
    const uint32_t n_head_log2 = 1u << (uint32_t) floor(log2(n_head));               // :12295
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2);
    const float m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    const size_t rk = pkv_o_row_size(type_k, D), rv = pkv_o_row_size(type_v, D);
    const int gq = n_head / n_head_kv;                                               // rk2 :12264
    // k's vec_dot_type: f16 for f16, bf16 for bf16, q8_0 for q8_0/q4_0 (type_traits_cpu, ggml-cpu.c:240-300,403-405)
    // (q4_1 / q5_1: q8_1; q5_0 / iq4_nl: q8_0)
    const int q_type = type_k == PKV_O_F16 ? PKV_O_F16 : type_k == PKV_O_BF16 ? PKV_O_BF16 : (type_k == PKV_O_Q4_1 || type_k == PKV_O_Q5_1) ? PKV_O_Q8_1 : PKV_O_Q8_0;
    const int nr = n_head * n_q;
    const int D32 = D & ~31;                                                         // GGML_F32_STEP == GGML_F16_STEP == 32 (AVX2)
    if (n_threads < 1) n_threads = 1;

#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int ir = 0; ir < nr; ++ir) {
        const int h = ir / n_q, iq = ir % n_q;                                       // :12311-12313
        const float slope = (max_bias > 0.0f)
            ? ((uint32_t) h < n_head_log2 ? powf(m0, h + 1) : powf(m1, 2 * (h - n_head_log2) + 1)) : 1.0f;
        float    S = 0.0f, M = -INFINITY;
        float    VKQ32[512], V32[512];
        uint16_t VKQ16[512];
        uint8_t  Qq[512 * 2];
        memset(VKQ32, 0, sizeof(float) * D);
        memset(VKQ16, 0, sizeof(uint16_t) * D);
        const uint16_t * mp = mask ? mask + (size_t) iq * mask_stride : NULL;
        const int hk = h / gq;
        const float * pq = q + ((size_t) h * n_q + iq) * D;
        if (q_type == PKV_O_F16) pkv_o_f32_to_f16(pq, (uint16_t *) Qq, D);
        else if (q_type == PKV_O_BF16) pkv_o_f32_to_bf16(pq, (uint16_t *) Qq, D);
        else if (q_type == PKV_O_Q8_1) quantize_q8_1_cpu(pq, Qq, D);
        else                     pkv_o_quantize_q8_0_cpu(pq, Qq, D);                 // :12343
        float * lg = logits ? logits + ((size_t) h * n_q + iq) * n_kv : NULL;

        for (int ic = 0; ic < n_kv; ++ic) {
            const float mv = mp ? slope * h2f(mp[ic]) : 0.0f;
            if (mv == -INFINITY) { if (lg) lg[ic] = -INFINITY; continue; }           // :12349-12352
            const uint8_t * kd = (const uint8_t *) k + ((size_t) hk * n_kv + ic) * rk;
            float s;
            if (type_k == PKV_O_F16)       s = vec_dot_f16(D, (const uint16_t *) kd, (const uint16_t *) Qq);
            else if (type_k == PKV_O_BF16) s = vec_dot_bf16(D, (const uint16_t *) kd, (const uint16_t *) Qq);
            else if (type_k == PKV_O_Q8_0) s = vec_dot_q8_0_q8_0(D, kd, Qq);
            else if (type_k == PKV_O_Q4_0) s = vec_dot_q4_0_q8_0(D, kd, Qq);
            else                           s = vec_dot_new(type_k, D, kd, Qq);
            s = s * scale;
            if (logit_softcap != 0.0f) s = logit_softcap * tanhf(s);
            s += mv;
            if (lg) lg[ic] = s;
            const float Mold = M;
            float ms = 1.0f, vs = 1.0f;
            const uint8_t * vd = (const uint8_t *) v + ((size_t) hk * n_kv + ic) * rv;
            if (type_v == PKV_O_F16) {                                               // :12374-12388 (f16 accumulator)
                if (s > M) {
                    M = s; ms = expf(Mold - M);
                    for (int d = 0; d < D; ++d) VKQ16[d] = f2h(h2f(VKQ16[d]) * ms);  // ggml_vec_scale_f16
                } else {
                    vs = expf(s - M);
                }
                const uint16_t * vh = (const uint16_t *) vd;
                for (int d = 0; d < D32; ++d) VKQ16[d] = f2h(fmaf(h2f(vh[d]), vs, h2f(VKQ16[d])));  // ggml_vec_mad_f16 (FMA)
                for (int d = D32; d < D; ++d) VKQ16[d] = f2h(h2f(VKQ16[d]) + h2f(vh[d]) * vs);     // leftovers of a 32-element step: product, then sum (ggml-cpu.c:1693-1723)
            } else {                                                                 // :12389-12406
                if (s > M) {
                    M = s; ms = expf(Mold - M);
                    for (int d = 0; d < D; ++d) VKQ32[d] *= ms;
                } else {
                    vs = expf(s - M);
                }
                pkv_o_to_float(type_v, vd, V32, D);
                for (int d = 0; d < D32; ++d) VKQ32[d] = fmaf(V32[d], vs, VKQ32[d]);   // ggml_vec_mad_f32 (FMA)
                for (int d = D32; d < D; ++d) VKQ32[d] = VKQ32[d] + V32[d] * vs;      // leftovers (head size 80): not fused (:1662-1691; -std=c11 contracts nothing)
            }
            S = S * ms + vs;                                                         // :12408
        }
        if (type_v == PKV_O_F16) for (int d = 0; d < D; ++d) VKQ32[d] = h2f(VKQ16[d]);
        const float S_inv = 1.0f / S;                                                // :12418
        float * o = out + ((size_t) iq * n_head + h) * D;                            // permuted dst :12430
        for (int d = 0; d < D; ++d) o[d] = VKQ32[d] * S_inv;
    }
    return 0;
}

int pkv_o_flash_attn_f64(const float * q, const void * k, const void * v, const uint16_t * mask,
                         double * out, int D, int n_q, int n_head, int n_kv, int n_head_kv, int mask_stride,
                         int type_k, int type_v, float scale, float logit_softcap) {
    const size_t rk = pkv_o_row_size(type_k, D), rv = pkv_o_row_size(type_v, D);
    if (!rk || !rv) return -1;
    const int gq = n_head / n_head_kv;
    double sc = scale;
    if (logit_softcap != 0) sc /= logit_softcap;
#pragma omp parallel for schedule(static)
    for (int ir = 0; ir < n_head * n_q; ++ir) {
        const int h = ir / n_q, iq = ir % n_q, hk = h / gq;
        const float * pq = q + ((size_t) h * n_q + iq) * D;
        double * s = (double *) malloc(sizeof(double) * (size_t) n_kv);
        float row[512];
        double M = -INFINITY;
        for (int ic = 0; ic < n_kv; ++ic) {
            const float mv = mask ? h2f(mask[(size_t) iq * mask_stride + ic]) : 0.0f;
            if (mv == -INFINITY) { s[ic] = -INFINITY; continue; }
            pkv_o_to_float(type_k, (const uint8_t *) k + ((size_t) hk * n_kv + ic) * rk, row, D);
            double acc = 0;
            for (int d = 0; d < D; ++d) acc += (double) row[d] * (double) pq[d];
            acc *= sc;
            if (logit_softcap != 0) acc = logit_softcap * tanh(acc);
            s[ic] = acc + mv;
            if (s[ic] > M) M = s[ic];
        }
        double S = 0, o[512];
        memset(o, 0, sizeof(double) * D);
        for (int ic = 0; ic < n_kv; ++ic) {
            if (s[ic] == -INFINITY) continue;
            const double p = exp(s[ic] - M);
            S += p;
            pkv_o_to_float(type_v, (const uint8_t *) v + ((size_t) hk * n_kv + ic) * rv, row, D);
            for (int d = 0; d < D; ++d) o[d] += p * (double) row[d];
        }
        double * od = out + ((size_t) iq * n_head + h) * D;
        for (int d = 0; d < D; ++d) od[d] = o[d] / S;
        free(s);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// paged pool layout (include/mi355_paged_kv.h "Pool layout")
size_t pkv_o_tile_bytes(int type, int block_size, int D) {
    switch (type) {
        case PKV_O_F16:  return (size_t) block_size * D * 2;
        case PKV_O_Q8_0: return (size_t) block_size * D + (size_t) block_size * (D / QK) * 2;
        case PKV_O_Q4_0: return (size_t) block_size * D / 2 + (size_t) block_size * (D / QK) * 2;
        default: return 0;
    }
}

// one reference-format row (one kv head of one token) <-> its place inside a tile.
// K tiles: quants [tok][...] then scales [tok][D/32].
// V tiles: [tok/TL][D] dwords, each dword = the TL tokens of one d (f16 TL 2, q8_0 TL 4, q4_0 TL 8 nibbles), then
// scales [tok/TL][D/32][TL].  See include/mi355_paged_kv.h "Pool layout".
static inline int q4_get(const uint8_t * rb, int e) { return e < 16 ? rb[2 + e] & 0xF : rb[2 + e - 16] >> 4; }
static inline void q4_set(uint8_t * rb, int e, int n) {
    if (e < 16) rb[2 + e] = (uint8_t) ((rb[2 + e] & 0xF0) | n);
    else        rb[2 + e - 16] = (uint8_t) ((rb[2 + e - 16] & 0x0F) | (n << 4));
}
static void tile_row(uint8_t * tile, int type, int is_v, int off, uint8_t * row, int D, int block_size, int put) {
#define MOVE(tp, rp, n) do { if (put) memcpy((tp), (rp), (n)); else memcpy((rp), (tp), (n)); } while (0)
    const int nb = D / QK;
    if (type == PKV_O_F16) {
        if (!is_v) { MOVE(tile + (size_t) off * D * 2, row, (size_t) D * 2); return; }
        for (int d = 0; d < D; ++d) MOVE(tile + (((size_t) (off / 2) * D + d) * 2 + (off & 1)) * 2, row + (size_t) d * 2, 2);
        return;
    }
    uint8_t * sc = tile + (type == PKV_O_Q8_0 ? (size_t) block_size * D : (size_t) block_size * D / 2);
    for (int b = 0; b < nb; ++b) {
        if (type == PKV_O_Q8_0) {
            uint8_t * rb = row + (size_t) b * 34;
            if (!is_v) {
                MOVE(sc + ((size_t) off * nb + b) * 2, rb, 2);
                MOVE(tile + (size_t) off * D + b * QK, rb + 2, QK);
            } else {
                MOVE(sc + (((size_t) (off / 4) * nb + b) * 4 + (off & 3)) * 2, rb, 2);
                for (int e = 0; e < QK; ++e) MOVE(tile + ((size_t) (off / 4) * D + b * QK + e) * 4 + (off & 3), rb + 2 + e, 1);
            }
        } else {
            uint8_t * rb = row + (size_t) b * 18;
            if (!is_v) {
                MOVE(sc + ((size_t) off * nb + b) * 2, rb, 2);
                MOVE(tile + (size_t) off * (D / 2) + b * 16, rb + 2, 16);
            } else {
                MOVE(sc + (((size_t) (off / 8) * nb + b) * 8 + (off & 7)) * 2, rb, 2);
                for (int e = 0; e < QK; ++e) {
                    uint8_t * tp = tile + ((size_t) (off / 8) * D + b * QK + e) * 4 + (off & 7) / 2;  // nibble (off & 7) of the dword
                    const int sh = 4 * (off & 1);
                    if (put) *tp = (uint8_t) ((*tp & ~(0xF << sh)) | (q4_get(rb, e) << sh));
                    else     q4_set(rb, e, (*tp >> sh) & 0xF);
                }
            }
        }
    }
#undef MOVE
}

void pkv_o_pool_store(void * pool, int type, int is_v, const void * rows, const int32_t * slot_block,
                      const int32_t * slot_off, int n_tok, int n_head_kv, int D, int block_size) {
    const size_t tb = pkv_o_tile_bytes(type, block_size, D), rb = pkv_o_row_size(type, D);
    for (int t = 0; t < n_tok; ++t) {
        if (slot_block[t] < 0) continue;
        for (int h = 0; h < n_head_kv; ++h)
            tile_row((uint8_t *) pool + ((size_t) slot_block[t] * n_head_kv + h) * tb, type, is_v, slot_off[t],
                     (uint8_t *) rows + ((size_t) t * n_head_kv + h) * rb, D, block_size, 1);
    }
}

void pkv_o_pool_gather(const void * pool, int type, int is_v, const int32_t * block_table, int n_kv,
                       int n_head_kv, int D, int block_size, void * dst) {
    const size_t tb = pkv_o_tile_bytes(type, block_size, D), rb = pkv_o_row_size(type, D);
    for (int h = 0; h < n_head_kv; ++h)
        for (int i = 0; i < n_kv; ++i)
            tile_row((uint8_t *) pool + ((size_t) block_table[i / block_size] * n_head_kv + h) * tb, type, is_v,
                     i % block_size, (uint8_t *) dst + ((size_t) h * n_kv + i) * rb, D, block_size, 0);
}

int pkv_o_attn_paged(const float * q, const void * k_pool, const void * v_pool,
                     const int32_t * block_table, int max_blocks, const int32_t * seq_lens,
                     float * out, int n_seq, int n_q, int D, int n_head, int n_head_kv, int block_size,
                     int type_k, int type_v, float scale, float logit_softcap, int n_threads) {
    int rc = 0;
    for (int s = 0; s < n_seq && rc == 0; ++s) {
        const int n_kv = seq_lens[s];
        float * o = out + (size_t) s * n_q * n_head * D;
        if (n_kv <= 0) { memset(o, 0, sizeof(float) * (size_t) n_q * n_head * D); continue; }
        void * kc = malloc(pkv_o_row_size(type_k, D) * (size_t) n_kv * n_head_kv);
        void * vc = malloc(pkv_o_row_size(type_v, D) * (size_t) n_kv * n_head_kv);
        uint16_t * mask = (uint16_t *) malloc(sizeof(uint16_t) * (size_t) n_q * n_kv);
        pkv_o_pool_gather(k_pool, type_k, 0, block_table + (size_t) s * max_blocks, n_kv, n_head_kv, D, block_size, kc);
        pkv_o_pool_gather(v_pool, type_v, 1, block_table + (size_t) s * max_blocks, n_kv, n_head_kv, D, block_size, vc);
        // causal mask as llm_graph_input_attn_kv_unified::set_input builds it (src/llama-graph.cpp:429-468):
        // 0 where cell.pos <= query pos, -inf otherwise
        for (int j = 0; j < n_q; ++j) {
            const int qpos = n_kv - n_q + j;
            for (int i = 0; i < n_kv; ++i) mask[(size_t) j * n_kv + i] = i <= qpos ? 0x0000 : 0xFC00;
        }
        rc = pkv_o_flash_attn_ext(q + (size_t) s * n_q * n_head * D, kc, vc, mask, o, NULL, D, n_q, n_head, n_kv,
                                  n_head_kv, n_kv, type_k, type_v, scale, 0.0f, logit_softcap, n_threads);
        free(kc); free(vc); free(mask);
    }
    return rc;
}
