// pkv_wlayout.h — the PLANE layout of K-quant weight rows ("packed" rows, type | PKV_TYPE_PACKED).
//
// A GGUF holds a q4_K / q5_K / q6_K row as K / 256 blocks of 144 / 176 / 210 bytes (/ref/ggml/src/ggml-common.h:279-290, :291-308, :309-319).  Streamed that way by the
// decode-step products every lane's 16-byte load sits at a 144- / 210-byte stride: one wave instruction touches 64 different 128-byte lines, and each
// line is touched by two or three instructions — the vector L1 looks up one line per cycle, which caps a CU near 12 B / cycle (profiles/r03_gemv_read_
// pattern.txt).  The precedent for re-laying weights when they are uploaded is the reference's own CPU backend (repack on set_tensor,
// /ref/ggml/src/ggml-cpu/ggml-cpu-aarch64.cpp:3663-3810, :4145-4172).  Here the SAME bytes are permuted INSIDE each row (row r still occupies bytes
// [r * row_size, (r + 1) * row_size) of the tensor, so row-granular views, get_rows and partial tensor_set / tensor_get keep working):
//
//   gpr = K / 64 = number of 64-element groups ("items") of a row; item g = 4 b + p belongs to block b
//   q4_K row:  [qa plane: 16 B per item g][qb plane: 16 B per item][header plane: 16 B per block]
//                 qa(g) = qs[32 p .. 32 p + 16) of block b, qb(g) = qs[32 p + 16 .. 32 p + 32)   (low nibbles = elements 64 p + [0, 32), high = + 32)
//                 header(b) = {d, dmin, scales[12]}
//   q6_K row:  [la plane: 16 B per item][lb plane][hh plane][scale plane: 16 B per block][d plane: 2 B per block]
//                 item g = 4 b + 2 n + is:  la = ql[64 n + 16 is ..+16), lb = ql[64 n + 32 + 16 is ..+16), hh = qh[32 n + 16 is ..+16)
//   q5_K row:  [qa plane: 16 B per item][qb plane: 16 B per item][ha plane: 16 B per block][hb plane: 16 B per block][header plane: 16 B per block]
//                 qa / qb / header as q4_K (block_q5_K's qs / {d, dmin, scales[12]}), ha = qh[0 .. 16), hb = qh[16 .. 32) of block b — the fifth bits of a
//                 64-element group are spread over all 32 bytes of its block's qh (bit 2 p / 2 p + 1 of byte l), so the four items of a block share ha / hb
//                 as they share the header: three 256-byte wave reads beside the two of 1 KiB
//   every plane starts 16-byte aligned when the row does (q4_K: always; q6_K: K % 2048 == 0, else 2-byte aligned rows as in the GGUF order).
// A wave instruction of the products then reads 1 KiB of contiguous bytes per plane and every line is consumed by one instruction (non-temporal loads).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef __host__
#define __host__
#define __device__
#endif

namespace pkv {

constexpr int TYPE_PACKED_FLAG = 0x100;   // == PKV_TYPE_PACKED

// canonical (GGUF) byte offset inside a row -> byte offset inside the packed row.  `o` addresses any byte of the row.
__host__ __device__ inline size_t pk_offset_q4k(int gpr, size_t o) {
    const size_t b = o / 144, r = o % 144;
    if (r < 16) return (size_t) 32 * gpr + 16 * b + r;
    const size_t q = r - 16, p = q >> 5, h = (q >> 4) & 1;
    return h * 16 * (size_t) gpr + 16 * (4 * b + p) + (q & 15);
}
__host__ __device__ inline size_t pk_offset_q5k(int gpr, size_t o) {      // block_q5_K: {d, dmin, scales[12]} 16 B, qh[32], qs[128] = 176 B
    const size_t b = o / 176, r = o % 176;
    if (r < 16) return (size_t) 40 * gpr + 16 * b + r;
    if (r < 48) { const size_t q = r - 16; return (size_t) 32 * gpr + (q >> 4) * 4 * (size_t) gpr + 16 * b + (q & 15); }
    const size_t q = r - 48, p = q >> 5, h = (q >> 4) & 1;
    return h * 16 * (size_t) gpr + 16 * (4 * b + p) + (q & 15);
}
__host__ __device__ inline size_t pk_offset_q6k(int gpr, size_t o) {
    const size_t b = o / 210, r = o % 210;
    if (r < 128) { const size_t n = r >> 6, hi = (r >> 5) & 1, is = (r >> 4) & 1; return hi * 16 * (size_t) gpr + 16 * (4 * b + 2 * n + is) + (r & 15); }
    if (r < 192) { const size_t q = r - 128, n = q >> 5, is = (q >> 4) & 1; return (size_t) 32 * gpr + 16 * (4 * b + 2 * n + is) + (q & 15); }
    if (r < 208) return (size_t) 48 * gpr + 16 * b + (r - 192);
    return (size_t) 52 * gpr + 2 * b + (r - 208);
}

}  // namespace pkv
