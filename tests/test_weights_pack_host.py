"""CPU: the plane layout of K-quant weight rows (csrc/pkv_wlayout.h) as the host restatement `pack_rows_host` states it — a permutation of each row's
own bytes, invertible, every plane where the header comment says it is.  The device kernel is checked against this restatement in
tests/test_gpu_weights_pack.py; the C mapping (pk_offset_q4k / _q5k / _q6k) itself is compiled here with g++ and compared byte for byte."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("t,K", [(12, 256), (12, 4096), (12, 14336), (12, 768), (14, 256), (14, 4096), (14, 14336), (14, 5632),
                                 (13, 256), (13, 4096), (13, 14336), (13, 768)])
def test_pack_is_a_row_permutation_and_inverts(pkv, t, K):
    rb = K // 256 * {12: 144, 13: 176, 14: 210}[t]
    rng = np.random.default_rng(K + t)
    rows = rng.integers(0, 256, (5, rb), dtype=np.uint8)
    pk = pkv.pack_rows_host(rows, t, K)
    assert pk.shape == rows.shape
    assert np.array_equal(np.sort(pk, axis=1), np.sort(rows, axis=1))           # the same bytes, row by row
    assert np.array_equal(pkv.pack_rows_host(pk, t, K, inverse=True), rows)
    # position-tagged rows: where does canonical byte o go?
    tag = np.arange(rb, dtype=np.int64)
    lo, hi = pkv.pack_rows_host((tag & 255).astype(np.uint8)[None], t, K)[0], pkv.pack_rows_host((tag >> 8).astype(np.uint8)[None], t, K)[0]
    src = lo.astype(np.int64) | (hi.astype(np.int64) << 8)                      # src[p] = canonical offset of packed byte p
    gpr = K // 64
    if t == 12:
        for g in (0, 1, gpr - 1):
            b, p = g // 4, g % 4
            assert list(src[16 * g:16 * g + 16]) == list(range(144 * b + 16 + 32 * p, 144 * b + 32 + 32 * p))
            assert list(src[16 * gpr + 16 * g:16 * gpr + 16 * g + 16]) == list(range(144 * b + 32 + 32 * p, 144 * b + 48 + 32 * p))
        assert list(src[32 * gpr:32 * gpr + 16]) == list(range(16))
    elif t == 13:
        for g in (0, 1, gpr - 1):
            b, p = g // 4, g % 4
            assert list(src[16 * g:16 * g + 16]) == list(range(176 * b + 48 + 32 * p, 176 * b + 64 + 32 * p))
            assert list(src[16 * gpr + 16 * g:16 * gpr + 16 * g + 16]) == list(range(176 * b + 64 + 32 * p, 176 * b + 80 + 32 * p))
        lb = gpr // 4 - 1                                                        # qh halves and header of the row's last block
        assert list(src[32 * gpr + 16 * lb:][:16]) == list(range(176 * lb + 16, 176 * lb + 32))
        assert list(src[36 * gpr + 16 * lb:][:16]) == list(range(176 * lb + 32, 176 * lb + 48))
        assert list(src[40 * gpr + 16 * lb:][:16]) == list(range(176 * lb, 176 * lb + 16))
    else:
        for g in (0, 3, gpr - 2):
            b, n, i = g // 4, (g % 4) // 2, g % 2
            assert list(src[16 * g:16 * g + 16]) == list(range(210 * b + 64 * n + 16 * i, 210 * b + 64 * n + 16 * i + 16))
            assert list(src[16 * gpr + 16 * g:][:16]) == list(range(210 * b + 64 * n + 32 + 16 * i, 210 * b + 64 * n + 48 + 16 * i))
            assert list(src[32 * gpr + 16 * g:][:16]) == list(range(210 * b + 128 + 32 * n + 16 * i, 210 * b + 144 + 32 * n + 16 * i))
        assert list(src[48 * gpr:48 * gpr + 16]) == list(range(192, 208))
        assert list(src[52 * gpr:52 * gpr + 2]) == [208, 209]


def test_c_offset_map_matches_host_restatement(pkv):
    """pk_offset_q4k / pk_offset_q6k of csrc/pkv_wlayout.h (what the device kernels index with), compiled for the host"""
    src = r'''
#include "pkv_wlayout.h"
extern "C" void map_row(int type, int K, long long * out) {
    const int gpr = K / 64; const long long rb = (long long) (K / 256) * (type == 12 ? 144 : (type == 13 ? 176 : 210));
    for (long long o = 0; o < rb; ++o) out[o] = (long long) (type == 12 ? pkv::pk_offset_q4k(gpr, (size_t) o) : (type == 13 ? pkv::pk_offset_q5k(gpr, (size_t) o) : pkv::pk_offset_q6k(gpr, (size_t) o)));
}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "m.cpp"), "w").write(src)
        so = os.path.join(d, "m.so")
        subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "llama.cpp_test_amd", "csrc"), os.path.join(d, "m.cpp"), "-o", so])
        L = ctypes.CDLL(so)
        for t, K in ((12, 4096), (12, 768), (14, 4096), (14, 5632), (13, 4096), (13, 768)):
            rb = K // 256 * {12: 144, 13: 176, 14: 210}[t]
            out = np.zeros(rb, np.int64)
            L.map_row(t, K, out.ctypes.data_as(ctypes.c_void_p))
            assert sorted(out) == list(range(rb))
            tag = np.arange(rb, dtype=np.int64)
            lo, hi = pkv.pack_rows_host((tag & 255).astype(np.uint8)[None], t, K)[0], pkv.pack_rows_host((tag >> 8).astype(np.uint8)[None], t, K)[0]
            src_of = lo.astype(np.int64) | (hi.astype(np.int64) << 8)
            assert np.array_equal(src_of[out], tag)                              # packed[map(o)] holds canonical byte o
