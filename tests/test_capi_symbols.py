"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares; compute entry points
refuse to run without a device (there is no CPU fallback to fall into)."""
import ctypes as C
import re

import numpy as np
import pytest


def declared_functions(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    return sorted(set(re.findall(r"\b(pkv_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pkv):
    names = declared_functions(open(pkv.HEADER_PATH).read())
    assert len(names) > 40
    L = pkv.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.pkv_abi_version() == 3


def test_layout_helpers_match_reference_row_sizes(pkv, O):
    for t in (pkv.F16, pkv.Q8_0, pkv.Q4_0):
        for D in (64, 128):
            assert pkv.row_size(t, D) == O.row_size(t, D)
            assert pkv.tile_bytes(t, 128, D) == O.tile_bytes(t, 128, D) == 128 * pkv.row_size(t, D)
    # bytes per 1024-element KV row
    assert [pkv.row_size(t, 1024) for t in (pkv.F16, pkv.Q8_0, pkv.Q4_0)] == [2048, 1088, 576]
    assert pkv.pool_bytes(pkv.F16, 32, 8, 128, 128) == 8 * 1024 * 1024    # 8B ctx 4096: 8 MiB K per layer


def test_compute_refuses_without_device(pkv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert pkv.device_count() == 0
    a = pkv.AttnArgs()
    assert pkv.lib().pkv_attn_decode(C.byref(a), None) == pkv.ERR_NO_DEVICE
    assert b"no CPU path" in pkv.lib().pkv_last_error()
    x = np.zeros(4, np.int32).ctypes.data_as(C.c_void_p)
    assert pkv.lib().pkv_kv_store(x, x, x, x, pkv.F16, pkv.F16, x, x, x, None, 1, 1, 128, 128, None) == pkv.ERR_NO_DEVICE
    with pytest.raises(pkv.PkvError):
        pkv.PagedKVCache(n_blocks=4, n_layer=2)       # device pools need a device


def test_decode_launch_shapes(pkv, monkeypatch):
    """pkv_attn_decode_tiles (host logic, no device): how a decode launch is cut into workgroups for BASELINE's shapes — one 128-token tile per
    workgroup while the launch has fewer than two workgroups per CU, otherwise about one 8-wave workgroup per CU with at most 8 tiles each; a
    workgroup that covers its whole sequence means no partials and no merge launch (bench.py reports the choice)."""
    monkeypatch.delenv("PKV_DECODE_NTILE", raising=False)
    t = pkv.attn_decode_tiles               # (max_seq_len, n_head, n_head_kv, n_seq)
    assert t(4096, 32, 8, 1) == 1           # config 2: 32 x 8 = 256 workgroups, one per CU
    assert t(16384, 64, 8, 1) == 4          # config 5 (70B geometry, gqa 8: one resident workgroup per CU): 1024 tiles -> 256 workgroups of 4
    assert t(640, 32, 8, 32) == 5           # config 3's decode step: one workgroup per (sequence, kv head) walks all 5 tiles, no merge launch
    assert t(4096, 32, 8, 32) == 8          # 8192 tiles: capped (a long sequence among short ones must not serialise the launch)
    assert t(4096, 32, 8, 16) == 8 and t(4096, 32, 8, 8) == 4
    assert t(4096, 32, 8, 3) == 1 and t(4096, 32, 8, 6) == 1      # measured: 3 / 6 tiles per workgroup lose 20 % here (fewer, uneven workgroups)
    assert t(640, 32, 8, 24) == 1 and t(640, 32, 8, 16) == 1
    assert t(1, 32, 8, 1) == 1 and t(128, 4, 1, 1) == 1
    assert t(4096, 32, 2, 4) == t(4096, 32, 8, 4)                 # gqa 16 runs as gqa 8 over twice the kv heads
    for msl, nh, nkv, ns in ((100, 8, 2, 3), (5000, 32, 8, 7), (131072, 64, 8, 1), (640, 32, 8, 256)):
        n = t(msl, nh, nkv, ns)
        assert 1 <= n <= min(8, -(-msl // 128))
    monkeypatch.setenv("PKV_DECODE_NTILE", "3")                              # the experiments' override, clamped to the sequence's tiles
    assert t(4096, 32, 8, 1) == 3 and t(200, 32, 8, 1) == 2
