"""-m gpu: pkv_convert_f32_paged (the KV-store CPY of the ggml-backend plugin once a cache tensor has its paged mirror) ==
pkv_convert_f32 into the flat tensor followed by pkv_pool_scatter_rows of those rows into the pool, bit for bit, for every KV type,
K and V layouts, head_dim 64 / 128 and cell ranges that start inside a block and cross a block boundary."""
import numpy as np
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tname", ["f16", "q8_0", "q4_0"])
@pytest.mark.parametrize("is_v", [False, True])
@pytest.mark.parametrize("D,HK,cell0,n_cells", [(128, 8, 4096 - 1, 1), (64, 4, 100, 60), (128, 2, 0, 130)])
def test_convert_paged_matches_convert_then_scatter(pkv, tname, is_v, D, HK, cell0, n_cells):
    t = pkv.TYPE_BY_NAME[tname]
    BS = 128
    nb = (cell0 + n_cells + BS - 1) // BS
    rng = np.random.default_rng(D + HK + cell0 + n_cells)
    x = dev(rng.standard_normal((n_cells, HK, D)).astype(np.float32))
    row_cell = pkv.row_size(t, D) * HK
    pool_bytes = pkv.pool_bytes(t, nb, HK, BS, D)
    # the reference sequence: convert into the flat tensor, scatter those rows into the pool
    flat_a = torch.zeros(n_cells * row_cell, dtype=torch.uint8, device="cuda")
    pool_a = torch.zeros(pool_bytes, dtype=torch.uint8, device="cuda")
    pkv.convert_f32(x.view(n_cells * HK, D), flat_a, t)
    cells = torch.arange(cell0, cell0 + n_cells, dtype=torch.int32, device="cuda")
    pkv.pool_scatter_rows(pool_a, flat_a, type=t, is_v=is_v, slot_block=(cells // BS).to(torch.int32), slot_off=(cells % BS).to(torch.int32),
                          n_head_kv=HK, head_dim=D, block_size=BS)
    # one launch
    flat_b = torch.zeros_like(flat_a)
    pool_b = torch.zeros_like(pool_a)
    pkv.convert_f32_paged(x, flat_b, t, pool_b, is_v=is_v, cell0=cell0, n_head_kv=HK, head_dim=D, block_size=BS)
    torch.cuda.synchronize()
    assert torch.equal(flat_a, flat_b)
    assert torch.equal(pool_a, pool_b)
    assert int(pool_b.to(torch.int64).sum()) != 0


@pytest.mark.parametrize("tname", ["f16", "q8_0", "q4_0"])
@pytest.mark.parametrize("D,HK,cell0,n_tok,mode,n_dims", [(128, 8, 4095, 1, 0, 128), (64, 4, 120, 20, 0, 64), (128, 2, 3, 5, 2, 128), (128, 2, 0, 3, 0, 64)])
def test_rope_convert_paged_matches_rope_then_convert(pkv, tname, D, HK, cell0, n_tok, mode, n_dims):
    """pkv_rope_convert_f32_paged == pkv_rope_f32 followed by pkv_convert_f32_paged, bit for bit (roped f32 rows, flat bytes, pool)"""
    t = pkv.TYPE_BY_NAME[tname]
    BS = 128
    nb = (cell0 + n_tok + BS - 1) // BS
    rng = np.random.default_rng(D + HK + cell0 + n_tok + mode)
    x = dev(rng.standard_normal((n_tok, HK, D)).astype(np.float32))
    pos = dev(np.arange(cell0, cell0 + n_tok, dtype=np.int32) * 3 + 1)
    rp = pkv.rope_params(n_dims=n_dims, mode=mode)
    row_cell = pkv.row_size(t, D) * HK
    flat_a = torch.zeros(n_tok * row_cell, dtype=torch.uint8, device="cuda")
    pool_a = torch.zeros(pkv.pool_bytes(t, nb, HK, BS, D), dtype=torch.uint8, device="cuda")
    roped_a = pkv.rope_f32(x, pos, rp)
    pkv.convert_f32_paged(roped_a, flat_a, t, pool_a, is_v=False, cell0=cell0, n_head_kv=HK, head_dim=D, block_size=BS)
    flat_b, pool_b, roped_b = torch.zeros_like(flat_a), torch.zeros_like(pool_a), torch.zeros_like(roped_a)
    pkv.rope_convert_f32_paged(x, pos, rp, flat_b, t, pool_b, cell0=cell0, block_size=BS, roped=roped_b)
    torch.cuda.synchronize()
    assert torch.equal(roped_a, roped_b)
    assert torch.equal(flat_a, flat_b) and torch.equal(pool_a, pool_b)


@pytest.mark.parametrize("tk,tv", [("f16", "f16"), ("q8_0", "q4_0"), ("q4_0", "q8_0")])
@pytest.mark.parametrize("D,H,HK,cell0,n_tok,mode", [(128, 32, 8, 4095, 1, 0), (64, 4, 4, 120, 20, 0), (128, 8, 2, 3, 5, 2)])
def test_qkv_store_paged_matches_the_three_launches(pkv, tk, tv, D, H, HK, cell0, n_tok, mode):
    """pkv_qkv_store_f32_paged == pkv_rope_f32(q), pkv_rope_convert_f32_paged(k), pkv_convert_f32_paged(v), bit for bit"""
    TK, TV = pkv.TYPE_BY_NAME[tk], pkv.TYPE_BY_NAME[tv]
    BS = 128
    nb = (cell0 + n_tok + BS - 1) // BS
    rng = np.random.default_rng(D + H + cell0 + n_tok + mode)
    q = dev(rng.standard_normal((n_tok, H, D)).astype(np.float32))
    k = dev(rng.standard_normal((n_tok, HK, D)).astype(np.float32))
    v = dev(rng.standard_normal((n_tok, HK, D)).astype(np.float32))
    pos = dev(np.arange(cell0, cell0 + n_tok, dtype=np.int32) * 2 + 5)
    rp = pkv.rope_params(n_dims=D, mode=mode)

    def bufs():
        return (torch.zeros(n_tok * pkv.row_size(TK, D) * HK, dtype=torch.uint8, device="cuda"), torch.zeros(pkv.pool_bytes(TK, nb, HK, BS, D), dtype=torch.uint8, device="cuda"),
                torch.zeros(n_tok * pkv.row_size(TV, D) * HK, dtype=torch.uint8, device="cuda"), torch.zeros(pkv.pool_bytes(TV, nb, HK, BS, D), dtype=torch.uint8, device="cuda"))
    kf_a, kp_a, vf_a, vp_a = bufs()
    q_a = pkv.rope_f32(q, pos, rp)
    pkv.rope_convert_f32_paged(k, pos, rp, kf_a, TK, kp_a, cell0=cell0, block_size=BS)
    pkv.convert_f32_paged(v, vf_a, TV, vp_a, is_v=True, cell0=cell0, n_head_kv=HK, head_dim=D, block_size=BS)
    kf_b, kp_b, vf_b, vp_b = bufs()
    q_b = torch.zeros_like(q_a)
    pkv.qkv_store_f32_paged(q, k, v, pos, rp, q_out=q_b, k_dst=kf_b, k_pool=kp_b, v_dst=vf_b, v_pool=vp_b, type_k=TK, type_v=TV, cell0=cell0, block_size=BS)
    torch.cuda.synchronize()
    assert torch.equal(q_a, q_b)
    assert torch.equal(kf_a, kf_b) and torch.equal(kp_a, kp_b)
    assert torch.equal(vf_a, vf_b) and torch.equal(vp_a, vp_b)
