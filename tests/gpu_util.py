"""helpers shared by the -m gpu tests: build paged pools in the oracle's numpy form and mirror them on the device"""
import os

import numpy as np
import torch


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def make_paged_case(O, rng, *, lens, D, n_head, n_head_kv, type_k, type_v, block_size=128, spare_blocks=3, scale_data=1.0):
    """random K/V rows for sequences of the given lengths scattered into shuffled pool blocks (non-contiguous on purpose)"""
    nb = sum(-(-n // block_size) for n in lens) + spare_blocks
    perm = rng.permutation(nb).astype(np.int32)
    max_blocks = max(1, max(-(-n // block_size) for n in lens))
    kp, vp = O.pool_alloc(type_k, nb, n_head_kv, block_size, D), O.pool_alloc(type_v, nb, n_head_kv, block_size, D)
    bt = np.zeros((len(lens), max_blocks), np.int32)
    cur = 0
    for s, n in enumerate(lens):
        m = -(-n // block_size)
        bt[s, :m] = perm[cur:cur + m]
        cur += m
        if n == 0:
            continue
        kr = O.from_float(type_k, (rng.uniform(-1, 1, (n, n_head_kv, D)) * scale_data).astype(np.float32))
        vr = O.from_float(type_v, (rng.uniform(-1, 1, (n, n_head_kv, D)) * scale_data).astype(np.float32))
        idx = np.arange(n)
        O.pool_store(kp, type_k, 0, kr, bt[s][idx // block_size], idx % block_size, D=D, block_size=block_size)
        O.pool_store(vp, type_v, 1, vr, bt[s][idx // block_size], idx % block_size, D=D, block_size=block_size)
    return kp, vp, bt, np.array(lens, np.int32)


def nmse(a, b):
    """tests/test-backend-ops.cpp:177-190"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30))


class forced_tiles:
    """PKV_DECODE_NTILE for the launches inside the block (the library reads it per launch): tiles one decode workgroup walks; None = the library's choice"""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.old = os.environ.get("PKV_DECODE_NTILE")
        if self.n is None:
            os.environ.pop("PKV_DECODE_NTILE", None)
        else:
            os.environ["PKV_DECODE_NTILE"] = str(self.n)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("PKV_DECODE_NTILE", None)
        else:
            os.environ["PKV_DECODE_NTILE"] = self.old
