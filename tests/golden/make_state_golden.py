#!/usr/bin/env python3
"""Generate tests/golden/state_io.npz: session-state files written by the REFERENCE's own llama_kv_cache_unified
(state_write, /root/reference/src/llama-kv-cache.cpp:848-1001), compiled in place (oracle/_ref/libref_kv.so, driver
oracle/ref_kv_driver.cpp), together with the ring content they were written from.

Run in the dev container only:   python tests/golden/make_state_golden.py
Per case `<name>`:  __blob (whole cache, seq_id -1), __blob_seq1 (sequence 1 alone), __cell_pos / __cell_nseq / __cell_seqs
(the ring's cells), __k<il> / __v<il> (the flat per-layer tensors as [kv_size][row bytes] rows; transposed V is stored here
as rows too), __meta = (type_k, type_v, v_trans, n_layer, n_head_kv, head_dim, kv_size)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_oracle as O  # noqa: E402
from oracle import ref_kv  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
L, HK, D, SIZE = 2, 1, 64, 256


def build(tk, tv, fa, seed):
    rng = np.random.default_rng(seed)
    r = ref_kv.RefKvCache(n_layer=L, n_head_kv=HK, head_dim=D, type_k=tk, type_v=tv, kv_size=SIZE, flash_attn=fa)
    # a 130-token prompt shared by two sequences, six interleaved decode steps, one removed range
    r.find_slot([0] * 130, list(range(130))); r.advance_head(130)
    r.seq_cp(0, 1, -1, -1)
    for t in range(6):
        r.find_slot([0, 1], [130 + t, 130 + t]); r.advance_head(2)
    r.seq_rm(1, 100, 104)
    n_used = 142
    k = [np.zeros((SIZE, O.row_size(tk, HK * D)), np.uint8) for _ in range(L)]
    v = [np.zeros((SIZE, O.row_size(tv, HK * D)), np.uint8) for _ in range(L)]
    for il in range(L):
        k[il][:n_used] = O.from_float(tk, rng.uniform(-1, 1, (n_used, HK * D)).astype(np.float32))
        v[il][:n_used] = O.from_float(tv, rng.uniform(-1, 1, (n_used, HK * D)).astype(np.float32))
        r.tensor_set(il, False, k[il])
        r.tensor_set(il, True, v[il] if fa else np.ascontiguousarray(v[il].view(np.uint16).T))
    cells = r.cells(max_seq=4)
    out = {
        "blob": np.frombuffer(r.state_write(-1), np.uint8), "blob_seq1": np.frombuffer(r.state_write(1), np.uint8),
        "cell_pos": np.array([c[0] for c in cells], np.int32), "cell_nseq": np.array([len(c[2]) for c in cells], np.int32),
        "cell_seqs": np.array([c[2] + [-1] * (4 - len(c[2])) for c in cells], np.int32),
        "meta": np.array([tk, tv, int(not fa), L, HK, D, SIZE], np.int32),
    }
    for il in range(L):
        out[f"k{il}"], out[f"v{il}"] = k[il], v[il]
    return out


def main():
    if not ref_kv.available():
        raise SystemExit("oracle/_ref/libref_kv.so missing: run `make -C oracle ref` (needs /root/reference)")
    cases = {"f16_fa": (O.F16, O.F16, True, 11), "q8_0_fa": (O.Q8_0, O.Q8_0, True, 12), "q4_0k_q8_0v_fa": (O.Q4_0, O.Q8_0, True, 13),
             "f16_vtrans": (O.F16, O.F16, False, 14)}
    flat = {}
    for name, spec in cases.items():
        for key, val in build(*spec).items():
            flat[f"{name}__{key}"] = val
    flat["names"] = np.array(list(cases))
    np.savez_compressed(os.path.join(OUT, "state_io.npz"), **flat)
    print("wrote state_io.npz:", {k: v.shape for k, v in flat.items() if k.endswith("__blob")})


if __name__ == "__main__":
    main()
