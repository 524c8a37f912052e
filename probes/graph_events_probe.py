"""Does hipExtLaunchKernel (events bound to a launch) survive stream capture + graph replay?  (measurement probe)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _load_pkg
import bench
pkv = _load_pkg.load()
w = bench.Workload(pkv, torch, ctx=4096, n_seq=1, type_k=pkv.F16, type_v=pkv.F16, seed=1)
L = 32
timer = pkv.KernelTimer(L)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    w.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            for l in range(L):
                pkv.qkv_prepare(w.q[l], w.k_cur[l], w.v_cur[l], w.k_pools[l], w.v_pools[l], type_k=w.tk, type_v=w.tv,
                                slot_block=w.slot_block, slot_off=w.slot_off, block_size=128, pos=w.pos, rope=w.rope)
                pkv.attn_decode(w.q[l], w.k_pools[l], w.v_pools[l], w.block_table, w.seq_lens, n_head_kv=8, type_k=w.tk,
                                type_v=w.tv, block_size=128, scale=w.scale, max_seq_len=4096, out=w.out[l], workspace=w.ws,
                                events=timer.pair(l))
        print("capture ok")
        res = []
        for r in range(20):
            g.replay()
            torch.cuda.synchronize()
            res.append(timer.elapsed_ms())
        a = np.array(res[2:]) * 1e3
        print("per-launch us inside graph replay: mean %.3f median %.3f min %.3f" % (a.mean(), np.median(a), a.min()))
    except Exception as e:
        print("capture with bound events failed:", repr(e)[:300])
