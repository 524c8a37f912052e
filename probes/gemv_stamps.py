#!/usr/bin/env python3
"""probes/gemv_stamps.py — where a decode-step product's time goes: per-wave s_memtime stamps of the diagnostic build probes/bin/libpkv_gemv_ablstamps.so
(probes/build_gemv_ablations.sh stamps; -DPKV_GEMV_STAMPS=2: stamps only, waits stay where the compiler put them).  Per shape the launch is run in a chain
of 8 eager launches over distinct weight buffers (the stamps of the last one are read); reported: microseconds after the wave's entry, median / max over
the waves, and the span of the launch (first wave in -> last wave out, 100 MHz wall clock).  Timing of a stamped build is not a measurement: read shares."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "probes"))
import torch
from gemv_shapes import load_pkg, SHAPES

NAMES = {2: "col+A requested", 3: "col landed (forced)", 4: "col share in LDS, B requested", 5: "barrier passed", 6: "unit 0 done", 7: "unit 1 done", 8: "unit 2 done", 9: "unit 3 done",
         10: "last A done", 11: "last B done", 12: "loop done", 13: "all acknowledged", 16: "(norm) block sums of squares done", 17: "(norm) barrier passed", 18: "(norm) rms scale known"}


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "probes", "bin", "libpkv_gemv_ablstamps.so")
    m = load_pkg("pkv_st", lib)
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(1)
    rs = {12: 144, 14: 210}
    stamps = torch.zeros((256, 16, 20), dtype=torch.int64, device=dev)
    os.environ["PKV_GEMV_STAMP_PTR"] = str(stamps.data_ptr())
    for name, mats, K, norm, resid, silu in SHAPES:
        nbytes = sum(M * (K // 256) * rs[t] for t, M in mats)
        n_buf = 8 if nbytes < 200e6 else 2
        ws = [[torch.randint(0, 256, (M * (K // 256) * rs[t],), dtype=torch.uint8, device=dev, generator=gen) for t, M in mats] for _ in range(n_buf)]
        x = torch.randn(K, device=dev, generator=gen)
        nw = torch.ones(K, device=dev) if norm else None
        outs = [torch.zeros(M, device=dev) for _, M in mats]
        ty = [t | 0x100 for t, _ in mats]
        Ms = [M for _, M in mats]
        for rep in range(3):
            stamps.zero_()
            torch.cuda.synchronize()
            torch.cuda._sleep(20_000_000)
            for i in range(n_buf):
                w = ws[i]
                if silu:
                    m.gemv_q(w, ty, Ms, K, None, [outs[0]], silu_mul=True, x=x, norm_w=nw, eps=1e-5)
                elif resid:
                    m.gemv_q(w, ty, Ms, K, None, [outs[0]], residual=outs[0], x=x)
                else:
                    m.gemv_q(w, ty, Ms, K, None, outs, x=x, norm_w=nw, eps=1e-5)
            torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 20)
        s = s[s[:, 1] != 0]
        clk = np.median((s[:, 13] - s[:, 1]) / np.maximum(1, (s[:, 14] - s[:, 0]))) * 100.0   # shader cycles per us
        span = (s[:, 14].max() - s[:, 0].min()) / 100.0
        spread = (s[:, 0].max() - s[:, 0].min()) / 100.0
        print(f"== {name}: {nbytes / 1e6:.1f} MB, {len(s)} waves, clock {clk:.0f} MHz, launch span {span:.2f} us, entry spread {spread:.2f} us, units per wave {int(s[:, 15].min())}-{int(s[:, 15].max())}")
        for i in (2, 3, 16, 17, 18, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
            v = s[:, i]
            ok = v != 0
            if ok.sum() == 0:
                continue
            d = (v[ok] - s[ok, 1]) / clk
            print(f"   {NAMES[i]:34s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us   ({ok.sum()} waves)")
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
