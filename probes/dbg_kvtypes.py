import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import _load_pkg
pkv = _load_pkg.load()
from oracle import cpu_oracle as O
for t, B in ((6, 22), (20, 18)):
    rng = np.random.default_rng(40 + t)
    x = (rng.standard_normal((513, 128)) * rng.choice([1e-3, 1.0, 50.0], (513, 1))).astype(np.float32)
    x[::17] = np.round(x[::17] * 4) / 4
    x[5] = 0.75
    x[6, :32] = -2.0
    x[7, 32:64] = np.arange(32, dtype=np.float32) - 16
    want = O.from_float(t, x)
    dst = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
    pkv.convert_f32(torch.from_numpy(x).cuda(), dst, t)
    got = dst.cpu().numpy().reshape(want.shape)
    w = want.reshape(-1, B); g = got.reshape(-1, B)
    bad = np.nonzero((w != g).any(axis=1))[0]
    print("type", t, "bad blocks", len(bad), "of", len(w), bad[:10])
    for b in bad[:4]:
        print(" block", b, "row", b // 4, "x", x.reshape(-1, 32)[b][:8], "...")
        print("  want", w[b].tolist())
        print("  got ", g[b].tolist())

# zero rows through the pool codecs of the existing types (the sign of d = max / -8 of an all-zero q4_0 block)
for t in (2, 8):
    x = np.zeros((4, 128), np.float32); x[1] = -0.0; x[2, :32] = 1.0
    want = O.from_float(t, x)
    dst = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
    pkv.convert_f32(torch.from_numpy(x).cuda(), dst, t)
    got = dst.cpu().numpy().reshape(want.shape)
    print("type", t, "zero rows equal:", np.array_equal(got, want), got[0][:4].tolist(), want[0][:4].tolist())
