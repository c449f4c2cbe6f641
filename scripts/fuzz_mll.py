"""Randomised sweep of the fused MLL step against the fp64 oracle: odd sizes, ragged padding, batches that do and do
not divide into stream groups, wide range of noise levels.  Prints the worst relative errors; exits non-zero on a miss."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import volt_oracle as vo

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {"mll": 0.0, "dsig": 0.0, "alpha": 0.0}
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for c in range(cases):
    n = int(rng.choice([2, 3, 5, 31, 32, 33, 64, 100, 127, 128, 129, 160, 255, 256, 257, 300, 383, 385, 500, 640, 700]))
    B = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 32, 33]))
    if n >= 500:
        B = min(B, 9)
    x, F, vol = sde_batch(B, n, seed=int(rng.randint(1, 10000)))
    raw = rng.uniform(-6, 2, size=B)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    y = np.log(F[:, 1:])
    mean = y.mean(-1, keepdims=True) + 0 * y
    s2 = torch.tensor([vo.noise_from_raw(r) for r in raw], dtype=torch.float32).cuda()
    o, a, info = ops.mll_step(K, torch.tensor(y - mean).float().cuda(), s2, want_grad=True)
    assert int(info.abs().sum()) == 0, (n, B, info)
    o, a = o.cpu().double().numpy(), a.cpu().double().numpy()
    Kh = K.cpu().double().numpy()
    for b in range(B):
        ref = vo.mll_and_grads(Kh[b], y[b], mean[b], float(raw[b]))
        e1 = abs(o[b, 0] - ref["mll"]) / max(1.0, abs(ref["mll"]))
        dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
        e2 = abs(o[b, 1] - dsig) / max(1e-6, abs(dsig), 1e-4 * ref["trinv"] / n)
        e3 = np.abs(a[b] - ref["alpha"]).max() / np.abs(ref["alpha"]).max()
        worst["mll"], worst["dsig"], worst["alpha"] = max(worst["mll"], e1), max(worst["dsig"], e2), max(worst["alpha"], e3)
        if e1 > 5e-5 or e2 > 5e-3 or e3 > 2e-3:
            print("MISS", n, B, b, raw[b], e1, e2, e3)
            sys.exit(1)
print("cases", cases, "worst", worst)
