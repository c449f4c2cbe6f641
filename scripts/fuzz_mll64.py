"""Randomised sweep of the fp64 MLL step (volt_mll_step_f64: one- / two-column look-ahead, the inverse's own look-ahead, K-sliced
and spread-out launches) against the fp64 oracle on two series per case, plus volt_trtri_f64 alone.  Exits non-zero on a miss."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import volt_oracle as vo

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst = {"mll": 0.0, "dsig": 0.0, "alpha": 0.0, "trtri": 0.0}
for c in range(cases):
    n = int(rng.choice([100, 129, 257, 384, 385, 500, 640, 897, 1025, 1500, 2049, 2900]))
    B = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17]))
    if n > 1100:
        B = min(B, 9)
    x, F, vol = sde_batch(B, n, seed=int(rng.randint(1, 10000)))
    raw = rng.uniform(-4, 1, size=B)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
    y = np.log(F[:, 1:]).astype(np.float64)
    mean = y.mean(-1, keepdims=True) + 0 * y
    s2 = torch.tensor([vo.noise_from_raw(r) for r in raw], dtype=torch.float64).cuda()
    o, a, info = ops.mll_step(K, torch.tensor(y - mean).cuda(), s2, want_grad=True)
    assert o.dtype == torch.float64 and int(info.abs().sum()) == 0, (n, B, info)
    o, a = o.cpu().numpy(), a.cpu().numpy()
    Kh = K.cpu().numpy()
    tol = 1e-9 if n <= 1100 else 1e-8
    for b in sorted({0, B - 1}):
        ref = vo.mll_and_grads(Kh[b], y[b], mean[b], float(raw[b]))
        e1 = abs(o[b, 0] - ref["mll"]) / max(1.0, abs(ref["mll"]))
        dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
        e2 = abs(o[b, 1] - dsig) / max(1.0, abs(dsig), abs(ref["aa"]) / n)
        e3 = np.abs(a[b] - ref["alpha"]).max() / np.abs(ref["alpha"]).max()
        worst["mll"], worst["dsig"], worst["alpha"] = max(worst["mll"], e1), max(worst["dsig"], e2), max(worst["alpha"], e3)
        assert e1 < tol and e2 < 100 * tol and e3 < 100 * tol, (n, B, b, e1, e2, e3)
    f = ops.potrf(K, s2)
    Yt = ops.trtri(f)
    b = B - 1
    L = np.linalg.cholesky(Kh[b] + float(s2[b]) * np.eye(n))
    refY = np.linalg.inv(L).T
    e4 = np.abs(Yt[b].cpu().numpy() - refY).max() / np.abs(refY).max()
    worst["trtri"] = max(worst["trtri"], e4)
    assert e4 < 100 * tol, (n, B, e4)
    print(f"case {c}: N={n} B={B} ok", flush=True)
print("cases", cases, "worst", worst)
