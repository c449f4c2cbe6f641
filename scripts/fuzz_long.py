"""Randomised sweep of ONE series through the one-launch step (long_step_kernel with the split spine: every number of block
columns from 3 to 32, ragged sizes, a wide range of noise levels) against the fp64 oracle, on a workspace that is reused
from case to case within a block-column count (the step counter and flags carry over) -- and every case twice: the same bits.
Prints the worst relative errors; exits non-zero on a miss.  Usage: fuzz_long.py [seed] [cases per block-column count]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import volt_oracle as vo

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
worst = {"mll": 0.0, "dsig": 0.0, "alpha": 0.0}
cases = 0
for nb in range(3, 33):
    for c in range(per):
        n = int(rng.randint(128 * (nb - 1) + 1, 128 * nb + 1))
        x, F, vol = sde_batch(1, n, seed=int(rng.randint(1, 10000)))
        raw = rng.uniform(-6, 2, size=1)
        K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
        y = np.log(F[:, 1:])
        mean = y.mean(-1, keepdims=True) + 0 * y
        s2 = torch.tensor([vo.noise_from_raw(r) for r in raw], dtype=torch.float32).cuda()
        r = torch.tensor(y - mean).float().cuda()
        ws = ops.MllWorkspace(1, n, True, K.device)
        o1 = ops.mll_step(K, r, s2, ws)[0].clone()
        a1 = ws.alpha.clone()
        assert int(ws.info.abs().sum()) == 0, (n, ws.info)
        for _ in range(2):
            o2 = ops.mll_step(K, r, s2, ws)[0]
            assert torch.equal(o1, o2) and torch.equal(a1, ws.alpha), ("not repeatable", n)
        o, a = o1.cpu().double().numpy(), a1.cpu().double().numpy()
        ref = vo.mll_and_grads(K[0].cpu().double().numpy(), y[0], mean[0], float(raw[0]))
        e1 = abs(o[0, 0] - ref["mll"]) / max(1.0, abs(ref["mll"]))
        dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
        e2 = abs(o[0, 1] - dsig) / max(1e-6, abs(dsig), 1e-4 * ref["trinv"] / n)
        e3 = np.abs(a[0] - ref["alpha"]).max() / np.abs(ref["alpha"]).max()
        worst["mll"], worst["dsig"], worst["alpha"] = max(worst["mll"], e1), max(worst["dsig"], e2), max(worst["alpha"], e3)
        cases += 1
        if e1 > 5e-5 or e2 > 5e-3 or e3 > 2e-3:
            print("MISS", n, raw[0], e1, e2, e3)
            sys.exit(1)
print("cases", cases, "(one series, 3 .. 32 block columns) worst", worst)
