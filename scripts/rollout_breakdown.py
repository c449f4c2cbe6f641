"""Where C5's share of the rollouts (8 series x 10^4 paths x 256 steps at N = 4096, the reference's factor route) spends its
wall time: run under `rocprofv3 --kernel-trace --stats` for the kernels; prints the wall time of rollout_series itself.
    python scripts/rollout_breakdown.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import rollout_engine as re_
from volt_amd.synthetic import sde_batch

G, S, H, n, K_EWMA = 8, 10000, 256, 4096, 400
dev = "cuda"
x, F, vol = sde_batch(G, n)
tx = torch.tensor(x, device=dev)
logy = torch.log(torch.tensor(F[:, 1:], device=dev))
lv = torch.log(torch.tensor(vol, device=dev))
test_x = tx[-1] + (tx[1] - tx[0]) * torch.arange(1, H + 1, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
pvd = torch.full((G, S, H), float(vol.mean()), device=dev)
zd = torch.randn(G, S, H, device=dev, generator=g)
for solve in ("factor", "closed"):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samples, info = re_.rollout_series(tx, logy, lv, test_x, pvd, zd, 0, K_EWMA, solve=solve)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"solve={solve}: wall {min(ts) * 1e3:.2f} ms (median {np.median(ts) * 1e3:.2f}), non-PD paths {int((info != 0).sum())}", flush=True)
