"""Rollout engine timing on the GPU box (BASELINE config 5 per-GPU share: 8 series x 10k paths x 256 steps
at N=4096 by default): the append-only engine (default) and, once, the full re-substitution (the round-2 engine,
H^3/6 * 4 redundant bytes per path) -- under `rocprofv3 --kernel-trace --stats` both kernels show up by name."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import rollout_engine as re_
from volt_amd.synthetic import sde_batch, rollout_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--series", type=int, default=8)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--samples", type=int, default=10000)
ap.add_argument("--horizon", type=int, default=256)
ap.add_argument("--k", type=int, default=25)
a = ap.parse_args()
G, n, S, H = a.series, a.n, a.samples, a.horizon
x, F, vol = sde_batch(G, n)
pv, z = rollout_inputs(vol[:, -1], S, H, seed=3)
dev = "cuda"
tx = torch.tensor(x, device=dev)
test_x = torch.arange(H, device=dev) / 252. + tx[-1] + tx[1]
logy = torch.log(torch.tensor(F[:, 1:], device=dev))
lv = torch.log(torch.tensor(vol, device=dev))
pvd, zd = torch.tensor(pv, device=dev), torch.tensor(z, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, info = re_.rollout_series(tx, logy, lv, test_x, pvd, zd, 0, a.k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
# kernel alone
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
bad = int((info != 0).sum())
# exact conditional check on a few paths (fp64 recursion)
from oracle import volt_oracle as vo
err = 0.0
for g in range(min(G, 2)):
    ly = np.log(F[g, 1:]).astype(np.float32)
    for s in range(2):
        ys = ly.copy()
        for i in range(H):
            full = vo.ewma(ys[-(a.k + 2):] if len(ys) > a.k + 2 else ys, a.k)
            val = (ys[-1] - full[-2]) + full[-1] + np.sqrt(0.5 / 252. * float(pv[g, s, i]) ** 2) * z[g, s, i]
            err = max(err, abs(val - float(samples[g, s, i])))
            ys = np.append(ys, np.float32(samples[g, s, i].item()))
torch.cuda.synchronize(); t0 = time.perf_counter()
s2, i2 = re_.rollout_series(tx, logy, lv, test_x, pvd, zd, 0, a.k, resubstitute=True)
torch.cuda.synchronize(); dt_r = time.perf_counter() - t0
print(json.dumps({"series": G, "N": n, "samples": S, "horizon": H, "total_s": round(dt, 5),
                  "sample_steps_per_s": round(G * S * H / dt), "non_pd_paths": bad,
                  "max_abs_dev_from_exact_one_step": err,
                  "algorithmic_GB": round(G * S * H * 12 / 1e9, 3),
                  "resubstitute_total_s": round(dt_r, 4), "resubstitute_redundant_GB": round(G * S * H ** 3 / 6 * 4 / 1e9, 2),
                  "max_abs_dev_resubstitute": float((samples - s2).abs().max())}))
