"""fp32 HIP step vs the fp64 oracle at the BASELINE sizes (SURVEY 8d "tolerances to state").  GPU box."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import volt_oracle as vo
from volt_amd import ops
from volt_amd.synthetic import sde_batch

rows = []
for B, n, raw in [(4, 256, 1e-5), (2, 2048, 1e-5), (2, 4096, 1e-5), (2, 4096, -6.0), (2, 2048, -9.0)]:
    x, F, vol = sde_batch(B, n)
    V = ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True)
    K = ops.fill(V)
    y = np.log(F[:, 1:])
    mean = np.stack([vo.ewma_mean(x, x, y[b], 25) for b in range(B)])
    s2 = float(vo.noise_from_raw(raw))
    out, alpha, info = ops.mll_step(K, torch.tensor(y - mean).cuda(), torch.full((B,), s2, device="cuda"))
    out = out.cpu().numpy().astype(np.float64)
    Kc = K.cpu().numpy()
    t0 = time.time()
    o = vo.mll_and_grads(Kc, y.astype(np.float32), mean.astype(np.float32), raw)
    dsig = 0.5 * (o["aa"] - o["trinv"]) / n
    a = alpha.cpu().numpy()
    cond = [np.linalg.cond(Kc[b].astype(np.float64) + s2 * np.eye(n)) for b in range(1)]
    rows.append({"B": B, "N": n, "sigma2": round(s2, 6), "cond": float(f"{cond[0]:.3g}"), "info": info.cpu().tolist(),
                 "mll_rel": float(np.abs(out[:, 0] / o["mll"] - 1).max()),
                 "dsigma2_rel": float(np.abs(out[:, 1] / dsig - 1).max()),
                 "alpha_rel_to_max": float(np.abs(a - o["alpha"]).max() / np.abs(o["alpha"]).max()),
                 "trinv_rel": float(np.abs(out[:, 4] / o["trinv"] - 1).max()),
                 "oracle_s": round(time.time() - t0, 1)})
    print(json.dumps(rows[-1]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/accuracy_table.json", "w"), indent=1)
