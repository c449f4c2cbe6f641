"""hipGraph-captured training loops against the eager ones at the reference's sizes (one ticker, ntrain = 400,
400 iterations per stage: GenerateMultiMeanPreds.py:85-107).  Same start values; reports s per stage and the final
parameters of both."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import train_utils as tu
from volt_amd.synthetic import sde_series

n, iters = 399, 400
F, vol = sde_series(n, 7)
dev = "cuda"
tx = torch.arange(n, device=dev) / 252.
prices = torch.tensor(F, device=dev)
res = {}
for graph in (False, True):
    torch.manual_seed(0)
    t = {}
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = tu.LearnGPCV(tx, prices, train_iters=iters, graph=graph)
        torch.cuda.synchronize(); t["LearnGPCV"] = time.perf_counter() - t0; t0 = time.perf_counter()
        vmod, vlh = tu.TrainVolModel(tx, v, train_iters=iters, graph=graph)
        torch.cuda.synchronize(); t["TrainVolModel"] = time.perf_counter() - t0; t0 = time.perf_counter()
        m, lh = tu.TrainVoltMagpieModel(tx, prices[1:], vmod, vlh, v, train_iters=iters, k=300, graph=graph)
        torch.cuda.synchronize(); t["TrainVoltMagpieModel"] = time.perf_counter() - t0
    res[graph] = (t, v.clone(), float(lh.raw_noise), float(vmod.covar_module.raw_vol) if hasattr(vmod.covar_module, "raw_vol") else 0.0)
    print("graph" if graph else "eager", {k: round(x, 3) for k, x in t.items()}, "total %.3f s" % sum(t.values()),
          "raw_noise %.6f" % res[graph][2], "raw_vol %.6f" % res[graph][3])
print("max rel diff of the extracted vol path:", float(((res[True][1] - res[False][1]).abs() / res[False][1].abs()).max()))
