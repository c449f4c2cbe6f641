"""LearnGPCV at the reference's iteration count (train_utils.py default 1000; drivers use 200-400): HIP path vs the fp64
oracle loop from the same start values -- final loss, kernel parameter, mean constant and the extracted volatility
path exp(m) (the readout without its random draws)."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpcv_oracle as GO
from volt_amd.synthetic import sde_series
from volt_amd.train_utils import FitGPCV
warnings.simplefilter("ignore")
n, iters = 300, int(sys.argv[1]) if len(sys.argv) > 1 else 400
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"          # hipGraph-captured iterations (train_utils._run_iterations)
F, V = sde_series(n, 2024)
F = torch.tensor(F)
x = torch.arange(n, dtype=torch.float32) / 252
m0, _, _ = FitGPCV(x.cuda(), F.cuda(), train_iters=0)
d0 = m0.variational_strategy._variational_distribution
init = (d0.variational_mean.detach().cpu(), d0.chol_variational_covar.detach().cpu(), m0.mean_module.constant.detach().cpu().reshape(()))
torch.cuda.synchronize(); t0 = time.perf_counter()
model, lh, losses = FitGPCV(x.cuda(), F.cuda(), train_iters=iters, graph=graph)
torch.cuda.synchronize(); t_hip = time.perf_counter() - t0
rec = []
t0 = time.perf_counter()
ref, ps = GO.learn_gpcv(x, F, train_iters=iters, eps=torch.zeros(1, n, dtype=torch.float64), dtype=torch.float64, record=rec, init=init)
t_cpu = time.perf_counter() - t0
d = model.variational_strategy._variational_distribution
vol_hip = d.variational_mean.detach().exp().cpu().double()
print(f"iters {iters}{' (graph)' if graph else ''}: HIP {t_hip:.2f} s, fp64 oracle on the host {t_cpu:.1f} s")
print("final loss", float(losses[-1]), "oracle", rec[-1])
print("raw_vol", float(model.covar_module.raw_vol), "oracle", float(ps[3]), " constant", float(model.mean_module.constant), "oracle", float(ps[2]))
print("vol path exp(m): max rel dev", float(((vol_hip - ref) / ref).abs().max()), " corr with the SDE's true vol", float(torch.corrcoef(torch.stack([vol_hip, torch.tensor(V).double()]))[0, 1]))
