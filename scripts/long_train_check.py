"""300 Adam iterations of the reference loop (TrainVoltMagpieModel defaults) on the device vs the same
recursion driven by the fp64 oracle, N=399: reports the trajectories' divergence."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import volt_oracle as vo
from volt_amd.synthetic import sde_series
from volt_amd.train_utils import TrainVoltMagpieModel
from volt_amd.gp import ExactMarginalLogLikelihood
n, k, iters = 399, 25, 300
F, vol = sde_series(n, 2024)
tx = torch.arange(n, device="cuda") / 252.
model, lh = TrainVoltMagpieModel(tx, torch.tensor(F[1:]).cuda(), None, None, torch.tensor(vol).cuda(), train_iters=iters, k=k)
x = (np.arange(n) / 252.).astype(np.float32)
K = vo.volatility_kernel(x, np.exp(np.log(vol)).astype(np.float32))
y = np.log(F[1:]); mean = vo.ewma_mean(x, x, y, k)
raw = torch.tensor([1e-5], dtype=torch.float64, requires_grad=True)
opt = torch.optim.Adam([raw], lr=0.1)
for i in range(iters):
    opt.zero_grad()
    o = vo.mll_and_grads(K, y, mean, float(raw.detach()))
    raw.grad = torch.tensor([-o["d_raw"]], dtype=torch.float64)
    opt.step()
print("device raw_noise", float(lh.raw_noise), "oracle", float(raw), "sigma2 dev", float(lh.noise), "oracle", float(vo.noise_from_raw(float(raw))))
with torch.no_grad():
    model.train()
    last = float(-ExactMarginalLogLikelihood(lh, model)(model(tx), torch.tensor(F[1:]).cuda().log()))
print("final loss device", last, "oracle at oracle's params", -float(vo.mll_and_grads(K, y, mean, float(raw))["mll"]))
