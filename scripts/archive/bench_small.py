"""Single-series training at the reference's default size (ntrain=400 -> N=399): HIP step vs the
torch-CPU restatement on this host, same loop body as TrainVoltMagpieModel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd.synthetic import sde_series
from volt_amd.train_utils import TrainVoltMagpieModel
from oracle import torch_cpu_path as tp
from oracle import volt_oracle as vo
for n in (399, 1023, 4096):
    F, vol = sde_series(n, 1)
    tx = torch.arange(n, device="cuda") / 252.
    iters = 50
    TrainVoltMagpieModel(tx, torch.tensor(F[1:]).cuda(), None, None, torch.tensor(vol).cuda(), train_iters=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    TrainVoltMagpieModel(tx, torch.tensor(F[1:]).cuda(), None, None, torch.tensor(vol).cuda(), train_iters=iters)
    torch.cuda.synchronize(); g = (time.perf_counter() - t0) / iters
    x = (np.arange(n) / 252.).astype(np.float32)
    K = torch.tensor(vo.volatility_kernel(x, vol))
    y = torch.tensor(np.log(F[1:])); m = torch.tensor(vo.ewma_mean(x, x, np.log(F[1:]), 25))
    raw = torch.full((1,), 1e-5, requires_grad=True)
    torch.set_num_threads(os.cpu_count())
    tp.mll_step(K, y, m, raw)
    reps = 5 if n < 2000 else 2
    t0 = time.perf_counter()
    for _ in range(reps): tp.mll_step(K, y, m, raw)
    c = (time.perf_counter() - t0) / reps
    print(f"N={n}: HIP loop {g*1e3:.3f} ms/iter (incl. Python/torch glue)   torch-CPU {c*1e3:.1f} ms/iter on {os.cpu_count()} cores   x{c/g:.0f}")
