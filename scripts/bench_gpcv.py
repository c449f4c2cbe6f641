"""GPCV stage (LearnGPCV, voltron/train_utils.py:15-67): time of one ELBO+gradient step on the HIP path, per
configuration, next to the torch-CPU oracle on this host.  Algorithmic flops per series: 2N^3/3 (factor + L^-T)
+ N^3/3 (T' = Lq' L^-T) + 2N^3/3 (G = Y T') = 5N^3/3."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import gpcv_oracle as GO

dev = "cuda:0"
cpu = "--no-cpu" not in sys.argv
for B, n, reps in ((1, 399, 50), (64, 399, 30), (8, 2048, 10), (8, 4096, 5), (32, 4096, 3)):
    x, F, _ = sde_batch(B, n, seed=5)
    xt = torch.tensor(x, dtype=torch.float64)
    yy = torch.stack([GO.scaled_returns(xt, torch.tensor(F[b], dtype=torch.float64)) for b in range(B)])
    f = yy.abs().clamp_min(1e-2).log()
    K = GO.bm_cov(xt, torch.tensor(0.2, dtype=torch.float64)).float().to(dev).expand(B, n, n).contiguous()
    g = torch.Generator().manual_seed(0)
    Lq = (0.05 * torch.eye(n) + 0.001 * torch.randn(n, n, generator=g)).tril().to(dev).expand(B, n, n).contiguous()
    m = f.float().to(dev)
    mu = torch.full((B, n), -1.5, device=dev)
    y = yy.float().to(dev)
    gx, gw = GO.gauss_hermite(75)
    gx, gw = gx.float().to(dev), (gw / math.sqrt(math.pi)).float().to(dev)
    ws = ops.gpcv_step(K, m - mu, m, Lq, y, gx, gw, w_ell=1 / n, w_kl=1 / n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.gpcv_step(K, m - mu, m, Lq, y, gx, gw, ws, w_ell=1 / n, w_kl=1 / n)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    fl = B * 5 * n ** 3 / 3
    line = f"B={B} N={n}: {t*1e3:.3f} ms/step  {fl/t/1e12:.1f} TFLOP/s (algorithmic 5N^3/3)"
    if cpu and B * n ** 3 <= 8 * 2048 ** 3:
        torch.set_num_threads(min(32, os.cpu_count()))
        raw = torch.logit(torch.tensor([0.2]))
        t0 = time.perf_counter()
        GO.elbo_and_grads(m[0].cpu(), Lq[0].cpu(), mu[0, :1].cpu(), raw, xt.float(), yy[0].float())
        c = (time.perf_counter() - t0) * B
        line += f"   torch-CPU oracle {c*1e3:.1f} ms/step ({torch.get_num_threads()} threads)  x{c/t:.0f}"
    print(line, flush=True)
