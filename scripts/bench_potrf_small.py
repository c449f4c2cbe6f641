"""ms per call of ops.potrf (copy-in + factorisation alone) at small batch sizes, N = 4096 (and the residual of L L' = K)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for B in (1, 2, 4, 8, 16, 24, 32, 64):
    x, F, vol = sde_batch(B, n)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    s2 = torch.full((B,), 0.6933, device="cuda")
    for _ in range(3): f = ops.potrf(K, s2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f = ops.potrf(K, s2)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    L = f.L[0].double()
    ref = K[0].double() + 0.6933 * torch.eye(n, device="cuda", dtype=torch.float64)
    err = float((L @ L.T - ref).abs().max() / ref.abs().max())
    print(f"B={B:2d} N={n}: {t*1e3:7.3f} ms per call ({t*1e3/B:.3f} per matrix)   |LL'-K|/|K| {err:.1e}  info {int(f.info.abs().sum())}")
