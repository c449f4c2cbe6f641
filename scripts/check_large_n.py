"""Sizes beyond the BASELINE ones: N = 8192 and N = 12288 (n = 64 / 96 block columns), device-side fp64 residual checks."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
for B, n in ((8, 8192), (2, 12288), (1, 16384)):
    x, F, vol = sde_batch(B, n)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda())
    r = (y - y.mean(-1, keepdim=True)).float()
    s2 = torch.full((B,), 0.5, device="cuda")
    o, a, info = ops.mll_step(K, r, s2, want_grad=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o, a, info = ops.mll_step(K, r, s2, want_grad=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    back = torch.stack([(K[b].double() @ a[b].double()) + 0.5 * a[b].double() for b in range(B)])
    rel = ((back - r.double()).norm(dim=-1) / r.double().norm(dim=-1)).max().item()
    print(f"B={B} N={n}: {dt*1e3:.1f} ms/step = {B*2*n**3/3/dt/1e12:.1f} TF/s  info={int(info.abs().sum())}  "
          f"|(K+s2 I) alpha - r|/|r| = {rel:.2e}  mll={o[:, 0].tolist()[:2]}")
    assert int(info.abs().sum()) == 0 and rel < 5e-3
    del K, back
    torch.cuda.empty_cache()
