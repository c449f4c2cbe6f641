"""Chained one-launch triangular solves (csrc/trsv.hip): ms per solve, forward and transposed, fp32 and fp64.
    python scripts/bench_trsv.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dt in (torch.float64, torch.float32):
    for B, n in ((1, 4096), (8, 4096), (1, 1024), (64, 2048)):
        x, F, vol = sde_batch(min(B, 4), n)
        vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]
        K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().to(dt), torch.tensor(x).cuda().to(dt), square=True))
        s2 = torch.full((B,), 0.05, device="cuda", dtype=dt)
        f = ops.potrf(K, s2)
        r = torch.randn(B, n, device="cuda", dtype=dt)
        z = ops.trsv(f, r)
        a = ops.trsv(f, z, transpose=True)
        Ks = K.double() + torch.diag_embed(s2.double()[:, None].expand(B, n))
        err = float(((Ks @ a.double()[..., None])[..., 0] - r.double()).abs().max() / r.abs().max())
        print(f"{str(dt)[6:]} {B:3d} x {n}: forward {timeit(lambda: ops.trsv(f, r)):.4f} ms, transposed {timeit(lambda: ops.trsv(f, z, transpose=True)):.4f} ms, residual |K a - r| / |r| {err:.1e}", flush=True)
