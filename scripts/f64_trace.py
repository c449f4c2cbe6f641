"""fp64 factorisation under rocprofv3: ten factorisations of 1 x 4096 and ten of 8 x 4096 (the kernel trace then holds the
one-launch kernel's durations: the two shapes are told apart by their grid).  usage: rocprofv3 --kernel-trace --stats -- python scripts/f64_trace.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

for B in (1, 8):
    n = 4096
    x, F, vol = sde_batch(min(B, 4), n)
    vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
    s2 = torch.full((B,), 0.6932, device="cuda", dtype=torch.float64)
    f = ops.potrf(K, s2)
    Aprep = f.A.clone()
    from volt_amd import _lib
    L = _lib.lib()
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, Aprep.data_ptr(), B, n, _lib.stream_ptr()), "prep")
    for _ in range(3):
        f.A.copy_(Aprep); ops.potrf_f64_inplace(f.A, f.Winv, f.info)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        f.A.copy_(Aprep)
        e0.record(); ops.potrf_f64_inplace(f.A, f.Winv, f.info); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{B} x {n} fp64 potrf: events around the call {np.mean(ts):.3f} ms (min {min(ts):.3f}), info {int(f.info.abs().sum())}", flush=True)
