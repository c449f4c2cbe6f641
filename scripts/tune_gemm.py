"""A/B the panel-update kernel variants (interleaved rounds, one process).  GPU box only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import _lib, ops
from volt_amd.synthetic import sde_batch

B, n = 64, 4096
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0".split(","))]
ks = [1, 4, 16, 28]
x, F, vol = sde_batch(B, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
f = ops.potrf(K, torch.full((B,), 0.6933, device="cuda"))
A0 = f.A.clone()
L = _lib.lib()
reps = 5
res = {}
for rnd in range(4):
    for var in variants:
        for k in ks:
            A = A0.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.volt_tune_update_f32(A.data_ptr(), f.Winv.data_ptr(), f.info.data_ptr(), B, n, k, var, reps,
                                              _lib.stream_ptr()), "tune")
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = B * (n // 128 - k - 1) * (k + (0.625 if var == 0 else 0)) * 2 * 128 ** 3     # var 0: + the W product (160 of 256 MFMAs)
            if rnd > 0:
                res.setdefault((var, k), []).append(fl / ms / 1e9)
for var in variants:
    print("var", var, " ".join(f"k={k}: {np.median(res[(var, k)]):6.1f} TF (min {min(res[(var,k)]):.1f} max {max(res[(var,k)]):.1f})" for k in ks))
