"""Lone vs paired workgroups: one launch of exactly <= 256 tiles (one per CU) or <= 512 (two per CU) at block column k."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import _lib, ops
from volt_amd.synthetic import sde_batch
n = 4096
L = _lib.lib()
for B in (17, 34, 68):
    x, F, vol = sde_batch(B, n)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    f = ops.potrf(K, torch.full((B,), 0.6933, device="cuda"))
    for var in (0, 1):
        for k in (4, 16):
            ts = []
            for rep in range(6):
                A = f.A.clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.volt_tune_update_f32(A.data_ptr(), f.Winv.data_ptr(), f.info.data_ptr(), B, n, k, var, 1, _lib.stream_ptr()), "tune")
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            tiles = (n // 128 - k - 1) * B
            units = k + (0.625 if var == 0 else 0)
            t = np.median(ts[1:])
            ideal_lone = units * 2 * 128 ** 3 / (157.3e12 / 256) * 1e6
            print(f"B={B:3d} tiles={tiles:4d} var={var} k={k:2d}: {t:7.1f} us per launch; one tile alone at MFMA peak {ideal_lone:6.1f} us; "
                  f"TF/s {tiles * units * 2 * 128**3 / t / 1e6:6.1f}")
