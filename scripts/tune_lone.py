"""Tile time vs K for lone workgroups (<= 256 tiles, one per CU) and pairs (<= 512): fit t = c0 + u * k."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import _lib, ops
from volt_amd.synthetic import sde_batch
n = 4096
L = _lib.lib()
x, F, vol = sde_batch(64, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
f = ops.potrf(K, torch.full((64,), 0.6933, device="cuda"))
for per_cu in (1, 2):
    for var in ((1, 2, 3, 4, 5) if per_cu == 1 else (1, 4)):
        pts = []
        for k in (1, 2, 4, 8, 12, 16, 20):
            B = min(64, (256 * per_cu) // (31 - k))
            ts = []
            for rep in range(6):
                A = f.A[:B].clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.volt_tune_update_f32(A.data_ptr(), f.Winv.data_ptr(), f.info.data_ptr(), B, n, k, var, 1, _lib.stream_ptr()), "tune")
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            pts.append((k, np.median(ts[1:]), (31 - k) * B))
        ks = np.array([p[0] for p in pts], float); tt = np.array([p[1] for p in pts])
        u, c0 = np.polyfit(ks[3:], tt[3:], 1)
        print(f"per_cu={per_cu} var={var}: " + " ".join(f"k={int(k)}:{t:.0f}us({nt})" for k, t, nt in pts))
        print(f"     fit (k>=8): c0 = {c0:.1f} us, {u:.2f} us per K-block of 128 (MFMA peak: {6.826 * per_cu:.2f})")
