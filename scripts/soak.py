"""Soak: many repeated steps over shapes that exercise every hand-off protocol (one-launch long / short series, split-K
tickets, balanced-schedule slabs, two stream groups; round 5: the one-launch batched step's progress words in both hand-off
protocols -- batches that are a multiple of 8 and those that are not), checking info == 0 and BITWISE repeatability of every step.
    python scripts/soak.py [seconds per shape]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
F32, F64 = torch.float32, torch.float64
SHAPES = [(1, 4096, F32), (1, 1500, F32), (8, 399, F32), (40, 399, F32), (2, 4096, F32), (8, 4096, F32), (16, 4096, F32), (12, 2900, F32),
          (3, 3072, F32), (24, 1536, F32), (64, 2048, F32), (64, 4096, F32),
          # fp64: the one-launch step of csrc/batch64_step.hip (no atomics: repeatable bit for bit), both hand-off protocols
          (1, 4096, F64), (8, 2048, F64), (3, 1500, F64), (16, 700, F64)]
for B, n, dt in SHAPES:
    x, F, vol = sde_batch(min(B, 8), n)
    vol = np.tile(vol, (B // min(B, 8) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 8) + 1, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().to(dt), torch.tensor(x).cuda().to(dt), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda().to(dt)); r = (y - y.mean(-1, keepdim=True)).contiguous()
    s2 = torch.full((B,), 0.05, device="cuda", dtype=dt)
    ws = ops.MllWorkspace(B, n, True, K.device, dt)
    o0, a0, _ = ops.mll_step(K, r, s2, ws); o0, a0 = o0.clone(), a0.clone()
    t0, it, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        for _ in range(20):
            o, a, info = ops.mll_step(K, r, s2, ws)
            it += 1
        if int(info.abs().sum()) != 0 or not torch.equal(o, o0) or not torch.equal(a, a0):
            bad += 1
    print(f"{B:>3d} x {n:<5d} {str(dt)[6:]}: {it} steps, {bad} mismatching checks", flush=True)
    assert bad == 0
print("soak ok")
