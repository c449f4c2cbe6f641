# a longer soak of the fp64 one launch after the round-6 chain work (tsl slices, staged operands, DPP pivots): factorisation AND
# step, both hand-off protocols, shapes with 2 .. 32 block columns, every result bitwise equal to the first
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from volt_amd import ops
from volt_amd.synthetic import sde_batch
F64 = torch.float64
SHAPES = [(1, 4096), (2, 4096), (3, 2048), (5, 1100), (7, 640), (8, 1024), (8, 4096), (9, 300), (16, 2048), (24, 700), (1, 256), (1, 384), (6, 129), (64, 399)]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for B, n in SHAPES:
    x, F, vol = sde_batch(min(B, 8), n)
    vol = np.tile(vol, (B // min(B, 8) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 8) + 1, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().to(F64), torch.tensor(x).cuda().to(F64), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda().to(F64)); r = (y - y.mean(-1, keepdim=True)).contiguous()
    s2 = torch.full((B,), 0.05, device="cuda", dtype=F64)
    ws = ops.MllWorkspace(B, n, True, K.device, F64)
    o0, a0, _ = ops.mll_step(K, r, s2, ws); o0, a0 = o0.clone(), a0.clone()
    f0 = ops.potrf(K, s2); L0, W0 = f0.A.clone(), f0.Winv.clone()
    t0, it, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        for _ in range(10):
            o, a, info = ops.mll_step(K, r, s2, ws)
            f = ops.potrf(K, s2)
            it += 2
        eq = [int(info.abs().sum()) == 0, int(f.info.abs().sum()) == 0, torch.equal(o, o0), torch.equal(a, a0), torch.equal(torch.tril(f.A), torch.tril(L0)), torch.equal(f.Winv, W0)]
        if not all(eq):
            bad += 1
            if bad == 1: print('   first mismatch: info, potrf info, out, alpha, L, Winv equal =', eq, ' max |dL|', float((torch.tril(f.A) - torch.tril(L0)).abs().max()), ' |dout|', float((o - o0).abs().max()), flush=True)
    print(f"{B:>3d} x {n:<5d}: {it} launches, {bad} mismatching checks", flush=True)
print("soak64 ok")
PY
