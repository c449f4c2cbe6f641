"""Measured fp32-vs-fp64-oracle error of the MLL+grad step in the end-of-training noise regime at the BASELINE sizes
(the table behind the tolerances of tests/test_gpu_lownoise.py), with the vendor's fp32 potrf + cholesky_solve on the same
matrices beside it.   python scripts/accuracy_lownoise.py > profiles/r04/accuracy_lownoise.txt"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import volt_oracle as vo

print(f"{'BxN':>8s} {'row':>3s} {'raw':>6s} {'sigma2':>9s} | {'mll rel':>9s} {'dsig rel':>9s} {'trinv rel':>9s} {'alpha/max':>9s} | vendor fp32: {'mll rel':>9s} {'alpha/max':>9s} {'factor hip/ven':>14s}")
for B, n, rows in [(2, 4096, (0, 1)), (8, 4096, (0, 7)), (64, 2048, (0, 32, 63)), (64, 4096, (0, 63))]:
    x, F, vol = sde_batch(B, n)
    Kd = ops.fill(ops.cumtrapz(torch.as_tensor(vol).cuda(), torch.as_tensor(x).cuda(), square=True))
    y = torch.log(torch.as_tensor(F[:, 1:]).cuda())
    ymean = y.mean(-1, keepdim=True).expand_as(y)
    r = (y - ymean).float()
    for raw in (1e-5, -3.0, -6.0, -9.0, -11.8):
        s2v = vo.noise_from_raw(raw)
        s2 = torch.full((B,), s2v, dtype=torch.float32).cuda()
        o, a, info = ops.mll_step(Kd, r, s2, want_grad=True)
        assert int(info.abs().sum()) == 0
        o, a = o.cpu().double().numpy(), a.cpu().double().numpy()
        f = ops.potrf(Kd, s2)
        for b in rows:
            ref = vo.mll_and_grads(Kd[b].cpu().numpy(), y[b].cpu().numpy(), ymean[b].cpu().numpy(), raw)
            dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
            Kb = Kd[b].double() + float(s2[b]) * torch.eye(n, device="cuda", dtype=torch.float64)
            L64 = torch.linalg.cholesky(Kb); Lv = torch.linalg.cholesky(Kb.float())
            av = torch.cholesky_solve(r[b].unsqueeze(-1), Lv).squeeze(-1).double().cpu().numpy()
            zv = torch.linalg.solve_triangular(Lv, r[b].unsqueeze(-1), upper=False).squeeze(-1).double()
            mll_v = -0.5 * (float(zv @ zv) + 2 * float(torch.log(torch.diagonal(Lv).double()).sum()) + n * np.log(2 * np.pi)) / n
            am = np.abs(ref["alpha"]).max()
            eh = float((f.L[b].double() - L64).abs().max()); ev = float((Lv.double() - L64).abs().max())
            print(f"{B:>3d}x{n:<4d} {b:>3d} {raw:>6.1f} {s2v:>9.2e} | {abs(o[b,0]-ref['mll'])/max(1,abs(ref['mll'])):>9.1e} {abs(o[b,1]-dsig)/abs(dsig):>9.1e} "
                  f"{abs(o[b,4]-ref['trinv'])/ref['trinv']:>9.1e} {np.abs(a[b]-ref['alpha']).max()/am:>9.1e} | {'':>12s} {abs(mll_v-ref['mll'])/max(1,abs(ref['mll'])):>9.1e} "
                  f"{np.abs(av-ref['alpha']).max()/am:>9.1e} {eh/ev:>14.2f}", flush=True)
