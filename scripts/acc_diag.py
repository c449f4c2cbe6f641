"""Where does the fp32 step lose accuracy against the vendor's substitution-based fp32 path once K + s2 I is
ill-conditioned?  Error maps per 128-block of the factor, quality of the inverted diagonal blocks, and alpha through
several routes.  Diagnostic (round 4); prints only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import volt_oracle as vo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2019
B = 2
x, F, vol = sde_batch(B, N, seed=seed)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
y = np.log(F[:, 1:]); mean = y.mean(-1, keepdims=True) + 0 * y
r = torch.tensor(y - mean).float().cuda()
nb = (N + 127) // 128
for raw in (-3.0, -6.0, -9.0, -11.8):
    s2v = vo.noise_from_raw(raw)
    s2 = torch.full((B,), s2v, dtype=torch.float32).cuda()
    f = ops.potrf(K, s2)
    assert int(f.info.abs().sum()) == 0
    o, a, info = ops.mll_step(K, r, s2, want_grad=True)
    a = a.clone()
    for b in (0,):
        Kb = K[b].double() + float(s2[b]) * torch.eye(N, device="cuda", dtype=torch.float64)
        L64 = torch.linalg.cholesky(Kb)
        Lv = torch.linalg.cholesky(Kb.float())
        Lh = f.L[b]
        sc = float(L64.abs().max())
        def bmap(L):
            E = (L.double() - L64).abs()
            Ep = torch.zeros(nb * 128, nb * 128, device="cuda", dtype=torch.float64); Ep[:N, :N] = E
            return Ep.reshape(nb, 128, nb, 128).amax(dim=(1, 3)) / sc
        Mh, Mv = bmap(Lh), bmap(Lv)
        dh, dv = torch.diagonal(Mh), torch.diagonal(Mv)
        offh = (Mh - torch.diag(dh)); offv = (Mv - torch.diag(dv))
        print(f"raw {raw} s2 {s2v:.2e} N {N}: factor err hip {float(Mh.max()):.2e} vendor {float(Mv.max()):.2e} ratio {float(Mh.max()/Mv.max()):.1f} | "
              f"diag blocks hip {float(dh.max()):.2e} ven {float(dv.max()):.2e} | panels hip {float(offh.max()):.2e} ven {float(offv.max()):.2e}")
        cols_h = Mh.amax(dim=0).cpu().numpy(); cols_v = Mv.amax(dim=0).cpu().numpy()
        print("   per block column hip/vendor:", " ".join(f"{h/v:.1f}" for h, v in zip(cols_h, cols_v)))
        # inverted diagonal blocks
        worst = [0, 0, 0, 0]
        for k in range(N // 128):
            Lkk = torch.tril(f.A[b, k*128:(k+1)*128, k*128:(k+1)*128]).double()
            W = f.Winv[b, k].double()
            I = torch.eye(128, device="cuda", dtype=torch.float64)
            Wt = torch.linalg.inv(Lkk.float()).double()      # vendor fp32 inverse of the same block
            Wx = torch.linalg.inv(Lkk)
            worst[0] = max(worst[0], float((W @ Lkk - I).abs().max())); worst[1] = max(worst[1], float((Lkk @ W - I).abs().max()))
            worst[2] = max(worst[2], float((W - Wx).abs().max() / Wx.abs().max())); worst[3] = max(worst[3], float((Wt - Wx).abs().max() / Wx.abs().max()))
        print(f"   W_k: |WL-I| {worst[0]:.1e} |LW-I| {worst[1]:.1e} rel err hip {worst[2]:.1e} torch-fp32-inv {worst[3]:.1e}")
        # alpha routes
        a64 = torch.cholesky_solve(r[b].double().unsqueeze(-1), L64).squeeze(-1)
        am = float(a64.abs().max())
        e = lambda v: float((v.double() - a64).abs().max()) / am
        a_v = torch.cholesky_solve(r[b].unsqueeze(-1), Lv).squeeze(-1)
        a_hs = ops.cholesky_solve(f, r)[b]                                  # hip factor, hip substitution
        a_hv = torch.cholesky_solve(r[b].unsqueeze(-1), Lh).squeeze(-1)     # hip factor, vendor substitution
        Yv = torch.linalg.inv(Lv)                                            # vendor factor, explicit inverse
        a_vi = Yv.T @ (Yv @ r[b])
        Yh = ops.trtri(f)[b]                                                 # hip Y = L^-T
        a_hy = Yh @ (Yh.T @ r[b])
        Yhx = torch.linalg.inv(Lh.double())
        eY = float((Yh.double().T - Yhx).abs().max() / Yhx.abs().max())
        eYv = float((torch.linalg.inv(Lh).double() - Yhx).abs().max() / Yhx.abs().max())
        # refinement with an fp64-accumulated residual
        r1 = (r[b].double() - Kb @ a[b].double()).float()
        a_ref = a[b] + Yh @ (Yh.T @ r1)
        print(f"   alpha err/max: step {e(a[b]):.2e} | vendor {e(a_v):.2e} | hipL+hip trsv {e(a_hs):.2e} | hipL+vendor trsv {e(a_hv):.2e} | "
              f"vendorL explicit inv {e(a_vi):.2e} | hip Y(Y'r) torch matvec {e(a_hy):.2e} | refined {e(a_ref):.2e}")
        print(f"   Y = inv(L_hip): hip trtri rel err {eY:.1e}, torch fp32 inv {eYv:.1e}")
