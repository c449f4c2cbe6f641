"""Randomised sweep of the MLL step through the small-batch / balanced schedules at sizes that reach the scheduled block
columns (N 2200..4100, odd sizes and ragged padding included), against the fp64 oracle on two series per case, plus
forward-only and potrf-with-scratch on the same inputs.  Exits non-zero on a miss."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
from oracle import volt_oracle as vo

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 14
worst = {"mll": 0.0, "dsig": 0.0, "alpha": 0.0, "fwd": 0.0, "potrf": 0.0}
for c in range(cases):
    n = int(rng.choice([2177, 2305, 2560, 2689, 2900, 3071, 3073, 3333, 3585, 3840, 4001, 4096]))
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 10, 12, 14, 17, 20, 24, 31]))
    x, F, vol = sde_batch(B, n, seed=int(rng.randint(1, 10000)))
    raw = rng.uniform(-5, 1, size=B)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    y = np.log(F[:, 1:])
    mean = y.mean(-1, keepdims=True) + 0 * y
    s2 = torch.tensor([vo.noise_from_raw(r) for r in raw], dtype=torch.float32).cuda()
    r = torch.tensor(y - mean).float().cuda()
    o, a, info = ops.mll_step(K, r, s2, want_grad=True)
    assert int(info.abs().sum()) == 0, (n, B, info)
    o, a = o.cpu().double().numpy(), a.cpu().double().numpy()
    of, _, inff = ops.mll_step(K, r, s2, want_grad=False)
    assert int(inff.abs().sum()) == 0
    fd = np.abs(of[:, 0].cpu().double().numpy() - o[:, 0]) / np.maximum(1.0, np.abs(o[:, 0]))   # (mll crosses 0: absolute below 1)
    worst["fwd"] = max(worst["fwd"], float(fd.max()))
    f = ops.potrf(K, s2)
    assert int(f.info.abs().sum()) == 0
    for b in sorted({0, B - 1}):
        ref = vo.mll_and_grads(K[b].cpu().double().numpy(), y[b], mean[b], float(raw[b]))
        worst["mll"] = max(worst["mll"], abs(o[b, 0] - ref["mll"]) / max(1.0, abs(ref["mll"])))
        dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
        worst["dsig"] = max(worst["dsig"], abs(o[b, 1] - dsig) / max(1e-30, abs(dsig)))
        worst["alpha"] = max(worst["alpha"], float(np.abs(a[b] - ref["alpha"]).max() / np.abs(ref["alpha"]).max()))
        Kb = K[b].double() + float(s2[b]) * torch.eye(n, device="cuda", dtype=torch.float64)
        Lr = torch.linalg.cholesky(Kb)
        e_lib = float((f.L[b].double() - Lr).abs().max() / Lr.abs().max())
        # yardstick: the vendor fp32 factorisation of the same matrix (small noise levels make K + s2 I ill-conditioned,
        # and any fp32 factor is then off by cond * eps)
        e_ven = float((torch.linalg.cholesky(Kb.float()).double() - Lr).abs().max() / Lr.abs().max())
        worst["potrf"] = max(worst["potrf"], e_lib / max(e_ven, 2e-6))
        a32 = torch.cholesky_solve(torch.tensor(y[b] - mean[b], device="cuda").float().unsqueeze(-1), torch.linalg.cholesky(Kb.float())).squeeze(-1).double().cpu().numpy()
        ea_ven = float(np.abs(a32 - ref["alpha"]).max() / np.abs(ref["alpha"]).max())
        worst["alpha_vs_vendor"] = max(worst.get("alpha_vs_vendor", 0.0), float(np.abs(a[b] - ref["alpha"]).max() / np.abs(ref["alpha"]).max()) / max(ea_ven, 1e-6))
    print(f"case {c}: N={n} B={B} ok   worst so far potrf x{worst['potrf']:.2f} alpha x{worst['alpha_vs_vendor']:.2f} (vendor alpha err {ea_ven:.1e}, raw {raw[b]:.2f})   fwd-vs-grad mll rel {fd.max():.1e} (series {int(fd.argmax())}, raw {raw[int(fd.argmax())]:.2f}, mll {o[int(fd.argmax()), 0]:.4f})", flush=True)
print("cases", cases, "worst", worst)
# potrf / alpha_vs_vendor: error relative to the vendor fp32 factorisation's error on the same matrix.  Reported, not
# gated: panels are solved by multiplication with the inverted diagonal block ("accuracy against conditioning"),
# which costs about an order of magnitude against substitution once K + s2 I is ill-conditioned (s2 < 0.05).
ok = worst["mll"] < 2e-5 and worst["dsig"] < 1e-3 and worst["fwd"] < 2e-6 and worst["potrf"] < 60 and worst["alpha_vs_vendor"] < 60
sys.exit(0 if ok else 1)
