"""Randomised sweep of the fp32 MLL step through the small-batch / balanced / grouped schedules, against the fp64 oracle
and -- as the yardstick for what ANY fp32 factorisation can deliver on the same matrix -- against the vendor's fp32
potrf + cholesky_solve.  Shared by tests/test_gpu_lownoise.py (a few cases under pytest) and scripts/fuzz_sched.py
(long runs).  Test infrastructure: imports the oracle."""
import numpy as np
import torch

from oracle import volt_oracle as vo
from volt_amd import ops
from volt_amd.synthetic import sde_batch

SIZES = [2177, 2305, 2560, 2689, 2900, 3071, 3073, 3333, 3585, 3840, 4001, 4096]
BATCHES = [1, 2, 3, 4, 5, 7, 8, 9, 10, 12, 14, 17, 20, 24, 31]

# Gates.  Absolute (fp32 step vs fp64 oracle, raw_noise in [-5, 1] i.e. sigma^2 down to 7e-3, cond up to ~1e6):
GATE = {"mll": 5e-6, "dsig": 1e-4, "fwd": 2e-6, "alpha": 1e-4,
        # relative to the vendor fp32 error on the same matrix (floored at 2e-6 / 1e-6: below that the vendor's own
        # error is a lucky draw, not a yardstick).  Two fp32 factorisations of equal quality leave INDEPENDENT errors,
        # so the per-case ratio scatters around 1 with a tail; the typical case is gated at 1.5x, the worst at 5x
        # (rounds 1-3: 8.8x / 26x worst, ~5x / ~10x typical -- the one-chain K sum).
        "potrf_median": 1.5, "alpha_median": 1.5, "potrf_max": 5.0, "alpha_max": 5.0}


def run(seed: int, cases: int, verbose: bool = False):
    rng = np.random.RandomState(seed)
    worst = {"mll": 0.0, "dsig": 0.0, "alpha": 0.0, "fwd": 0.0}
    potrf_r, alpha_r = [], []
    for c in range(cases):
        n = int(rng.choice(SIZES))
        B = int(rng.choice(BATCHES))
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
        fd = np.abs(of[:, 0].cpu().double().numpy() - o[:, 0]) / np.maximum(1.0, np.abs(o[:, 0]))   # (mll crosses 0)
        worst["fwd"] = max(worst["fwd"], float(fd.max()))
        f = ops.potrf(K, s2)
        assert int(f.info.abs().sum()) == 0
        for b in sorted({0, B - 1}):
            ref = vo.mll_and_grads(K[b].cpu().double().numpy(), y[b], mean[b], float(raw[b]))
            worst["mll"] = max(worst["mll"], abs(o[b, 0] - ref["mll"]) / max(1.0, abs(ref["mll"])))
            dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
            worst["dsig"] = max(worst["dsig"], abs(o[b, 1] - dsig) / max(1e-30, abs(dsig)))
            amax = np.abs(ref["alpha"]).max()
            ea = float(np.abs(a[b] - ref["alpha"]).max() / amax)
            worst["alpha"] = max(worst["alpha"], ea)
            Kb = K[b].double() + float(s2[b]) * torch.eye(n, device="cuda", dtype=torch.float64)
            Lr = torch.linalg.cholesky(Kb)
            Lv = torch.linalg.cholesky(Kb.float())
            e_lib = float((f.L[b].double() - Lr).abs().max() / Lr.abs().max())
            e_ven = float((Lv.double() - Lr).abs().max() / Lr.abs().max())
            potrf_r.append(e_lib / max(e_ven, 2e-6))
            a32 = torch.cholesky_solve(r[b].unsqueeze(-1), Lv).squeeze(-1).double().cpu().numpy()
            ea_ven = float(np.abs(a32 - ref["alpha"]).max() / amax)
            alpha_r.append(ea / max(ea_ven, 1e-6))
            if verbose:
                print(f"case {c}: N={n} B={B} series {b} raw {raw[b]:.2f}: factor x{potrf_r[-1]:.2f} of vendor ({e_ven:.1e}), "
                      f"alpha x{alpha_r[-1]:.2f} of vendor ({ea_ven:.1e})", flush=True)
    worst.update(potrf_median=float(np.median(potrf_r)), alpha_median=float(np.median(alpha_r)),
                 potrf_max=float(np.max(potrf_r)), alpha_max=float(np.max(alpha_r)))
    return worst


def failures(worst):
    return {k: (worst[k], g) for k, g in GATE.items() if not worst[k] < g}
