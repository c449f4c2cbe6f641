"""The fp64 one-launch step (csrc/batch64_step.hip) against LAPACK-in-torch (fp64 cholesky / cholesky_inverse on the GPU) and
against chol64.hip's launch-per-column schedules: relative errors of the factor, of the step's scalars and of alpha; run-to-run
repeatability (the one launch has no atomics: every repeat must equal the first bit for bit); ms per call.
    VOLT_TUNE=1 VOLT_BATCH64=0 python scripts/batch64_check.py 1x4096 8x4096      # the old schedule
    python scripts/batch64_check.py 1x4096 8x4096                                 # what the library does by default
--notime skips timing, --reps-check N repeats N times."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch

args = sys.argv[1:]
shapes = [a for a in args if "x" in a and a[0].isdigit()] or ["1x512", "2x1000", "1x4096", "3x2048", "8x1024", "8x4096"]
notime = "--notime" in args
noref = "--noref" in args            # timing only (the gate sweep): skip the LAPACK reference
nrep = int(args[args.index("--reps-check") + 1]) if "--reps-check" in args else 0
L = _lib.lib()


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
for sh in shapes:
    B, n = map(int, sh.split("x"))
    x, F, vol = sde_batch(min(B, 4), n)
    vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 4) + 1, 1))[:B]
    vol = vol * (1.0 + 0.01 * np.arange(B)[:, None])
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda().double())
    r = (y - y.mean(-1, keepdim=True)).contiguous()
    s2 = torch.full((B,), 0.05, device="cuda", dtype=torch.float64)
    Np = ops.padded_n(n)
    one = bool(L.volt_potrf_workspace_bytes_f64(B, Np))
    f = ops.potrf(K, s2)
    torch.cuda.synchronize()
    ws = ops.MllWorkspace(B, n, True, K.device, torch.float64)
    o, a, info = ops.mll_step(K, r, s2, ws)
    torch.cuda.synchronize()
    row = {"shape": sh, "one_launch": one, "info": int(f.info.abs().sum().item()) + int(info.abs().sum().item())}
    if not noref:
        Ks = K + torch.diag_embed(s2[:, None].expand(B, n))
        Lref = torch.linalg.cholesky(Ks)
        eL = float(((f.L - Lref).abs().amax() / Lref.abs().amax()).item())
        # (torch.cholesky_solve on a batch fails with a launch failure on this image: two triangular solves instead)
        st = torch.linalg.solve_triangular
        al_ref = st(Lref.mT, st(Lref, r[..., None], upper=False), upper=True)[..., 0]
        Linv = st(Lref, torch.eye(n, device=K.device, dtype=torch.float64).expand(B, n, n), upper=False)
        q = (r * al_ref).sum(-1); ld = 2 * torch.log(torch.diagonal(Lref, dim1=-2, dim2=-1)).sum(-1)
        mll = -0.5 * (q + ld + n * np.log(2 * np.pi)) / n
        dm = 0.5 * ((al_ref ** 2).sum(-1) - (Linv ** 2).sum((-2, -1))) / n
        del Linv, Ks
        row.update(err_L=eL, err_mll=float(((o[:, 0] - mll).abs() / mll.abs()).max().item()),
                   err_dmll=float(((o[:, 1] - dm).abs() / dm.abs().clamp_min(1e-30)).max().item()),
                   err_alpha=float(((a - al_ref).abs().amax() / al_ref.abs().amax()).item()))
    if nrep:
        A0, o0, a0 = f.A.clone(), o.clone(), a.clone()
        same = True
        for _ in range(nrep):
            f2 = ops.potrf(K, s2)
            o2, a2, _i = ops.mll_step(K, r, s2, ws)
            same = same and torch.equal(torch.tril(f2.A), torch.tril(A0)) and torch.equal(o2, o0) and torch.equal(a2, a0)
        row["repeatable"] = bool(same)
    if not notime:
        Aprep = f.A.clone()
        _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, Aprep.data_ptr(), B, n, _lib.stream_ptr()), "prep")
        t_copy = timeit(lambda: f.A.copy_(Aprep))

        def potrf_only():
            f.A.copy_(Aprep)
            ops.potrf_f64_inplace(f.A, f.Winv, f.info)
        row["potrf_ms"] = round(timeit(potrf_only) - t_copy, 3)
        row["step_ms"] = round(timeit(lambda: ops.mll_step(K, r, s2, ws)), 3)
        fl = B * Np ** 3 / 3
        row["potrf_TF"] = round(fl / row["potrf_ms"] / 1e9, 2)
        row["step_TF"] = round(2 * fl / row["step_ms"] / 1e9, 2)
    rows.append(row)
    print(json.dumps(row), flush=True)
