"""Kernel-by-kernel diagnostics on the GPU box; writes gpurun_out/diag.log.  Not a test."""
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import volt_oracle as vo
from volt_amd import ops
from volt_amd.synthetic import sde_batch

os.makedirs("gpurun_out", exist_ok=True)
log = open("gpurun_out/diag.log", "w")


def P(*a):
    s = " ".join(str(x) for x in a)
    print(s)
    log.write(s + "\n")
    log.flush()


def section(name, fn):
    P(f"==== {name}")
    try:
        fn()
    except Exception:
        P(traceback.format_exc())


SIG2 = float(vo.noise_from_raw(1e-5))


def prob(B, n):
    x, F, vol = sde_batch(B, n)
    V = vo.cumtrapz(vol * vol, x)
    K = V[:, np.minimum.outer(np.arange(n), np.arange(n))].astype(np.float32)
    y = np.log(F[:, 1:])
    mean = np.stack([vo.ewma_mean(x, x, y[b], 25) for b in range(B)])
    return x, vol, K, y, mean


def t_env():
    P(torch.__version__, torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)


def t_fill():
    x, vol, K, _, _ = prob(2, 300)
    V = ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True)
    P("cumtrapz exact:", np.array_equal(V.cpu().numpy(), vo.cumtrapz(vol * vol, x)))
    Kg = ops.fill(V).cpu().numpy()
    P("fill exact:", np.array_equal(Kg, K))


def t_potrf():
    for B, n in [(1, 128), (2, 256), (2, 300), (8, 512)]:
        x, vol, K, _, _ = prob(B, n)
        f = ops.potrf(torch.tensor(K).cuda(), torch.full((B,), SIG2, device="cuda"))
        torch.cuda.synchronize()
        L = f.L.cpu().numpy().astype(np.float64)
        for b in range(min(B, 2)):
            L64 = np.linalg.cholesky(K[b].astype(np.float64) + SIG2 * np.eye(n))
            err = np.abs(L[b] - L64)
            i, j = np.unravel_index(err.argmax(), err.shape)
            P(f"potrf B={B} n={n} b={b} info={int(f.info[b])} maxerr={err.max():.3e} at ({i},{j}) scale={np.abs(L64).max():.3f}"
              f" blockerr={[float(f'{err[p*128:(p+1)*128, q*128:(q+1)*128].max():.1e}') for p in range((n+127)//128) for q in range(p+1)][:10]}")


def t_solve():
    B, n = 2, 300
    x, vol, K, y, mean = prob(B, n)
    r = (y - mean).astype(np.float32)
    f = ops.potrf(torch.tensor(K).cuda(), torch.full((B,), SIG2, device="cuda"))
    z = ops.trsv(f, torch.tensor(r).cuda()).cpu().numpy()
    a = ops.cholesky_solve(f, torch.tensor(r).cuda()).cpu().numpy()
    Y = ops.trtri(f).cpu().numpy()
    for b in range(B):
        Ks = K[b].astype(np.float64) + SIG2 * np.eye(n)
        L64 = np.linalg.cholesky(Ks)
        z64 = np.linalg.solve(L64, r[b])
        a64 = np.linalg.solve(Ks, r[b])
        Y64 = np.linalg.inv(L64).T
        P(f"b={b} trsv err {np.abs(z[b]-z64).max()/np.abs(z64).max():.2e}  solve err {np.abs(a[b]-a64).max()/np.abs(a64).max():.2e}"
          f"  trtri err {np.abs(Y[b]-Y64).max()/np.abs(Y64).max():.2e}")


def t_mll():
    for B, n in [(1, 256), (2, 399), (4, 512)]:
        x, vol, K, y, mean = prob(B, n)
        r = (y - mean).astype(np.float32)
        o = vo.mll_and_grads(K, y.astype(np.float32), mean.astype(np.float32), 1e-5)
        for wg in (True, False):
            out, alpha, info = ops.mll_step(torch.tensor(K).cuda(), torch.tensor(r).cuda(),
                                            torch.full((B,), SIG2, device="cuda"), want_grad=wg)
            out = out.cpu().numpy()
            P(f"mll B={B} n={n} grad={wg} info={info.cpu().numpy()}")
            P("   gpu  mll", out[:, 0], "quad", out[:, 2], "logdet", out[:, 3], "tr", out[:, 4], "aa", out[:, 5], "dsig", out[:, 1])
            P("   orc  mll", o["mll"], "quad", o["quad"], "logdet", o["logdet"], "tr", o["trinv"], "aa", o["aa"],
              "dsig", 0.5 * (o["aa"] - o["trinv"]) / n)


def t_speed():
    for B, n in [(8, 2048), (16, 4096)]:
        x, F, vol = sde_batch(B, n)
        Kd = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
        r = torch.randn(B, n, device="cuda") * 0.01
        s2 = torch.full((B,), SIG2, device="cuda")
        ws = ops.MllWorkspace(B, n, True, "cuda")
        for wg in (True, False):
            ops.mll_step(Kd, r, s2, ws if wg else None, want_grad=wg)
            torch.cuda.synchronize()
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                ops.mll_step(Kd, r, s2, ws if wg else None, want_grad=wg)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / reps
            flops = B * (2 if wg else 1) * n ** 3 / 3
            P(f"speed B={B} n={n} grad={wg}: {dt*1e3:.2f} ms/step  {flops/dt/1e12:.1f} TF/s")


for name, fn in [("env", t_env), ("fill", t_fill), ("potrf", t_potrf), ("solve", t_solve), ("mll", t_mll),
                 ("speed", t_speed)]:
    section(name, fn)
