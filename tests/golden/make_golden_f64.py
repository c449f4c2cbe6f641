#!/usr/bin/env python3
"""fp64 factor / solve fixture from the reference's own kernel and the ATen calls it makes.

Loads voltron/kernels/VolKernel.py from /root/reference exactly like make_golden.py (same stand-ins), builds the
NOISE-FREE train block of a 200-point series in fp64 with the reference's VolatilityKernel.forward (dtype is
inherited, VolKernel.py:28-33) and runs the two calls of rollout_utils.py:35-36 on it:
    K_tr_chol = psd_safe_cholesky(K_tr, jitter=1e-4)            (stand-in: torch.linalg.cholesky_ex, no jitter needed)
    sol       = torch.cholesky_solve(train_diffs, K_tr_chol)
Writes tests/golden/chol64.npz: inputs (x, vol, rhs) and outputs (lower triangle of the factor, packed row-wise,
and the solve).  Build container only; the GPU box sees the .npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_f64.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg


def main():
    mg._install_standins()
    VK = mg._load("ref_volkernel", "kernels/VolKernel.py")
    from gpytorch.utils.cholesky import psd_safe_cholesky          # the stand-in installed above
    n = 200
    F, vol = mg.sde_series(n, 77)
    x = (torch.arange(n) / 252.).double()
    vol = torch.tensor(vol).double()
    K = VK.VolatilityKernel().forward(x.unsqueeze(-1), vol.unsqueeze(-1))
    assert K.dtype == torch.float64
    g = torch.Generator().manual_seed(77)
    rhs = torch.randn(n, 1, generator=g, dtype=torch.float64) * 0.02
    L = psd_safe_cholesky(K, jitter=1e-4)                            # rollout_utils.py:35
    sol = torch.cholesky_solve(rhs, L)                               # rollout_utils.py:36
    il = np.tril_indices(n)
    np.savez_compressed(os.path.join(mg.OUT, "chol64.npz"), x=x.numpy(), vol=vol.numpy(), rhs=rhs[:, 0].numpy(),
                        L_packed=L.numpy()[il], sol=sol[:, 0].numpy(),
                        cond=np.array(np.linalg.cond(K.numpy())))
    print("cond", float(np.linalg.cond(K.numpy())), "bytes", os.path.getsize(os.path.join(mg.OUT, "chol64.npz")))


if __name__ == "__main__":
    main()
