"""Brownian-motion GP over the log-volatility path -- voltron/models/BMGP.py:9-28.

SURVEY 8(f) row 1: it supplies ``pred_vol`` at voltron/rollout_utils.py:66 through
``model.vol_model(test_x).sample(...)``.  Training-mode call -> prior MVN (for an MLL); eval-mode
call -> exact-GP posterior at the test points, computed with the same HIP Cholesky / triangular
inverse as the data model (K_s^-1 = Y Y^T, Y = L^-T), the products on the library's own MFMA GEMM (ops.gemm_nt).
The botorch-based MultitaskBMGP (:30-56) is out of scope.

Addition: ``train_y`` [T,N] builds T independent vol models over shared inputs in one object (batched kernel
parameter, noise and posterior) -- what the batched forecast driver trains instead of looping over tickers.
"""
import torch

from .. import ops
from ..gp import ExactGP, MultivariateNormal, _safe_factor
from ..kernels.BMKernel import BMKernel
from ..kernels.FBMKernel import FBMKernel


class BMGP(ExactGP):
    def __init__(self, train_x, train_y, likelihood, kernel="bm"):
        super().__init__(train_x, train_y, likelihood)
        kw = {"batch_shape": train_y.shape[:-1]} if train_y.ndim > 1 else {}
        if kernel == "bm":
            self.covar_module = BMKernel(**kw).to(train_x.device)
        elif kernel == "fbm":
            self.covar_module = FBMKernel(**kw).to(train_x.device)      # BMGP.py:15-16; dense d mll / d K path
        self.scaling = (train_x[1] - train_x[0])

    def mean_module(self, x):
        return -0.5 * self.covar_module.vol.pow(2.0) * x.squeeze()        # BMGP.py:20-21

    def forward(self, x):
        return MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def posterior_call(self, x):
        """Exact-GP posterior at the test points: K_s = L L' on the HIP potrf, K_s^-1 = Y Y' with Y = L^-T from the HIP
        triangular inverse, every product on the library GEMM.  One series -> MultivariateNormal(mean [H], cov [H,H]);
        T series over the shared grid -> (mean [T,H], cov [T,H,H])."""
        with torch.no_grad():
            from ..gp import _dense
            xt = self.train_inputs[0]
            y = self.train_targets
            batched = y.ndim > 1
            T = y.shape[0] if batched else 1
            n, H = xt.shape[0], x.shape[0]
            Ktt = _dense(self.covar_module(xt, xt)).reshape(T, n, n)
            noise = self.likelihood.noise.reshape(-1).expand(T)
            A = Ktt + noise.reshape(T, 1, 1) * torch.eye(n, device=xt.device)
            f, _ = _safe_factor(A)
            Linv = ops.trtri(f).mT.contiguous()                                            # L^-1, rows K-contiguous
            Kst = _dense(self.covar_module(x, xt)).reshape(T, H, n)
            G = ops.gemm_nt(Kst, Linv, uplo_b=1)                                           # K_*t L^-T
            r = (y - self.mean_module(xt)).to(torch.float32).reshape(T, 1, n)
            w = ops.gemm_nt(r, Linv, uplo_b=1)                                             # (L^-1 r)'
            mean = self.mean_module(x).reshape(T, H) + ops.gemm_nt(G, w).reshape(T, H)
            cov = _dense(self.covar_module(x, x)).reshape(T, H, H) - ops.gemm_nt(G, G)
            if not batched:
                mean, cov = mean[0], cov[0]
            return MultivariateNormal(mean, cov)
