"""SingleTaskVariationalGP, voltron/models/single_task_variational_gp.py:69-236 -- SURVEY 8(f) row 4.

The variational GP LearnGPCV fits to the scaled returns (train_utils.py:26-31): inducing points fixed at the
training inputs, unwhitened strategy, Cholesky variational distribution.  Only that configuration is on the
accelerated path; the whitened strategy / learned inducing locations / botorch posterior interface of the
reference class are not (they are never reached from LearnGPCV).  The dense algebra -- here the start-up
factorisations of ``initialize_variational_parameters`` -- runs through the HIP library (potrf, trtri, gemm)."""
import torch

from .. import ops
from ..gp import ConstantMean, EqualMemo, Module, MultivariateNormal, _dense, _safe_factor
from ..likelihoods import VolatilityGaussianLikelihood  # noqa: F401  (re-exported like the reference's module namespace)
from ..variational import CholeskyVariationalDistribution, UnwhitenedVariationalStrategy, VariationalLatent


class SingleTaskVariationalGP(Module):
    def __init__(self, init_points=None, likelihood=None, learn_inducing_locations=True, covar_module=None,
                 mean_module=None, use_piv_chol_init=True, num_inducing=None, use_whitened_var_strat=True,
                 init_targets=None, train_inputs=None, train_targets=None, outcome_transform=None,
                 input_transform=None):
        super().__init__()
        if use_whitened_var_strat:
            raise NotImplementedError("use_whitened_var_strat=True is outside the accelerated path: LearnGPCV uses the "
                                      "unwhitened strategy (train_utils.py:30)")
        if outcome_transform is not None or input_transform is not None:
            raise NotImplementedError("botorch input/outcome transforms are not used by LearnGPCV")
        if covar_module is None:
            raise NotImplementedError("pass covar_module (BMKernel / FBMKernel, train_utils.py:22-25)")
        inducing_points = init_points.detach().clone()
        variational_distribution = CholeskyVariationalDistribution(inducing_points.shape[-2])
        self.variational_strategy = UnwhitenedVariationalStrategy(
            self, inducing_points, variational_distribution, learn_inducing_locations=learn_inducing_locations)
        self.mean_module = ConstantMean() if mean_module is None else mean_module
        self.covar_module = covar_module
        self.likelihood = likelihood
        self.train_inputs = [train_inputs] if train_inputs is not None else [init_points]
        self.train_targets = train_targets if train_targets is not None else init_targets
        self.condition_into_exact = True
        object.__setattr__(self, "_eq_memo", EqualMemo())
        self.to(init_points.device)

    @property
    def num_outputs(self):
        return 1

    def forward(self, x):
        """The prior at x (single_task_variational_gp.py:117-121)."""
        return MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def __call__(self, x):
        arg = x
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        Z = self.variational_strategy.inducing_points
        # LearnGPCV passes the same train_x object on every iteration (train_utils.py:51): compare once, not 1000 times
        if x.shape == Z.shape and self._eq_memo.equal(arg, Z, lambda: torch.equal(x, Z)):
            return VariationalLatent(self)
        raise NotImplementedError("SingleTaskVariationalGP(x) away from the inducing points is outside the accelerated "
                                  "path: LearnGPCV only evaluates the model at train_x (train_utils.py:51,60)")

    def initialize_variational_parameters(self, likelihood, x, f=None, y=None):
        """single_task_variational_gp.py:190-236 for the "exp" likelihood: mean = log running std of y, covariance
        S = L (L'HL + I)^-1 L' with H the (clamped) inverse Hessian, stored as 10 * chol(S).
        y [N], or [T,N] for T series sharing the inducing points (batched parameters)."""
        if getattr(likelihood, "param", "exp") != "exp":
            raise NotImplementedError('the "cv" initialisation (single_task_variational_gp.py:214-224) is not implemented')
        with torch.no_grad():
            Z = self.variational_strategy.inducing_points
            kuu = _dense(self.covar_module(Z)).to(torch.float32)
            y2 = y.reshape(-1, y.shape[-1]).to(torch.float32)
            T, N = y2.shape
            # running std of y[:i] (unbiased), one prefix-sum pass instead of the reference's N slices
            idx = torch.arange(N, device=y2.device, dtype=torch.float64)
            c1 = torch.cumsum(y2.double(), -1)
            c2 = torch.cumsum(y2.double() ** 2, -1)
            s1 = torch.cat([torch.zeros(T, 1, dtype=torch.float64, device=y2.device), c1[:, :-1]], -1)
            s2 = torch.cat([torch.zeros(T, 1, dtype=torch.float64, device=y2.device), c2[:, :-1]], -1)
            var = (s2 - s1 * s1 / idx.clamp_min(1)) / (idx - 1).clamp_min(1)
            running_std = var.clamp_min(0).sqrt().to(torch.float32)
            running_std[:, :10] = running_std[:, 10:11]
            if f is None:
                f = running_std.clamp(min=1e-4).log()
            f = f.reshape(T, N)
            # torch.diag_embed(...).clamp(min=1e-4, max=1000.): the clamp also lifts the off-diagonal zeros to 1e-4
            h = (0.5 * y2.pow(-2.0) * (f * 2.0).exp()).clamp(min=1e-4, max=1000.0)
            ih = torch.full((T, N, N), 1e-4, device=y2.device)
            ih.diagonal(dim1=-2, dim2=-1).copy_(h)
            L = _safe_factor(kuu.reshape(-1, N, N))[0].L.expand(T, N, N)                 # kuu.cholesky()
            Lt = L.mT.contiguous()
            HL = ops.gemm_nt(ih, Lt, uplo_b=2)                                           # H L
            inner = ops.gemm_nt(Lt, HL.mT.contiguous(), uplo_a=2)                        # L' H L
            inner = inner + torch.eye(N, device=y2.device)
            fi = _safe_factor(inner)[0]
            Yi = ops.trtri(fi)                                                           # chol(inner)^-T
            R = ops.gemm_nt(L.contiguous(), Yi.mT.contiguous(), uplo_a=1, uplo_b=1)        # L chol(inner)^-1 ... R R' = S
            S = ops.gemm_nt(R, R)
            S_root = _safe_factor(S)[0].L * 10.0
            dist = self.variational_strategy._variational_distribution
            squeeze = y.ndim == 1
            dist.variational_mean.data = f[0] if squeeze else f
            dist.chol_variational_covar.data = S_root[0] if squeeze else S_root
            self.variational_strategy.variational_params_initialized.fill_(1)
            const = running_std.mean(-1).log()
            self.mean_module.constant.data = const[0] if squeeze else const.unsqueeze(-1)
