"""GPCV stage on the MI355X vs the CPU oracle (LearnGPCV, voltron/train_utils.py:15-67).

Tolerances (fp32 HIP path vs the fp64 oracle): ELBO value rel 2e-5; gradients rel 2e-3 of the gradient's
max-norm (the likelihood term sums 75 nodes of magnitude up to y^2/s^2 in fp32)."""
import math

import numpy as np
import pytest
import torch

from oracle import gpcv_oracle as GO
from volt_amd.synthetic import sde_series

pytestmark = pytest.mark.gpu


def _problem(n, seed, dtype=torch.float64):
    F, _ = sde_series(n, seed)
    x = torch.arange(n, dtype=dtype) / 252
    yy = GO.scaled_returns(x, torch.tensor(F, dtype=dtype))
    f, S_root, c0 = GO.init_variational(x, yy)
    g = torch.Generator().manual_seed(seed)
    m = f + 0.1 * torch.randn(n, generator=g, dtype=dtype)
    Lq = (S_root / 10 + 0.01 * torch.randn(n, n, generator=g, dtype=dtype)).tril()
    Lq = Lq + torch.triu(torch.randn(n, n, generator=g, dtype=dtype), 1)       # junk above the diagonal: must be ignored
    return x, yy, m, Lq, c0.reshape(1)


@pytest.mark.parametrize("n,B,kernel", [(200, 1, "bm"), (399, 3, "bm"), (512, 2, "bm"), (300, 2, "fbm"),
                                        (33, 1, "bm"), (129, 2, "bm"), (257, 1, "fbm"), (640, 9, "bm")])
def test_gpcv_step_matches_oracle(n, B, kernel):
    from volt_amd import ops
    dev = "cuda:0"
    probs = [_problem(n, 2019 + b) for b in range(B)]
    raw_vol = torch.logit(torch.tensor([0.2], dtype=torch.float64))
    gh_x, gh_w = GO.gauss_hermite(75)
    Ks, vals, grads, terms = [], [], [], []
    for (x, yy, m, Lq, c) in probs:
        ps = [t.clone().requires_grad_(True) for t in (m, Lq, c)]
        vol = torch.sigmoid(raw_vol)
        K = (GO.bm_cov(x, vol) if kernel == "bm" else GO.fbm_cov(x, vol)).detach().requires_grad_(True)
        t = GO.elbo_terms(ps[0], ps[1], ps[2], K, yy, gh_x, gh_w)
        g = torch.autograd.grad(t["elbo"], ps + [K])
        Ks.append(K.detach())
        terms.append({k: float(v.detach()) for k, v in t.items()})
        grads.append(g)
    f32 = lambda ts: torch.stack([t.to(torch.float32) for t in ts]).to(dev)
    K = f32(Ks)
    m = f32([p[2] for p in probs])
    Lq = f32([p[3] for p in probs])
    y = f32([p[1] for p in probs])
    mu = f32([p[4].expand(n) for p in probs])
    ws = ops.gpcv_step(K, m - mu, m, Lq, y, gh_x.to(dev), (gh_w / math.sqrt(math.pi)).to(dev), want_dk=True,
                       w_ell=1.0 / n, w_kl=1.0 / n)
    torch.cuda.synchronize()
    assert int(ws.info.abs().sum()) == 0
    out = ws.out.double().cpu()
    for b in range(B):
        t = terms[b]
        for col, key in ((0, "ell"), (1, "kl"), (2, "quad"), (3, "logdet_k"), (4, "logdet_s"), (5, "trace")):
            assert abs(out[b, col] - t[key]) <= 5e-5 * max(1.0, abs(t[key])), (key, out[b, col], t[key])
        assert abs(out[b, 9] - t["elbo"]) <= 5e-5 * max(1.0, abs(t["elbo"]))
        gm, gL, gc, gK = grads[b]
        rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
        assert rel(ws.grad_m[b], gm) < 2e-3
        assert rel(ws.grad_Lq[b], gL) < 2e-3
        assert abs(float(ws.grad_mu[b].sum()) - float(gc)) < 2e-3 * max(1.0, abs(float(gc)))
        assert rel(ws.grad_K[b], gK) < 5e-3


def test_gemm_nt_matches_torch():
    from volt_amd import ops
    g = torch.Generator().manual_seed(0)
    A = torch.randn(2, 300, 200, generator=g).cuda()
    B = torch.randn(2, 130, 200, generator=g).cuda()
    C = ops.gemm_nt(A, B)
    ref = (A.double() @ B.double().mT)
    assert float((C.double() - ref).abs().max()) < 2e-4
    L = torch.randn(1, 384, 384, generator=g).tril().cuda()
    U = torch.randn(1, 384, 384, generator=g).triu().cuda()
    C2 = ops.gemm_nt(U, L, uplo_a=2, uplo_b=1)
    assert float((C2.double() - U.double() @ L.double().mT).abs().max()) < 2e-4


def _prices(n, seed):
    F, V = sde_series(n, seed)
    return torch.tensor(F), V


def test_initialize_variational_parameters_matches_oracle():
    from volt_amd.kernels import BMKernel
    from volt_amd.likelihoods import VolatilityGaussianLikelihood
    from volt_amd.models import SingleTaskVariationalGP
    from volt_amd import gp
    n = 300
    F, _ = _prices(n, 2019)
    x = torch.arange(n, dtype=torch.float32) / 252
    yy = GO.scaled_returns(x, F)
    f, S_root, c0 = GO.init_variational(x.double(), yy.double())
    lh = VolatilityGaussianLikelihood(param="exp")
    model = SingleTaskVariationalGP(init_points=x.cuda().view(-1, 1), likelihood=lh, use_piv_chol_init=False,
                                    mean_module=gp.ConstantMean(), covar_module=BMKernel().cuda(),
                                    learn_inducing_locations=False, use_whitened_var_strat=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.initialize_variational_parameters(lh, x.cuda(), y=yy.cuda())
    d = model.variational_strategy._variational_distribution
    assert float((d.variational_mean.detach().cpu().double() - f).abs().max()) < 1e-5
    assert abs(float(model.mean_module.constant.detach()) - float(c0)) < 1e-5
    S_hip = d.chol_variational_covar.cpu().double()
    S_ref = S_root
    # compare the covariance the factor stands for (the factor itself is conditioned like S, ~1e-6 .. 1)
    cov_h, cov_r = S_hip @ S_hip.mT, S_ref @ S_ref.mT
    assert float((cov_h - cov_r).abs().max() / cov_r.abs().max()) < 5e-4


@pytest.mark.parametrize("tag", ["n60_f64", "n120_f64", "n90_f32", "n80_wind_f64"])
def test_start_up_matches_the_reference_code(golden, tag):
    """The HIP-backed initialize_variational_parameters, BMKernel and the "exp" likelihood against tests/golden/gpcv.npz --
    outputs of the reference's OWN code for them (single_task_variational_gp.py:204-254, BMKernel.py:38-52,
    volatility_likelihood.py:42-50, executed by tests/golden/make_golden_gpcv.py).  The product computes in fp32: mean and
    constant 1e-5, the covariance the factor stands for 2e-3 (its condition number is > 1e6)."""
    from volt_amd.kernels import BMKernel
    from volt_amd.likelihoods import VolatilityGaussianLikelihood
    from volt_amd.models import SingleTaskVariationalGP
    from volt_amd import gp
    g = golden("gpcv")
    x = torch.tensor(g[f"{tag}_x"]).float().cuda()
    yy = torch.tensor(g[f"{tag}_y"]).float().cuda()
    lh = VolatilityGaussianLikelihood(param="exp")
    kern = BMKernel().cuda()
    assert abs(float(kern.vol) - float(g[f"{tag}_vol"])) < 1e-6
    from volt_amd.gp import _dense
    kuu = _dense(kern(x.view(-1, 1))).cpu().double()
    kg = torch.tensor(g[f"{tag}_kuu"]).double()
    assert float((kuu - kg).abs().max()) <= 1e-6 * float(kg.abs().max())
    model = SingleTaskVariationalGP(init_points=x.view(-1, 1), likelihood=lh, use_piv_chol_init=False,
                                    mean_module=gp.ConstantMean(), covar_module=kern,
                                    learn_inducing_locations=False, use_whitened_var_strat=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.initialize_variational_parameters(lh, x, y=yy)
    d = model.variational_strategy._variational_distribution
    assert float((d.variational_mean.detach().cpu().double() - torch.tensor(g[f"{tag}_mean"]).double()).abs().max()) < 1e-5
    assert abs(float(model.mean_module.constant.detach()) - float(g[f"{tag}_const"].reshape(-1)[0])) < 1e-5
    S_hip, S_ref = d.chol_variational_covar.cpu().double(), torch.tensor(g[f"{tag}_chol"]).double()
    cov_h, cov_r = S_hip @ S_hip.mT, S_ref @ S_ref.mT
    assert float((cov_h - cov_r).abs().max() / cov_r.abs().max()) < 2e-3
    f = torch.tensor(g["lik_f"]).cuda()
    assert torch.allclose(lh.forward(f).scale.cpu(), torch.tensor(g["lik_scale"]), rtol=1e-6, atol=0)


@pytest.mark.parametrize("kernel", ["bm", "fbm"])
def test_learn_gpcv_tracks_oracle(kernel):
    """40 Adam iterations of LearnGPCV (train_utils.py:15-67) against the fp64 oracle loop.  Adam's first steps are
    +-lr whatever the gradient's size, so parameters whose gradient is at the fp32 noise floor wander by O(lr) per
    step; the loss and the variational mean are insensitive to that and are held tight, the readout (which mixes
    in 10 x N normal draws through Lq) to 3 %."""
    import warnings
    from volt_amd.train_utils import FitGPCV
    n, iters = 250, 40
    F, _ = _prices(n, 2021)
    x = torch.arange(n, dtype=torch.float32) / 252
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m0, _, _ = FitGPCV(x.cuda(), F.cuda(), train_iters=0, kernel=kernel)
        d0 = m0.variational_strategy._variational_distribution
        init = (d0.variational_mean.detach().cpu(), d0.chol_variational_covar.detach().cpu(),
                m0.mean_module.constant.detach().cpu().reshape(()))
        model, lh, losses = FitGPCV(x.cuda(), F.cuda(), train_iters=iters, kernel=kernel)
    eps = torch.randn(10, n, generator=torch.Generator().manual_seed(7))
    rec = []
    ref, ps = GO.learn_gpcv(x, F, train_iters=iters, kernel=kernel, eps=eps.double(), dtype=torch.float64, record=rec,
                            init=init)
    got = torch.stack(losses).cpu().double()
    want = torch.tensor(rec, dtype=torch.float64)
    assert float(((got - want).abs() / want.abs().clamp_min(1.0)).max()) < 5e-4
    assert abs(float(got[0] - want[0])) < 1e-4 * abs(float(want[0]))          # same start: one step, no optimiser
    d = model.variational_strategy._variational_distribution
    assert float((d.variational_mean.detach().cpu().double() - ps[0]).abs().max()) < 5e-3
    assert abs(float(model.covar_module.raw_vol.detach()) - float(ps[3])) < 5e-3
    assert abs(float(model.mean_module.constant.detach()) - float(ps[2])) < 5e-3
    latent = model(x.cuda())
    f = latent.rsample(base_samples=eps.cuda())
    vol = lh(f).scale.mean(0).cpu().double()
    assert float((vol - ref).abs().max() / ref.abs().max()) < 3e-2


def test_learn_gpcv_batched_equals_single():
    """T series in one batched fit == T separate fits (independent GPs)."""
    import warnings
    from volt_amd.train_utils import LearnGPCV
    n, iters = 200, 15
    Fs = torch.stack([_prices(n, 2019 + i)[0] for i in range(3)])
    x = (torch.arange(n, dtype=torch.float32) / 252).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(3)
        vb = LearnGPCV(x, Fs.cuda(), train_iters=iters)
        assert vb.shape == (3, n)
        torch.manual_seed(3)
        eps = torch.randn(10, 3, n, device="cuda")
        for i in range(3):
            # same normal draws as the batched readout used for series i
            import volt_amd.variational as V
            orig = torch.randn
            try:
                torch.randn = lambda *a, **k: eps[:, i, :].clone()
                vi = LearnGPCV(x, Fs[i].cuda(), train_iters=iters)
            finally:
                torch.randn = orig
            assert float((vi - vb[i]).abs().max() / vb[i].abs().max()) < 2e-3


def test_reference_pipeline_end_to_end():
    """The stock driver's body (experiments/stocks/GenerateMultiMeanPreds.py:95-107), statement for statement:
    LearnGPCV -> TrainVolModel -> TrainVoltMagpieModel -> Rollouts, every stage on the HIP path."""
    import warnings
    from volt_amd.train_utils import LearnGPCV, TrainVolModel, TrainVoltMagpieModel
    from volt_amd.rollout_utils import Rollouts
    n, H, S = 150, 6, 16
    F, _ = _prices(n, 2030)
    train_y = F.cuda()
    train_x = (torch.arange(train_y.shape[0] - 1) / 252.).cuda()
    test_x = (torch.arange(H) / 252.).cuda() + train_x[-1] + train_x[1]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vol = LearnGPCV(train_x, train_y, train_iters=30, printing=False)
        assert vol.shape == (n,) and bool((vol > 0).all())
        vmod, vlh = TrainVolModel(train_x, vol, train_iters=10, printing=False)
        voltron, lh = TrainVoltMagpieModel(train_x, train_y[1:], vmod, vlh, vol, printing=False, train_iters=10, k=20,
                                           mean_func="ewma")
        vmod.eval()
        samples = Rollouts(train_x, train_y, test_x, voltron, nsample=S)
    assert tuple(samples.shape) == (S, H) and bool(torch.isfinite(samples).all())
    assert float((samples[:, 0].mean() - train_y[-1].log().cpu()).abs()) < 0.5


def test_batched_vol_model_equals_per_series():
    """TrainVolModelBatch (T vol models in one batched BMGP) == T runs of TrainVolModel (train_utils.py:69-95), and the
    batched posterior equals the per-series posteriors."""
    import warnings
    from volt_amd.train_utils import TrainVolModel, TrainVolModelBatch
    n, T, H, iters = 90, 3, 5, 12
    vols = torch.stack([torch.tensor(sde_series(n, 50 + i)[1]) for i in range(T)]).cuda()
    tx = (torch.arange(n) / 252.).cuda()
    test_x = (torch.arange(H) / 252.).cuda() + tx[-1] + tx[1]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vb, lb = TrainVolModelBatch(tx, vols, train_iters=iters)
        vb.eval()
        pb = vb(test_x)
        assert tuple(pb.mean.shape) == (T, H) and tuple(pb.covariance_matrix.shape) == (T, H, H)
        s = pb.sample(torch.Size((7,)))
        assert tuple(s.shape) == (7, T, H) and bool(torch.isfinite(s).all())
        for i in range(T):
            vi, li = TrainVolModel(tx, vols[i], train_iters=iters)
            assert abs(float(vi.covar_module.raw_vol.detach()) - float(vb.covar_module.raw_vol[i].detach())) < 1e-4
            assert abs(float(li.raw_noise.detach()) - float(lb.raw_noise[i].detach())) < 1e-4
            vi.eval()
            pi = vi(test_x)
            assert float((pi.mean - pb.mean[i]).abs().max()) < 2e-3
            assert float((pi.covariance_matrix - pb.covariance_matrix[i]).abs().max()) < 2e-3 * float(pi.covariance_matrix.abs().max()) + 1e-6


def test_expected_log_prob_agrees_with_the_fused_step():
    """VolatilityGaussianLikelihood.expected_log_prob (volatility_likelihood.py:52-57; torch, per point) sums to the
    likelihood term the HIP step computes."""
    import warnings
    from volt_amd import ops
    from volt_amd.train_utils import FitGPCV
    from volt_amd.variational import num_gauss_hermite_locs
    n = 140
    F, _ = _prices(n, 77)
    x = (torch.arange(n, dtype=torch.float32) / 252).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, lik, _ = FitGPCV(x, F.cuda(), train_iters=3)
    yy = GO.scaled_returns(x.cpu(), F).cuda()
    latent = model(x)
    with num_gauss_hermite_locs(75), torch.no_grad():
        per_point = lik.expected_log_prob(yy, latent)
    assert tuple(per_point.shape) == (n,)
    d = model.variational_strategy._variational_distribution
    gh_x, gh_w = GO.gauss_hermite(75)
    K = GO.bm_cov(x.cpu().double(), torch.tensor(0.2, dtype=torch.float64)).float().cuda().unsqueeze(0)
    ws = ops.gpcv_step(K, d.variational_mean.detach().reshape(1, n), d.variational_mean.detach().reshape(1, n),
                       d.chol_variational_covar.detach().reshape(1, n, n), yy.reshape(1, n), gh_x.cuda(),
                       (gh_w / math.sqrt(math.pi)).cuda())
    assert abs(float(per_point.sum()) - float(ws.out[0, 0])) < 2e-4 * abs(float(ws.out[0, 0]))
