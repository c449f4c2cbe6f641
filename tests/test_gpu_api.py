"""The voltron-shaped Python surface (kernels / means / models / train_utils / rollout_utils) on the
MI355X, checked against golden vectors produced by the reference's own code and against the oracle.
Reads like the reference's call sites on purpose."""
import copy

import numpy as np
import pytest
import torch

from oracle import volt_oracle as vo
from volt_amd.synthetic import sde_batch, sde_series, rollout_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    assert torch.cuda.is_available()
    import volt_amd
    return volt_amd


def dev(a, dtype=None):
    t = torch.as_tensor(a)
    return (t if dtype is None else t.to(dtype)).cuda()


# ------------------------------------------------------------------ kernels
@pytest.mark.parametrize("tag", ["n7", "n64", "n257", "b3n50", "b2n130"])
def test_volatility_kernel_class_golden(va, golden, tag):
    from volt_amd.kernels import VolatilityKernel
    g = golden("fill")
    vol, x = dev(g[f"{tag}_vol"]), dev(g[f"{tag}_x"])
    xx = x if vol.ndim == 1 else x.unsqueeze(0).repeat(vol.shape[0], 1)
    kern = VolatilityKernel()
    K = kern.forward(xx.unsqueeze(-1), vol.unsqueeze(-1))
    assert np.array_equal(K.cpu().numpy(), g[f"{tag}_K"])
    d = kern.forward(xx.unsqueeze(-1), vol.unsqueeze(-1), diag=True)
    assert np.array_equal(d.cpu().numpy(), g[f"{tag}_diag"])
    assert np.array_equal(kern(xx.unsqueeze(-1), vol.unsqueeze(-1)).evaluate().cpu().numpy(), g[f"{tag}_K"])


# ------------------------------------------------------------------ means
@pytest.mark.parametrize("tag,k", [("n60k5", 5), ("n60k25", 25), ("b3n40k7", 7), ("n300k100", 100)])
def test_mean_classes_golden(va, golden, tag, k):
    from volt_amd import means
    g = golden("ewma")
    y, x = dev(g[f"{tag}_y"]), dev(g[f"{tag}_x"])
    np.testing.assert_allclose(means.EWMA(y, k).cpu().numpy(), g[f"{tag}_ewma"], rtol=2e-6)
    n = x.shape[0]
    for cname, cls in (("ewma", means.EWMAMean), ("dewma", means.DEWMAMean), ("tewma", means.TEWMAMean),
                       ("meanrevert", means.MeanRevertingEMAMean)):
        mod = cls(x, y, k)
        for branch, xq in (("train", x), ("one", x[-1:] + 1 / 252.), ("other", x[: n // 2])):
            out = mod(xq)
            ref = g[f"{tag}_{cname}_{branch}"]
            assert tuple(out.shape) == ref.shape, (cname, branch)
            assert out.is_cuda
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ MLL autograd
def test_mll_autograd_matches_oracle(va):
    from volt_amd.gp import ExactMarginalLogLikelihood, GaussianLikelihood, MultivariateNormal
    B, n = 3, 320
    x, F, vol = sde_batch(B, n)
    K = vo.volatility_kernel(np.repeat(x[None], B, 0)[..., None], vol[..., None])
    y = np.log(F[:, 1:])
    lh = GaussianLikelihood(batch_shape=torch.Size([B])).cuda()
    lh.raw_noise.data.fill_(1e-5)
    mean = torch.full((B, n), 2.3, device="cuda", requires_grad=True)
    mll = ExactMarginalLogLikelihood(lh, None)
    val = mll(MultivariateNormal(mean, dev(K)), dev(y))
    assert val.shape == (B,)
    (-val.sum()).backward()
    o = vo.mll_and_grads(K, y, np.full((B, n), 2.3, dtype=np.float32), 1e-5)
    np.testing.assert_allclose(val.detach().cpu().numpy(), o["mll"], rtol=2e-5)
    np.testing.assert_allclose(-lh.raw_noise.grad.cpu().numpy()[:, 0], o["d_raw"], rtol=1e-3)
    gm = -mean.grad.cpu().numpy()
    assert np.abs(gm - o["d_mean"]).max() <= 1e-4 * np.abs(o["d_mean"]).max()


def test_mll_jitter_ladder_like_psd_safe_cholesky(va):
    """A matrix that fails plain fp32 Cholesky but passes with gpytorch's 1e-6..1e-4 jitter ladder warns
    and returns a finite value."""
    import warnings
    from volt_amd.gp import ExactMarginalLogLikelihood, GaussianLikelihood, MultivariateNormal, NumericalWarning
    n = 256
    u = torch.randn(n, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    K = u @ u.T - 2e-5 * torch.eye(n, device="cuda")            # rank 3, slightly indefinite
    lh = GaussianLikelihood().cuda()
    lh.noise_covar.raw_noise.data.fill_(-30.0)                    # sigma^2 = 1e-4 floor: K + s2 I barely not PD in fp32
    K = K - 1.0e-4 * torch.eye(n, device="cuda")                  # cancel the floor: eigenvalues ~ -2e-5
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        val = ExactMarginalLogLikelihood(lh, None)(MultivariateNormal(torch.zeros(n, device="cuda"), K),
                                                    torch.zeros(n, device="cuda"))
    assert torch.isfinite(val)
    assert any(issubclass(x.category, NumericalWarning) for x in w)


def test_mll_raises_on_non_pd(va):
    from volt_amd.gp import ExactMarginalLogLikelihood, GaussianLikelihood, MultivariateNormal, NotPSDError
    n = 130
    K = -5 * torch.eye(n, device="cuda")
    lh = GaussianLikelihood().cuda()
    with pytest.raises(NotPSDError):
        ExactMarginalLogLikelihood(lh, None)(MultivariateNormal(torch.zeros(n, device="cuda"), K),
                                              torch.zeros(n, device="cuda"))


# ------------------------------------------------------------------ training loops
def _oracle_adam(K, y, mean, iters, lr=0.1):
    raw = torch.tensor([1e-5], dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([raw], lr=lr)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        o = vo.mll_and_grads(K, y, mean, float(raw))
        losses.append(-float(o["mll"]))
        raw.grad = torch.tensor([-o["d_raw"]], dtype=torch.float64)
        opt.step()
    return np.array(losses), float(raw)


@pytest.mark.parametrize("mean_func", ["ewma", "dewma"])
def test_train_volt_magpie_model_tracks_oracle(va, mean_func):
    """The reference loop (train_utils.py:192-257) on the device vs the same Adam recursion driven by the
    fp64 oracle: losses agree to 1e-4 over 12 iterations (sigma^2 moves from 0.69 to ~0.2)."""
    from volt_amd.train_utils import TrainVoltMagpieModel
    n, k, iters = 255, 20, 12
    F, vol = sde_series(n, 2021)
    train_x = torch.arange(n, device="cuda") / 252.
    prices = dev(F)
    model, lh = TrainVoltMagpieModel(train_x, prices[1:], None, None, dev(vol), train_iters=iters, k=k,
                                     mean_func=mean_func)
    x = (np.arange(n) / 252.).astype(np.float32)
    K = vo.volatility_kernel(x, vol)
    y = np.log(F[1:])
    mfn = vo.ewma_mean if mean_func == "ewma" else vo.dewma_mean
    losses, raw_end = _oracle_adam(K, y, mfn(x, x, y, k), iters)
    assert abs(float(lh.raw_noise) - raw_end) < 2e-3 * max(1.0, abs(raw_end))
    # last loss of the device loop, recomputed
    from volt_amd.gp import ExactMarginalLogLikelihood
    with torch.no_grad():
        model.train()
        last = -ExactMarginalLogLikelihood(lh, model)(model(train_x), prices[1:].log())
    o = vo.mll_and_grads(K, y, mfn(x, x, y, k), float(lh.raw_noise))
    assert abs(float(last) + float(o["mll"])) < 1e-4 * max(1.0, abs(float(o["mll"])))
    # only the likelihood noise is trainable with an EWMA-family mean (train_utils.py:201)
    flags = [p.requires_grad for p in model.parameters()]
    assert flags[0] is True and not any(flags[1:])


def test_train_data_model_loglinear(va):
    from volt_amd.train_utils import TrainDataModel
    n = 200
    F, vol = sde_series(n, 7)
    train_x = torch.arange(n, device="cuda") / 252.
    model, lh = TrainDataModel(train_x, dev(F)[1:], None, None, dev(vol), train_iters=5)
    assert model.mean_module.weights.grad is not None and model.mean_module.bias.grad is not None
    assert torch.isfinite(lh.raw_noise).all()
    assert model.train_cov.shape == (n, n)


def test_batched_training_matches_per_series(va):
    from volt_amd.train_utils import TrainVoltMagpieBatch, TrainVoltMagpieModel
    B, n, iters = 3, 200, 6
    x, F, vol = sde_batch(B, n)
    tx = dev(x)
    model, lh, losses = TrainVoltMagpieBatch(tx, dev(F[:, 1:]), dev(vol), train_iters=iters, k=10)
    for b in range(B):
        m1, l1 = TrainVoltMagpieModel(tx, dev(F[b, 1:]), None, None, dev(vol[b]), train_iters=iters, k=10)
        assert abs(float(l1.raw_noise) - float(lh.raw_noise[b])) < 1e-4


# ------------------------------------------------------------------ rollouts
class _ConstMean(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c

    def forward(self, x):
        return torch.full((x.shape[0],), self.c, device=x.device)


def test_generate_prediction_multipoint_golden(va, golden):
    from volt_amd.kernels import VolatilityKernel
    from volt_amd.rollout_utils import GeneratePrediction
    g = golden("rollouts")

    class M:
        pass
    m = M()
    m.train_x = dev(g["gpm_train_x"])
    m.train_y = dev(g["gpm_train_y"])[1:].log()
    m.log_vol_path = dev(g["gpm_vol_path"]).log()
    m.mean_module = _ConstMean(float(g["gpm_const"]))
    m.covar_module = VolatilityKernel()
    out = GeneratePrediction(m.train_x, dev(g["gpm_train_y"]), dev(g["gpm_test_x"]), dev(g["gpm_pred_vol"]), m,
                             z=dev(g["gpm_z"]))
    assert tuple(out.shape) == g["gpm_samples"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["gpm_samples"], atol=2e-3, rtol=0)


RO = [("ewma", "ewma"), ("ewma_theta", "ewma"), ("dewma", "dewma"), ("tewma", "tewma"), ("ewma_n120", "ewma")]


def _golden_model(g, tag, mean):
    from volt_amd import means
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltMagpie
    tx, ty = dev(g[f"{tag}_train_x"]), dev(g[f"{tag}_train_y"])
    k = int(g[f"{tag}_k"])
    model = VoltMagpie(tx, ty[1:].log(), GaussianLikelihood().cuda(), dev(g[f"{tag}_vol_path"]), k=k)
    cls = {"ewma": means.EWMAMean, "dewma": means.DEWMAMean, "tewma": means.TEWMAMean}[mean]
    model.mean_module = cls(tx, ty[1:].log(), k)
    return model, tx, ty


@pytest.mark.parametrize("engine", ["dense", "bordered"])
@pytest.mark.parametrize("tag,mean", RO)
def test_rollouts_golden_pathwise(va, golden, tag, mean, engine):
    """Reference Rollouts outputs (tests/golden/rollouts.npz) with the same pred_vol and N(0,1) draws.
    Tolerance 2e-3 abs on log-prices (one-step predictive sd ~ 1e-2): the reference factors the
    noise-free K in fp32, so its own outputs carry round-off of this size."""
    from volt_amd.rollout_utils import Rollouts
    g = golden("rollouts")
    model, tx, ty = _golden_model(g, tag, mean)
    theta = float(g[f"{tag}_theta"])
    theta = None if np.isnan(theta) else theta
    S, H = g[f"{tag}_z"].shape
    out = Rollouts(tx, ty, dev(g[f"{tag}_test_x"]), model, nsample=S, theta=theta,
                   pred_vol=dev(g[f"{tag}_pred_vol"]), z=dev(g[f"{tag}_z"]), engine=engine)
    assert not out.is_cuda and tuple(out.shape) == (S, H)          # CPU result like the reference (:65)
    np.testing.assert_allclose(out.numpy(), g[f"{tag}_samples"], atol=2e-3, rtol=0)
    # the model is left mutated as the reference leaves it (:80-86)
    n = tx.shape[0]
    assert tuple(model.train_y.shape) == (S, n + H - 1) and tuple(model.log_vol_path.shape) == (S, n + H - 1)
    assert tuple(model.train_x.shape) == (n + H - 1,)


@pytest.mark.parametrize("mean_name", ["ewma", "dewma", "tewma", "meanrevert"])
def test_bordered_vs_oracle_exact_conditional(va, mean_name):
    """Against the fp64 exact conditional (last residual + mean + sqrt(1/2 dx v^2) z) driven by
    the oracle's means, N = 399 (the reference's default ntrain), S = 16, H = 24."""
    from volt_amd import means
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltMagpie
    from volt_amd.rollout_utils import Rollouts
    n, S, H, k = 399, 16, 24, 25
    F, vol = sde_series(n, 11)
    pv, z = rollout_inputs(vol[-1], S, H, seed=5)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    model = VoltMagpie(tx, dev(F)[1:].log(), GaussianLikelihood().cuda(), dev(vol), k=k)
    cls = {"ewma": means.EWMAMean, "dewma": means.DEWMAMean, "tewma": means.TEWMAMean,
           "meanrevert": means.MeanRevertingEMAMean}[mean_name]
    model.mean_module = cls(tx, dev(F)[1:].log(), k)
    out = Rollouts(tx, dev(F), test_x, copy.deepcopy(model), nsample=S, pred_vol=dev(pv), z=dev(z),
                   engine="bordered").numpy()
    # exact recursion in fp64
    x = (np.arange(n) / 252.).astype(np.float32)
    logy = np.log(F[1:]).astype(np.float32)
    fn = {"ewma": vo.ewma_mean, "dewma": vo.dewma_mean, "tewma": vo.tewma_mean, "meanrevert": vo.meanrevert_mean}[mean_name]
    lat = logy.mean(dtype=np.float32)
    ref = np.zeros((S, H))
    for s in range(S):
        ys = logy.copy()
        xs = x.copy()
        for i in range(H):
            kw = {"latent": lat} if mean_name == "meanrevert" else {}
            m_all = fn(np.arange(len(ys) + 1), xs, ys, k, **kw) if False else None
            full = {"ewma": vo.ewma(ys, k)}.get(mean_name)
            if mean_name == "ewma":
                mtr, mnew = full[:-1], full[-1]
            else:
                tr = fn(xs, xs, ys, k, **kw)
                one = fn(np.array([0.0]), xs, ys, k, **kw)
                mtr, mnew = tr, one[0]
            resid_last = ys[-1] - mtr[-1]
            sd = np.sqrt(0.5 * (1 / 252.) * float(pv[s, i]) ** 2)
            ref[s, i] = resid_last + mnew + sd * z[s, i]
            ys = np.append(ys, np.float32(ref[s, i]))
            xs = np.append(xs, np.float32(xs[-1] + 1 / 252.))
    # fp32 factor of the noise-free K (cond ~1e6 at N=399): a few 1e-4 on values ~2.3
    np.testing.assert_allclose(out, ref, atol=1.5e-3, rtol=0)


# ------------------------------------------------------------------ next row (f)1: vol-path forecaster
def test_train_vol_model_gradients_match_fp64_autograd(va):
    """BMGP + exact MLL (train_utils.py:69-95): d mll / d raw_vol and d / d raw_noise from the HIP step vs
    fp64 autograd of the dense formula, then a short training run stays finite and decreases the loss."""
    from volt_amd.gp import ExactMarginalLogLikelihood
    from volt_amd.train_utils import TrainVolModel
    n = 200
    F, vol = sde_series(n, 13)
    tx = torch.arange(n, device="cuda") / 252. + 1 / 252.
    model, lh = TrainVolModel(tx, dev(vol), train_iters=0)
    model.train()
    mll = ExactMarginalLogLikelihood(lh, model)
    loss = -mll(model(tx), dev(vol).log())
    loss.backward()
    g_vol, g_noise = float(model.covar_module.raw_vol.grad), float(lh.raw_noise.grad)
    # fp64 reference
    x = tx.double().cpu()
    y = torch.tensor(np.log(vol), dtype=torch.float64)
    raw_vol = torch.tensor([float(model.covar_module.raw_vol)], dtype=torch.float64, requires_grad=True)
    raw_noise = torch.tensor([float(lh.raw_noise)], dtype=torch.float64, requires_grad=True)
    v = torch.sigmoid(raw_vol)
    K = v * torch.minimum(x[:, None], x[None, :]) + (torch.nn.functional.softplus(raw_noise) + 1e-4) * torch.eye(n, dtype=torch.float64)
    mean = -0.5 * v ** 2 * x
    ref = -torch.distributions.MultivariateNormal(mean, covariance_matrix=K).log_prob(y) / n
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1, abs(float(ref)))
    assert abs(g_vol - float(raw_vol.grad)) < 2e-3 * max(abs(float(raw_vol.grad)), 1e-3)
    assert abs(g_noise - float(raw_noise.grad)) < 2e-3 * max(abs(float(raw_noise.grad)), 1e-3)
    model2, lh2 = TrainVolModel(tx, dev(vol), train_iters=25)
    with torch.no_grad():
        model2.train()
        end = -ExactMarginalLogLikelihood(lh2, model2)(model2(tx), dev(vol).log())
    assert torch.isfinite(end) and float(end) < float(loss)


# ------------------------------------------------------------------ next row (f)3: driver loop + output format
def test_batched_forecast_driver_writes_reference_layout(va, tmp_path):
    from volt_amd.forecast import GenerateStockPredictionsBatch
    B, T, ntrain, H, S = 3, 70, 64, 4, 5
    x, F, vol = sde_batch(B, T - 1, seed=77)
    closes = dev(F)                                              # [B, T]
    g = torch.Generator(device="cuda").manual_seed(1)
    out = GenerateStockPredictionsBatch(["AAA", "BBB", "CCC"], closes, forecast_horizon=H, train_iters=3, nsample=S,
                                        ntrain=ntrain, mean="ewma", save=True, k=10, ntimes=2, vol_iters=2,
                                        par_dir=str(tmp_path), generator=g)
    assert tuple(out.shape) == (B, S, H) and torch.isfinite(out).all()
    files = sorted(p.name for p in (tmp_path / "BBB").iterdir())
    assert len(files) == 2 and all(f.startswith("volt_ewma10_") and f.endswith(".pt") for f in files)
    saved = torch.load(tmp_path / "CCC" / files[-1])
    assert tuple(saved.shape) == (S, H) and torch.equal(saved, out[2])
    # log-price scale: forecasts start near the last observed log price
    assert float((out[:, :, 0].mean(1) - closes[:, ntrain * 0 + T - 1 - ((T - ntrain) % 3) * 0 - 1].log().cpu()).abs().max()) < 0.5


@pytest.mark.parametrize("mean", ["ewma", "dewma", "constant", "loglinear"])
def test_batched_driver_matches_per_series_functions(va, mean):
    """the batched driver's samples equal what the reference's per-ticker body produces with the same
    vol path, vol-forecast sample and N(0,1) draws -- Rollouts for the EWMA family (GenerateMultiMeanPreds.py:110-112),
    one multi-point GeneratePrediction for a standard mean (:113-119)."""
    from volt_amd import gp
    from volt_amd.forecast import GenerateStockPredictionsBatch, realised_vol
    from volt_amd.means import DEWMAMean, LogLinearMean
    from volt_amd.models import VoltMagpie
    from volt_amd.rollout_utils import GeneratePrediction, Rollouts
    B, T, ntrain, H, S, k = 3, 90, 80, 6, 7, 10
    x, F, vol = sde_batch(B, T - 1, seed=91)
    closes = dev(F)
    g = torch.Generator(device="cuda").manual_seed(4)
    dbg = {}
    out = GenerateStockPredictionsBatch(["A", "B", "C"], closes, forecast_horizon=H, train_iters=5, nsample=S,
                                        ntrain=ntrain, mean=mean, k=k, ntimes=1, vol_iters=2, vol_fn=realised_vol,
                                        generator=g, debug=dbg)
    assert tuple(out.shape) == (B, S, H) and torch.isfinite(out).all()
    train_x = torch.arange(ntrain - 1, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + train_x[-1] + train_x[1]
    for b in range(B):
        ty = dbg["train_y"][b]                                            # [ntrain] prices of the last window
        m = VoltMagpie(train_x, ty[1:].log(), gp.GaussianLikelihood().cuda(), dbg["vol"][b], k=k)
        if mean == "dewma":
            m.mean_module = DEWMAMean(train_x, ty[1:].log(), k)
        if mean in ("ewma", "dewma"):
            ref = Rollouts(train_x, ty, test_x, m, nsample=S, pred_vol=dbg["pred_vol"][b], z=dbg["z"][b])
        else:
            bm = dbg["model"].mean_module
            if mean == "constant":
                m.mean_module = gp.ConstantMean().cuda()
                m.mean_module.constant.data = bm.constant.data[b].clone()
            else:
                m.mean_module = LogLinearMean(1).cuda()
                m.mean_module.weights.data = bm.weights.data[b].clone()
                m.mean_module.bias.data = bm.bias.data[b].clone()
            ref = GeneratePrediction(train_x, ty, test_x, dbg["pred_vol"][b], m, z=dbg["z"][b]).detach().cpu()      # :118
        np.testing.assert_allclose(out[b].numpy(), ref.numpy(), atol=2e-4, rtol=0)
    if mean in ("constant", "loglinear"):                                # the per-series mean parameters were trained
        p0 = [p for p in dbg["model"].mean_module.parameters()][0]
        assert p0.shape[0] == B and float(p0.detach().std()) > 0


def test_graph_captured_loop_on_a_long_series(va):
    """One series of 1 200 points runs its gradient step as ONE launch (long_step_kernel: host-made piece list, slabs,
    a step counter on the device) -- captured once and replayed, it must train like the eager loop."""
    from volt_amd.train_utils import TrainVoltMagpieModel
    n = 1200
    F, vol = sde_series(n, 11)
    tx = torch.arange(n, device="cuda") / 252.
    prices, v = dev(F), dev(vol)
    out = {}
    for graph in (False, True):
        m, lh = TrainVoltMagpieModel(tx, prices[1:], None, None, v, train_iters=25, k=50, graph=graph)
        out[graph] = float(lh.raw_noise.detach())
    assert abs(out[False] - out[True]) < 2e-3 * max(1.0, abs(out[False])), out


def test_graph_captured_training_loops_match_eager(va):
    """train_utils graph=True: the warm-up iterations run eagerly, then ONE captured iteration (every HIP launch of the
    step + the capturable Adam update) is replayed.  Same number of optimiser steps, same arithmetic: the trained
    parameters agree with the eager loop to fp32 noise; a non-PD matrix inside the captured loop is reported after it."""
    from volt_amd import gp
    from volt_amd.train_utils import TrainVolModel, TrainVoltMagpieModel
    n = 200
    F, vol = sde_series(n, 5)
    tx = torch.arange(n, device="cuda") / 252.
    prices, v = dev(F), dev(vol)
    out = {}
    for graph in (False, True):
        vmod, vlh = TrainVolModel(tx, v, train_iters=60, graph=graph)
        m, lh = TrainVoltMagpieModel(tx, prices[1:], vmod, vlh, v, train_iters=60, k=20, graph=graph)
        out[graph] = (float(vmod.covar_module.raw_vol.detach()), float(vlh.raw_noise.detach()), float(lh.raw_noise.detach()))
    for a, b in zip(out[False], out[True]):
        assert abs(a - b) < 2e-3 * max(1.0, abs(a)), out
    # the deferred check: a covariance that is not PD inside the captured loop raises after the loop
    with pytest.raises(gp.NotPSDError):
        with gp.deferred_checks() as chk:
            lik = gp.GaussianLikelihood().cuda()
            mll = gp.ExactMarginalLogLikelihood(lik, None)
            Kbad = -torch.eye(64, device="cuda")
            mll(gp.MultivariateNormal(torch.zeros(64, device="cuda"), Kbad), torch.ones(64, device="cuda"))
            chk.raise_if_bad()


def test_model_method_generate_prediction_twin(va):
    """VoltronGP.GeneratePrediction(test_x, pred_vol, n_sample) (VoltronGP.py:62-95, the notebook's cell 15):
    H-point joint prediction with n_sample draws, checked against the oracle's GeneratePrediction restatement
    with the same N(0,1) draws (CPU generator, as torch.randn(...) in the reference)."""
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltronGP
    n, T, ns = 90, 6, 3
    F, vol = sde_series(n, 31)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(T, device="cuda") / 252. + tx[-1] + tx[1]
    m = VoltronGP(tx, dev(F)[1:].log(), GaussianLikelihood().cuda(), dev(vol))
    with torch.no_grad():
        m.mean_module.weights.fill_(0.3)
        m.mean_module.bias.fill_(2.2)
    pv = dev(np.full(T, vol[-1], dtype=np.float32))
    torch.manual_seed(11)
    out = m.GeneratePrediction(test_x, pv, ns)                      # [T, ns]
    torch.manual_seed(11)
    z = torch.randn(T, ns).numpy()
    x = (np.arange(n) / 252.).astype(np.float32)
    txn = test_x.cpu().numpy()
    lin = lambda q: (0.3 * np.asarray(q, dtype=np.float32).reshape(-1) + 2.2).astype(np.float32)
    assert tuple(out.shape) == (T, ns)
    for c in range(ns):
        ref = vo.generate_prediction(x, np.log(F[1:]), np.log(m.log_vol_path.exp().cpu().numpy()), txn,
                                     pv.cpu().numpy()[None], z[None, :, c:c + 1], lin, jitter=None)
        np.testing.assert_allclose(out[:, c].cpu().numpy(), ref[0], atol=2e-3, rtol=0)
    # n_sample = 1 (the default): the reference squeezes the sample axis away, VoltronGP.py:93-95 / VoltMagpie.py:96-99
    torch.manual_seed(11)
    one = m.GeneratePrediction(test_x, pv)
    torch.manual_seed(11)
    z1 = torch.randn(T, 1).numpy()
    assert tuple(one.shape) == (T,)
    ref1 = vo.generate_prediction(x, np.log(F[1:]), np.log(m.log_vol_path.exp().cpu().numpy()), txn,
                                  pv.cpu().numpy()[None], z1[None], lin, jitter=None)
    np.testing.assert_allclose(one.cpu().numpy(), ref1[0], atol=2e-3, rtol=0)
    m.vol_model.eval()
    s = m.SamplePrediction(test_x, n_sample=2)
    assert tuple(s.shape) == (T, 2) and torch.isfinite(s).all()
    assert tuple(m.SamplePrediction(test_x).shape) == (T,)
    assert tuple(m.MeanPrediction(test_x).shape) == (T,)


def test_full_length_training_run_tracks_oracle(va):
    """The reference's loop for 200 Adam iterations at its default size (N=399): sigma^2 is driven from 0.69 to the
    1e-4 floor (cond(K + s2 I) grows to ~1e7) and the device trajectory must still end where the fp64 oracle's does."""
    from volt_amd.gp import ExactMarginalLogLikelihood
    from volt_amd.train_utils import TrainVoltMagpieModel
    n, k, iters = 399, 25, 200
    F, vol = sde_series(n, 2024)
    tx = torch.arange(n, device="cuda") / 252.
    model, lh = TrainVoltMagpieModel(tx, dev(F)[1:], None, None, dev(vol), train_iters=iters, k=k)
    x = (np.arange(n) / 252.).astype(np.float32)
    K = vo.volatility_kernel(x, model.log_vol_path.exp().cpu().numpy())
    y = np.log(F[1:])
    mean = vo.ewma_mean(x, x, y, k)
    losses, raw_end = _oracle_adam(K, y, mean, iters)
    assert abs(float(lh.raw_noise) - raw_end) < 1e-4 * abs(raw_end)
    assert float(lh.noise) < 2e-4                                   # reached the neighbourhood of the floor
    with torch.no_grad():
        model.train()
        last = float(-ExactMarginalLogLikelihood(lh, model)(model(tx), dev(F)[1:].log()))
    o = vo.mll_and_grads(K, y, mean, raw_end)
    assert abs(last + float(o["mll"])) < 1e-4 * abs(float(o["mll"]))
    assert losses[-1] < losses[0]


def test_batched_wind_driver_writes_reference_layout(va, tmp_path):
    """experiments/weather/GPGenerator.py volt/ewma branch, all stations per window: window schedule (:33-34), +1 offset
    and -99 handling (:47,55), theta = 0.01 rollouts, file names (:103-106)."""
    from volt_amd.forecast import GenerateWindPredictionsBatch
    B, T, ntrain, H, S = 3, 120, 64, 5, 6
    rng = np.random.RandomState(3)
    wind = np.abs(np.cumsum(rng.normal(0, 0.3, size=(B, T)), axis=1) + 6).astype(np.float32)
    wind[1, 10] = -99.0                                                    # a missing reading
    g = torch.Generator(device="cuda").manual_seed(2)
    out = GenerateWindPredictionsBatch([0, 1, 2], dev(wind), forecast_horizon=H, ntrain=ntrain, n_test_times=3, nsample=S,
                                       k=20, gpcv_iters=3, vol_iters=2, save=True, par_dir=str(tmp_path), generator=g)
    assert tuple(out.shape) == (B, S, H) and torch.isfinite(out).all()
    step = int((T - H - ntrain) / 3)
    want = [f"volt_ema20_theta0.01_{d}.pt" for d in range(ntrain, T - H, step)]
    assert sorted(p.name for p in (tmp_path / "stn1").iterdir()) == sorted(want)
    saved = torch.load(tmp_path / "stn2" / want[-1])
    assert tuple(saved.shape) == (S, H) and torch.equal(saved, out[2])


def test_fused_adam_matches_torch_adam():
    """optim.FusedAdam (two launches per step, device-side step count) against torch.optim.Adam(lr=0.1) -- the optimiser
    every training loop of the reference builds (train_utils.py:43,100,166,238,291): same trajectory over 30 steps on
    parameters of the shapes those loops have (scalars, a vector, a 300 x 300 matrix), eagerly and replayed from a graph."""
    from volt_amd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(1,), (), (300,), (300, 300), (1, 1)]
    target = [torch.randn(s, device="cuda") for s in shapes]

    def make():
        torch.manual_seed(4)
        return [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]

    def loss_of(ps):
        return sum(((p - t) ** 2 * (1.0 + 0.1 * i)).sum() for i, (p, t) in enumerate(zip(ps, target)))

    pa, pb, pc = make(), make(), make()
    oa, ob, oc = torch.optim.Adam(pa, lr=0.1), FusedAdam(pb, lr=0.1), FusedAdam(pc, lr=0.1)
    for _ in range(30):
        for ps, o in ((pa, oa), (pb, ob)):
            o.zero_grad()
            loss_of(ps).backward()
            o.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), float((a - b).abs().max())
    # graph: 3 eager steps, then one captured step replayed 27 times
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            oc.zero_grad(set_to_none=True)
            loss_of(pc).backward()
            oc.step()
    torch.cuda.current_stream().wait_stream(side)
    oc.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_of(pc).backward()
        oc.step()
    for _ in range(27):
        g.replay()
    torch.cuda.synchronize()
    assert int(oc._state[0]) == 30
    for a, c in zip(pa, pc):
        assert torch.allclose(a, c, rtol=2e-5, atol=2e-6), float((a - c).abs().max())


@pytest.mark.parametrize("graph", [False, True])
def test_deferred_check_replays_a_failed_stretch(va, graph, monkeypatch):
    """The batched loops do not read the factorisation's `info` back every step (train_utils._run_iterations): a
    failure found at the next check sends the parameters and the optimiser state back to the last clean snapshot and
    that stretch is replayed with the per-step check.  Inject ONE spurious failure flag in the middle of a run: the
    trained parameters must equal an undisturbed run's (the replay itself sees no failure), in the eager-deferred loop
    and in the graph-captured one."""
    from volt_amd import ops, train_utils
    from volt_amd.train_utils import TrainVoltMagpieBatch
    B, n, iters = 3, 200, 70
    x, F, vol = sde_batch(B, n, seed=3)
    tx, prices, v = dev(x), dev(F[:, 1:]), dev(vol)
    ref_model, ref_lh, ref_losses = TrainVoltMagpieBatch(tx, prices, v, train_iters=iters, k=20, defer=False)
    real = ops.mll_step
    calls = {"n": 0, "injected": 0}

    def flaky(K, resid, sigma2, ws=None, want_grad=True, jitter=0.0, **kw):
        out, alpha, info = real(K, resid, sigma2, ws, want_grad=want_grad, jitter=jitter, **kw)
        calls["n"] += 1
        if calls["n"] == 33 and not graph:                  # one step of the deferred stretch reports a failed pivot
            info[1] = 7
            out[1, :2] = float("nan")                       # ... and hands garbage to the optimiser
            calls["injected"] += 1
        return out, alpha, info
    monkeypatch.setattr(ops, "mll_step", flaky)
    monkeypatch.setattr(train_utils, "CHECK_EVERY", 20)
    m, lh, losses = TrainVoltMagpieBatch(tx, prices, v, train_iters=iters, k=20, graph=graph)
    if not graph:
        assert calls["injected"] == 1 and calls["n"] > iters          # the stretch was replayed
    assert bool(torch.isfinite(lh.raw_noise).all())
    assert torch.allclose(lh.raw_noise.detach(), ref_lh.raw_noise.detach(), rtol=2e-4, atol=2e-4)
    assert torch.allclose(losses, ref_losses, rtol=1e-4, atol=1e-5)


def test_batched_model_has_a_vol_forecaster(va):
    """voltron/models/VoltronGP.py:46-50 gives a batched model a (botorch multitask) vol model; here it is the batched
    BMGP -- SamplePrediction / MeanPrediction work on batched models, and the forecast vol agrees with per-series models."""
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltMagpie, VoltronGP
    T, n, H = 3, 120, 5
    x, F, vol = sde_batch(T, n, seed=11)
    tx = dev(x)
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    mb = VoltronGP(tx, torch.log(dev(F[:, 1:])), GaussianLikelihood(batch_shape=torch.Size([T])).cuda(), dev(vol))
    assert mb.vol_model is not None and VoltMagpie(tx, torch.log(dev(F[:, 1:])), GaussianLikelihood(
        batch_shape=torch.Size([T])).cuda(), dev(vol)).vol_model is not None
    pred, pv = mb.MeanPrediction(test_x, return_vol=True)
    assert tuple(pv.shape) == (T, H) and tuple(pred.shape) == (T, H) and bool(torch.isfinite(pred).all())
    for t in range(T):
        m1 = VoltronGP(tx, torch.log(dev(F[t, 1:])), GaussianLikelihood().cuda(), dev(vol[t]))
        pv1 = m1.vol_model.eval()(test_x).mean.exp()
        assert torch.allclose(pv[t], pv1, rtol=2e-4, atol=1e-6)
    smp = mb.SamplePrediction(test_x)
    assert tuple(smp.shape) == (T, H) and bool(torch.isfinite(smp).all())


def test_batched_driver_isolates_a_failing_series(va, tmp_path, capsys):
    """One ticker whose volatility path is NaN must not take the window down: the batched pass fails, the window is redone
    series by series, the bad ticker gets NaN samples and a "Failed:" line (the reference's per-ticker try / except,
    experiments/stocks/GenerateMultiMeanPreds.py:185-198), the others finite forecasts."""
    from volt_amd.forecast import GenerateStockPredictionsBatch, realised_vol
    B, T, ntrain, H, S = 3, 66, 64, 4, 5
    x, F, vol = sde_batch(B, T - 1, seed=5)
    closes = dev(F)

    def vol_fn(train_x, train_y):
        v = realised_vol(train_x, train_y)
        v[train_y[:, 5] == closes[1, 5]] = float("nan")          # ticker BBB, in the batch of three and alone
        return v
    with pytest.warns(Warning):
        out = GenerateStockPredictionsBatch(["AAA", "BBB", "CCC"], closes[:, :-1], forecast_horizon=H, train_iters=3, nsample=S,
                                            ntrain=ntrain, mean="ewma", save=True, k=10, ntimes=1, vol_iters=2, vol_fn=vol_fn,
                                            par_dir=str(tmp_path))
    assert tuple(out.shape) == (B, S, H)
    assert bool(torch.isnan(out[1]).all()) and bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[2]).all())
    assert "Failed:  BBB" in capsys.readouterr().out
    saved = torch.load(next((tmp_path / "BBB").iterdir()))
    assert bool(torch.isnan(saved).all())
