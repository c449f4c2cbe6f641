"""Edge cases of the C ABI and the host mirror on the MI355X: tiny / ragged / empty inputs, strided
views, jitter policy, model state handling, statistical behaviour of rollouts."""
import copy
import warnings

import numpy as np
import pytest
import torch

from oracle import volt_oracle as vo
from volt_amd.synthetic import rollout_inputs, sde_batch, sde_series

pytestmark = pytest.mark.gpu
SIG2 = float(vo.noise_from_raw(1e-5))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from volt_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(a).cuda()


@pytest.mark.parametrize("n", [2, 3, 5, 127, 128, 129])
def test_small_and_boundary_sizes(ops, n):
    x = (np.arange(n) / 252.0).astype(np.float32)
    vol = np.random.RandomState(n).uniform(0.1, 0.4, (2, n)).astype(np.float32)
    V = ops.cumtrapz(dev(vol), dev(x), square=True)
    assert np.array_equal(V.cpu().numpy(), vo.cumtrapz(vol * vol, x))
    K = ops.fill(V)
    Ko = vo.volatility_kernel(np.repeat(x[None], 2, 0)[..., None], vol[..., None])
    assert np.array_equal(K.cpu().numpy(), Ko)
    r = np.random.RandomState(1).randn(2, n).astype(np.float32) * 0.01
    out, alpha, info = ops.mll_step(K, dev(r), torch.full((2,), SIG2, device="cuda"))
    o = vo.mll_and_grads(Ko, r, np.zeros_like(r), 1e-5)
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), o["mll"], rtol=2e-5)
    np.testing.assert_allclose(out[:, 1].cpu().numpy(), 0.5 * (o["aa"] - o["trinv"]) / n, rtol=1e-3)


def test_empty_batch_is_a_no_op(ops):
    from volt_amd import _lib
    L = _lib.lib()
    assert L.volt_fill_f32(1, 1, 0, 8, 8, 64, None) == 0
    assert L.volt_potrf_f32(1, 1, 1, 0, 128, None) == 0
    assert L.volt_cumtrapz_f32(1, 8, 1, 0, 1, 0, 8, 1, None) == 0


def test_cumtrapz_needs_two_points(ops):
    from volt_amd._lib import VoltHipError
    with pytest.raises(VoltHipError):
        ops.cumtrapz(torch.ones(1, device="cuda"), torch.zeros(1, device="cuda"))


def test_potrf_on_strided_view_and_jitter(ops):
    """rollout_utils.py:27-35 factors cov_mat[..., :cut, :cut], a strided view; psd_safe_cholesky adds
    jitter only when the plain factorisation fails."""
    from volt_amd.gp import NotPSDError, NumericalWarning, psd_safe_cholesky
    n, cut = 200, 173
    x, F, vol = sde_batch(2, n)
    K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True)) + SIG2 * torch.eye(n, device="cuda")
    view = K[:, :cut, :cut]
    assert not view.is_contiguous()
    L = psd_safe_cholesky(view)
    ref = np.linalg.cholesky(K.cpu().numpy().astype(np.float64)[:, :cut, :cut])
    assert np.abs(L.cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    rank1 = torch.ones(1, 130, 130, device="cuda")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Lj = psd_safe_cholesky(rank1, jitter=1e-3)
    assert any(issubclass(x.category, NumericalWarning) for x in w)
    rec = (Lj @ Lj.transpose(-1, -2))[0].cpu().numpy()
    np.testing.assert_allclose(rec, np.ones((130, 130)) + 1e-3 * np.eye(130), atol=2e-3)
    with pytest.raises(NotPSDError):
        psd_safe_cholesky(-torch.eye(140, device="cuda").unsqueeze(0))


def test_update_vol_path_and_forward_cache(ops):
    """train_cov is built once (VoltMagpie.py:46), reused when x is train_x (:123-124), rebuilt by
    UpdateVolPath (:57-60)."""
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltMagpie, VoltronGP
    n = 150
    F, vol = sde_series(n, 3)
    tx = torch.arange(n, device="cuda") / 252.
    for cls in (VoltMagpie, VoltronGP):
        m = cls(tx, dev(F)[1:].log(), GaussianLikelihood().cuda(), dev(vol))
        # the models keep log(vol) and fill from exp(log(vol)) like the reference (VoltMagpie.py:44-46);
        # that elementwise round trip is torch's, so the oracle is fed the same round-tripped path
        K0 = vo.volatility_kernel((np.arange(n) / 252.).astype(np.float32), m.log_vol_path.exp().cpu().numpy())
        assert np.array_equal(m.train_cov.cpu().numpy(), K0)
        m.train()
        out = m(tx)
        assert out.covariance_matrix.data_ptr() == m.train_cov.data_ptr()          # cached, not refilled
        vol2 = (vol * 1.5).astype(np.float32)
        m.UpdateVolPath(dev(vol2))
        K1 = vo.volatility_kernel((np.arange(n) / 252.).astype(np.float32), m.log_vol_path.exp().cpu().numpy())
        assert np.array_equal(m(tx).covariance_matrix.cpu().numpy(), K1)
        other = m.forward((tx + 1.0).unsqueeze(-1))                                 # not train_x -> refill branch
        assert other.covariance_matrix.shape == (n, n)


def test_default_vol_path_when_none(ops):
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltMagpie
    n = 64
    tx = torch.arange(n, device="cuda") / 252.
    m = VoltMagpie(tx, torch.zeros(n, device="cuda"), GaussianLikelihood().cuda())
    assert torch.allclose(m.log_vol_path, -torch.ones(n, device="cuda"))           # VoltMagpie.py:41-42


def test_volt_class_train_and_forecast(ops):
    from volt_amd.models.Volt import Volt
    n, H, S = 130, 6, 8
    F, vol = sde_series(n, 5)
    tx = torch.arange(n + 1, device="cuda") / 252.
    m = Volt(tx, dev(F).log(), mean="ewma", vol_path=dev(vol), k=10)
    m.Train(data_mod_iters=4, vol=dev(vol))                               # data-model stage only
    assert torch.isfinite(m.likelihood.raw_noise).all()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.Train(gpcv_iters=5, vol_mod_iters=5, data_mod_iters=4)          # Volt.py:95-146: GPCV -> vol model -> data model
    assert m.log_vol_path.shape == (n,) and torch.isfinite(m.log_vol_path).all()
    assert torch.isfinite(m.likelihood.raw_noise).all()
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + 1 / 252.
    pv, z = rollout_inputs(vol[-1], S, H, seed=1)
    out = m.Forecast(test_x, nsample=S, pred_vol=dev(pv), z=dev(z))
    assert tuple(out.shape) == (S, H) and torch.isfinite(out).all()


def test_bmgp_posterior_sampling_feeds_rollouts(ops):
    """Rollouts draws pred_vol from model.vol_model(test_x).sample((S,)) (rollout_utils.py:66)."""
    from volt_amd.gp import GaussianLikelihood
    from volt_amd.models import VoltMagpie
    from volt_amd.rollout_utils import Rollouts
    n, H, S = 120, 5, 6
    F, vol = sde_series(n, 9)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    m = VoltMagpie(tx, dev(F)[1:].log(), GaussianLikelihood().cuda(), dev(vol), k=10)
    m.vol_model.eval()
    post = m.vol_model(test_x)
    # exact BM-GP posterior in fp64
    xt = (np.arange(n) / 252.)
    xs = test_x.cpu().numpy().astype(np.float64)
    v = float(m.vol_model.covar_module.vol)
    noise = float(m.vol_lh.noise)
    Ktt = v * np.minimum.outer(xt, xt) + noise * np.eye(n)
    Kst = v * np.minimum.outer(xs, xt)
    y = np.log(vol).astype(np.float64)
    mean = -0.5 * v * v * xs + Kst @ np.linalg.solve(Ktt, y + 0.5 * v * v * xt)
    cov = v * np.minimum.outer(xs, xs) - Kst @ np.linalg.solve(Ktt, Kst.T)
    np.testing.assert_allclose(post.mean.cpu().numpy(), mean, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(post.covariance_matrix.cpu().numpy(), cov, rtol=5e-2, atol=2e-4)
    torch.manual_seed(0)
    out = Rollouts(tx, dev(F), test_x, m, nsample=S)
    assert tuple(out.shape) == (S, H) and torch.isfinite(out).all()


def test_rollout_statistics_with_independent_draws(ops):
    """Posterior-sample statistics (BASELINE north_star): with fresh N(0,1) draws the first-step sample
    mean / sd over S paths match the exact conditional within Monte-Carlo error."""
    from volt_amd import rollout_engine as re_
    n, S, H, k = 399, 4096, 8, 25
    F, vol = sde_series(n, 21)
    g = torch.Generator(device="cuda").manual_seed(123)
    pv = torch.full((1, S, H), float(vol[-1]), device="cuda")
    z = torch.randn(1, S, H, device="cuda", generator=g)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    logy = torch.log(dev(F)[1:])
    samples, info = re_.rollout_series(tx, logy[None], torch.log(dev(vol))[None], test_x, pv, z, 0, k)
    assert int((info != 0).sum()) == 0
    s0 = samples[0, :, 0].double().cpu().numpy()
    ly = np.log(F[1:]).astype(np.float32)
    ema = vo.ewma(ly, k)
    mu = (ly[-1] - ema[-2]) + ema[-1]
    sd = np.sqrt(0.5 / 252. * float(vol[-1]) ** 2)
    assert abs(s0.mean() - mu) < 5 * sd / np.sqrt(S)
    assert abs(s0.std() - sd) < 5 * sd / np.sqrt(2 * S)
    # variance of the h-step-ahead increment grows like the sum of one-step variances (random walk in the residual)
    inc = (samples[0, :, H - 1] - samples[0, :, 0]).double().cpu().numpy()
    assert 0.5 * (H - 1) * sd ** 2 < inc.var() < 3.0 * (H - 1) * sd ** 2


@pytest.mark.parametrize("H,mode,theta", [(1, 0, None), (2, 1, None), (50, 2, None), (256, 0, 0.3), (257, 0, None),
                                          (300, 1, None), (512, 3, None), (700, 0, None), (1024, 0, 0.1)])
def test_rollout_append_only_is_bitwise_the_full_resubstitution(ops, H, mode, theta):
    """The default engine extends w_s by one entry per step; `resubstitute=True` re-solves every sample's triangular
    system from its stored rows at every step (what the engine did in round 2, H^3/6 * 4 B per path).  Rounds 3 - 5 held the two
    to BITWISE equality (same arithmetic in the same order); since round 6 the default engine carries w_s'w_s and w_s'z_s as
    fp64 running sums of the ONE new entry per step (one lane per path, csrc/rollout.hip) where the re-substitution still
    sums all of its recomputed entries in fp32 -- the same numbers to rounding: the paths agree to 5e-5 of the paths' magnitude
    over up to 1024 dependent steps (measured: <= 2.7e-5; the synthetic vol paths of this test make some of them wander to
    |log price| ~ 400), for every mean family, every chunking of the rows (H <= 256 / 512 / 1024) and with mean reversion;
    the same pivots fail (info)."""
    from volt_amd import rollout_engine as re_
    n, S, k = 200, 9, 25
    F, vol = sde_series(n, 4)
    pv, z = rollout_inputs(vol[-1], S, H, seed=H)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    logy = torch.log(dev(F)[1:])[None]
    kw = dict(latent_mean=logy.mean(), theta=theta) if theta is not None else {}
    args = (tx, logy, torch.log(dev(vol))[None], test_x, dev(pv)[None], dev(z)[None], mode, k)
    a, ia = re_.rollout_series(*args, **kw)
    b, ib = re_.rollout_series(*args, resubstitute=True, **kw)
    assert bool(torch.isfinite(a).all()) and torch.equal(ia, ib)
    assert float((a - b).abs().max()) <= 5e-5 * max(1.0, float(a.abs().max())), (float((a - b).abs().max()), float(a.abs().max()))
    a2, _ = re_.rollout_series(*args, **kw)
    assert torch.equal(a, a2)                                # and the engine itself is deterministic


def test_rollout_lane_per_path_matches_wave_per_path(ops, tmp_path):
    """Round 6: the product's rollout engine gives every LANE a path (csrc/rollout.hip, rollout_lane_kernel); the engine of
    rounds 3 - 5 -- a wave per path -- stays for windows too long for the LDS ring and is what a process started with
    VOLT_TUNE=1 VOLT_ROLLOUT_LANE=0 runs.  Same recursion, same operations; only the k tap products of the mean are added in
    another order (both in fp64): every mean mode, H not a multiple of 4 (scalar loads), S not a multiple of 64, theta -- the
    paths agree to 1e-5 of their magnitude and the same pivots fail."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, "tests")
from volt_amd import rollout_engine as re_
from volt_amd.synthetic import rollout_inputs, sde_series
out = {}
for H, mode, theta, S, k in ((256, 0, None, 130, 25), (255, 1, None, 70, 25), (64, 2, None, 9, 40), (300, 3, 0.3, 65, 25),
                            (37, 0, 0.1, 5, 300), (1024, 0, None, 3, 25)):
    n = 200
    F, vol = sde_series(n, 4)
    pv, z = rollout_inputs(vol[-1], S, H, seed=H)
    d = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    logy = torch.log(d(F)[1:])[None]
    kw = dict(latent_mean=logy.mean(), theta=theta) if theta is not None else {}
    a, ia = re_.rollout_series(tx, logy, torch.log(d(vol))[None], test_x, d(pv)[None], d(z)[None], mode, k, **kw)
    out[f"{H}_{mode}_{S}_{k}"] = (a.cpu().numpy().tolist(), ia.cpu().numpy().tolist())
json.dump(out, open(sys.argv[1], "w"))
"""
    res = {}
    for tag, env in (("lane", {}), ("wave", {"VOLT_TUNE": "1", "VOLT_ROLLOUT_LANE": "0"})):
        e = dict(os.environ)
        e.pop("VOLT_TUNE", None)
        e.update(env)
        f = str(tmp_path / f"{tag}.json")
        r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        res[tag] = json.load(open(f))
    for key in res["lane"]:
        a, ia = res["lane"][key]
        b, ib = res["wave"][key]
        assert ia == ib, key
        dev_ = float(np.abs(np.asarray(a) - np.asarray(b)).max())
        assert np.isfinite(np.asarray(a)).all() and dev_ <= 1e-5 * max(1.0, float(np.abs(np.asarray(a)).max())), (key, dev_)


@pytest.mark.parametrize("S,H,k", [(1, 1, 3), (5, 2, 300), (7, 256, 25), (3, 600, 25), (2, 1024, 40)])
def test_rollout_engine_shapes_and_extremes(ops, S, H, k):
    """S not a multiple of the 4 paths per workgroup, H = 1, H = 256 (one chunk per row), H > 256 (two and four chunks,
    up to the maximum 1024), k larger than N."""
    from volt_amd import rollout_engine as re_
    n = 140
    F, vol = sde_series(n, 2)
    pv, z = rollout_inputs(vol[-1], S, H, seed=8)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    samples, info = re_.rollout_series(tx, torch.log(dev(F)[1:])[None], torch.log(dev(vol))[None], test_x,
                                       dev(pv)[None], dev(z)[None], 0, k)
    out = samples[0].cpu().numpy()
    ly = np.log(F[1:]).astype(np.float32)
    for s in range(min(S, 2)):
        ys = ly.copy()
        for i in range(H):
            full = vo.ewma(ys, k)
            ref = (ys[-1] - full[-2]) + full[-1] + np.sqrt(0.5 / 252. * float(pv[s, i]) ** 2) * z[s, i]
            assert abs(ref - out[s, i]) < 2e-4, (s, i)
            ys = np.append(ys, np.float32(out[s, i]))


def test_long_series_cumtrapz_needs_large_lds(ops):
    """N = 30000 fp32 = 120 KB of dynamic LDS (the per-series prefix sum stays in one workgroup)."""
    n = 30000
    x = (np.arange(n) / 252.0).astype(np.float32)
    vol = np.random.RandomState(0).uniform(0.1, 0.4, (2, n)).astype(np.float32)
    V = ops.cumtrapz(dev(vol), dev(x), square=True)
    assert np.array_equal(V.cpu().numpy(), vo.cumtrapz(vol * vol, x))


@pytest.mark.parametrize("seed", [0, 1])
def test_randomised_sizes_batches_and_noise_against_oracle(ops, seed):
    """Odd sizes (ragged padding inside the last 128-block and inside its 32-sub-blocks), batches that do and do not
    divide into stream groups / XCDs, noise from 2.5e-3 to 2: fused MLL step vs the fp64 oracle.
    (scripts/fuzz_mll.py is the long form: 140 cases, worst MLL 3e-7, d/dsigma2 2e-6, alpha 1e-5.)"""
    rng = np.random.RandomState(seed)
    for _ in range(14):
        n = int(rng.choice([2, 3, 5, 31, 33, 64, 100, 127, 129, 160, 255, 257, 300, 383, 385, 500]))
        B = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 32, 33]))
        if n >= 300:
            B = min(B, 9)
        x, F, vol = sde_batch(B, n, seed=int(rng.randint(1, 10000)))
        raw = rng.uniform(-6, 2, size=B)
        K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
        y = np.log(F[:, 1:])
        mean = y.mean(-1, keepdims=True) + 0 * y
        s2 = torch.tensor([vo.noise_from_raw(r) for r in raw], dtype=torch.float32).cuda()
        o, a, info = ops.mll_step(K, torch.tensor(y - mean).float().cuda(), s2, want_grad=True)
        assert int(info.abs().sum()) == 0, (n, B)
        o, a = o.cpu().double().numpy(), a.cpu().double().numpy()
        ref = vo.mll_and_grads(K.cpu().double().numpy(), y, mean, raw)
        dsig = 0.5 * (ref["aa"] - ref["trinv"]) / n
        assert np.all(np.abs(o[:, 0] - ref["mll"]) <= 2e-5 * np.maximum(1.0, np.abs(ref["mll"]))), (n, B)
        assert np.all(np.abs(o[:, 1] - dsig) <= 1e-3 * np.maximum(np.abs(dsig), 1e-4 * ref["trinv"] / n)), (n, B)
        assert np.all(np.abs(a - ref["alpha"]).max(-1) <= 1e-4 * np.abs(ref["alpha"]).max(-1)), (n, B)
