"""Pin the CPU oracle to outputs of the reference's own code (tests/golden/*.npz, produced by
tests/golden/make_golden.py from /root/reference).  CPU-only."""
import numpy as np
import pytest

from oracle import volt_oracle as vo

FILL_TAGS = ["n7", "n64", "n257", "b3n50", "b2n130"]


@pytest.mark.parametrize("tag", FILL_TAGS)
def test_cumtrapz_bit_exact(golden, tag):
    g = golden("fill")
    vol, x = g[f"{tag}_vol"], g[f"{tag}_x"]
    V = vo.cumtrapz(vol * vol, x)
    assert V.dtype == np.float32
    assert np.array_equal(V, g[f"{tag}_V"])          # bit-exact (VolKernel.py:4-10)


@pytest.mark.parametrize("tag", FILL_TAGS)
def test_volatility_kernel_bit_exact(golden, tag):
    g = golden("fill")
    vol, x = g[f"{tag}_vol"], g[f"{tag}_x"]
    xx = x if vol.ndim == 1 else np.repeat(x[None], vol.shape[0], 0)
    K = vo.volatility_kernel(xx[..., None], vol[..., None])
    assert np.array_equal(K, g[f"{tag}_K"])          # VolKernel.py:18-42
    d = vo.volatility_kernel(xx[..., None], vol[..., None], diag=True)
    assert np.array_equal(d, g[f"{tag}_diag"])


def test_volatility_kernel_fp64_inherits_dtype(golden):
    g = golden("fill")
    K = vo.volatility_kernel(g["f64_x"][:, None], g["f64_vol"][:, None])
    assert K.dtype == np.float64
    np.testing.assert_allclose(K, g["f64_K"], rtol=1e-15, atol=0)


EW_TAGS = [("n60k5", 5), ("n60k25", 25), ("b3n40k7", 7), ("n300k100", 100)]


@pytest.mark.parametrize("tag,k", EW_TAGS)
def test_ewma_matches_reference(golden, tag, k):
    g = golden("ewma")
    y = g[f"{tag}_y"]
    out = vo.ewma(y, k)
    assert out.shape == g[f"{tag}_ewma"].shape       # length N+1 (EWMA.py:20-37)
    # conv1d summation order is unspecified: fp32 round-off only
    np.testing.assert_allclose(out, g[f"{tag}_ewma"], rtol=2e-6, atol=0)


@pytest.mark.parametrize("tag,k", EW_TAGS)
@pytest.mark.parametrize("cname", ["ewma", "dewma", "tewma", "meanrevert"])
def test_mean_classes_three_way_return(golden, tag, k, cname):
    g = golden("ewma")
    y, x = g[f"{tag}_y"], g[f"{tag}_x"]
    fn = {"ewma": vo.ewma_mean, "dewma": vo.dewma_mean, "tewma": vo.tewma_mean,
          "meanrevert": vo.meanrevert_mean}[cname]
    n = x.shape[0]
    for branch, xq in (("train", x), ("one", x[-1:] + np.float32(1 / 252.)), ("other", x[: n // 2])):
        ref = g[f"{tag}_{cname}_{branch}"]
        out = fn(xq, x, y, k)
        assert out.shape == ref.shape, (branch, out.shape, ref.shape)
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)


RO_TAGS = [("ewma", "ewma"), ("ewma_theta", "ewma"), ("dewma", "dewma"), ("tewma", "tewma"),
           ("ewma_n120", "ewma")]


@pytest.mark.parametrize("tag,mean", RO_TAGS)
def test_rollouts_match_reference_pathwise(golden, tag, mean):
    """Reference Rollouts (rollout_utils.py:57-93) with the same pred_vol and N(0,1) draws.
    fp32 dense Cholesky of the noise-free K (cond ~1e5..1e6 here) => compare at 2e-3 abs on
    log-prices ~2.3 (the reference's own LAPACK-vs-LAPACK reproducibility class)."""
    g = golden("rollouts")
    theta = float(g[f"{tag}_theta"])
    out = vo.rollouts(g[f"{tag}_train_x"], g[f"{tag}_train_y"], g[f"{tag}_test_x"],
                      np.log(g[f"{tag}_vol_path"]), g[f"{tag}_pred_vol"], g[f"{tag}_z"],
                      mean_name=mean, k=int(g[f"{tag}_k"]), theta=None if np.isnan(theta) else theta)
    ref = g[f"{tag}_samples"]
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-3)


def test_generate_prediction_multipoint(golden):
    g = golden("rollouts")
    tx, ty = g["gpm_train_x"], g["gpm_train_y"]
    c = np.float32(g["gpm_const"])
    out = vo.generate_prediction(tx, np.log(ty[1:]), np.log(g["gpm_vol_path"]), g["gpm_test_x"],
                                 g["gpm_pred_vol"], g["gpm_z"],
                                 lambda x: np.full((np.asarray(x).shape[0],), c, dtype=np.float32))
    assert out.shape == g["gpm_samples"].shape
    np.testing.assert_allclose(out, g["gpm_samples"], rtol=0, atol=2e-3)


# ----------------------------------------------------------------------------- (f)2: nonvol_rollouts
@pytest.mark.parametrize("tag,mean,kern", [("matern_ewma", "ewma", "matern"), ("rbf_dewma", "dewma", "rbf"),
                                           ("matern_tewma", "tewma", "matern")])
def test_nonvol_rollouts_match_reference_loop(golden, tag, mean, kern):
    """oracle.nonvol_rollouts vs the reference's own loop (rollout_utils.py:95-115) run with its EWMA mean classes and
    a dense fp64 predictive standing in for botorch's ``posterior`` (tests/golden/make_golden.py)."""
    d = golden("nonvol")
    ls, os_, noise, k = (float(d[f"{tag}_{n}"]) for n in ("ls", "os", "noise", "k"))
    kf = vo.matern_kernel if kern == "matern" else vo.rbf_kernel
    out = vo.nonvol_rollouts(d[f"{tag}_train_x"], d[f"{tag}_train_y"], d[f"{tag}_test_x"],
                             lambda a, b: kf(a, b, ls, os_), noise, d[f"{tag}_z"], mean_name=mean, k=int(k))
    assert np.abs(out - d[f"{tag}_samples"]).max() < 5e-5


def test_fp64_factor_and_solve_fixture(golden):
    """tests/golden/chol64.npz (reference's fp64 VolatilityKernel.forward + the ATen calls of rollout_utils.py:35-36):
    the oracle's fp64 kernel matrix factors and solves to the reference's own numbers."""
    g = golden("chol64")
    n = g["x"].shape[0]
    K = vo.volatility_kernel(g["x"], g["vol"])
    assert K.dtype == np.float64
    L, used = vo.psd_safe_cholesky(K, jitter=1e-4)
    assert used == 0.0
    Lref = np.zeros((n, n))
    Lref[np.tril_indices(n)] = g["L_packed"]
    np.testing.assert_allclose(L, Lref, rtol=0, atol=1e-10 * np.abs(Lref).max())
    sol = np.linalg.solve(L.T, np.linalg.solve(L, g["rhs"]))
    np.testing.assert_allclose(sol, g["sol"], rtol=0, atol=1e-8 * np.abs(g["sol"]).max())


@pytest.mark.parametrize("tag", ["n12d3", "n9d9", "n5d8", "n130d2"])
def test_last_dim_is_batch_matches_the_reference_code(golden, tag):
    """VolKernel.py:24-26,35-40 (``last_dim_is_batch``, with and without ``diag`` -- "TODO: check this" upstream, mirrored as
    written): fixtures from the reference's own forward (tests/golden/make_golden_ldb.py), bit-exact."""
    g = golden("fill_ldb")
    x, vol = g[f"{tag}_x"], g[f"{tag}_vol"]
    assert np.array_equal(vo.volatility_kernel_last_dim_is_batch(x, vol), g[f"{tag}_K"])
    assert np.array_equal(vo.volatility_kernel_last_dim_is_batch(x, vol, diag=True), g[f"{tag}_diag"])
