"""Parity of the HIP path (through the C ABI) against the CPU oracle -- needs an MI355X.

Tolerances (fp32 HIP path vs fp64 oracle on the same fp32 inputs), stated per quantity:
  * cumtrapz / fill: bit-exact.
  * Cholesky factor of K + sigma^2 I (sigma^2 = softplus(1e-5)+1e-4 ~ 0.693, cond <= ~3e3):
    max |L - L64| <= 2e-5 * max|L64|.
  * MLL value: rel 2e-5; d mll / d sigma2: rel 1e-3 (difference of two O(1) traces); alpha: 1e-4 rel-to-max.
"""
import numpy as np
import pytest
import torch

from oracle import volt_oracle as vo
from volt_amd.synthetic import sde_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from volt_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(a).cuda()


# ------------------------------------------------------------------ a1 / a2
@pytest.mark.parametrize("tag", ["n7", "n64", "n257", "b3n50", "b2n130"])
def test_cumtrapz_and_fill_golden_bit_exact(ops, golden, tag):
    g = golden("fill")
    vol, x = g[f"{tag}_vol"], g[f"{tag}_x"]
    V = ops.cumtrapz(dev(vol), dev(x), square=True)
    assert np.array_equal(V.cpu().numpy(), g[f"{tag}_V"])
    K = ops.fill(V)
    assert np.array_equal(K.cpu().numpy(), g[f"{tag}_K"])


def test_cumtrapz_fill_fp64_golden(ops, golden):
    g = golden("fill")
    V = ops.cumtrapz(dev(g["f64_vol"]), dev(g["f64_x"]), square=True)
    K = ops.fill(V)
    assert K.dtype == torch.float64
    np.testing.assert_allclose(K.cpu().numpy(), g["f64_K"], rtol=1e-15, atol=0)


@pytest.mark.parametrize("B,n", [(4, 2048), (2, 4096), (3, 1001)])
def test_cumtrapz_fill_large_vs_oracle(ops, B, n):
    x, _, vol = sde_batch(B, n)
    V = ops.cumtrapz(dev(vol), dev(x), square=True)
    Vo = vo.cumtrapz(vol * vol, x)
    assert np.array_equal(V.cpu().numpy(), Vo)
    K = ops.fill(V)
    # size-independent property: K[i,j] == V[min(i,j)], symmetric, checked on the device
    idx = torch.arange(n, device="cuda")
    mn = torch.minimum(idx[:, None], idx[None, :])
    assert torch.equal(K, V[:, mn])
    assert torch.equal(K, K.transpose(-1, -2))


# ------------------------------------------------------------------ a5 / a6 building blocks
def _problem(B, n, seed=2019):
    x, F, vol = sde_batch(B, n, seed)
    V = vo.cumtrapz(vol * vol, x)
    idx = np.minimum.outer(np.arange(n), np.arange(n))
    K = V[:, idx]
    y = np.log(F[:, 1:])
    mean = np.stack([vo.ewma_mean(x, x, y[b], 25) for b in range(B)])
    return K.astype(np.float32), y, mean


SIG2 = float(vo.noise_from_raw(1e-5))


@pytest.mark.parametrize("B,n", [(1, 128), (3, 256), (2, 300), (8, 512), (1, 399)])
def test_potrf_vs_fp64(ops, B, n):
    K, _, _ = _problem(B, n)
    f = ops.potrf(dev(K), torch.full((B,), SIG2, device="cuda"))
    assert int(f.info.abs().sum()) == 0
    L = f.L.cpu().numpy().astype(np.float64)
    for b in range(B):
        L64 = np.linalg.cholesky(K[b].astype(np.float64) + SIG2 * np.eye(n))
        assert np.abs(L[b] - L64).max() <= 2e-5 * np.abs(L64).max()


def test_potrf_analytic_known_answer(ops):
    """Raw K (no noise), as the rollouts factor it: chol(K)[i,j] = sqrt(d_j)."""
    n = 256
    x = (np.arange(n) / 252.0).astype(np.float32)
    vol = np.random.RandomState(3).uniform(0.1, 0.4, (1, n)).astype(np.float32)
    V = vo.cumtrapz(vol * vol, x)
    K = V[:, np.minimum.outer(np.arange(n), np.arange(n))]
    f = ops.potrf(dev(K))
    assert int(f.info[0]) == 0
    ref = vo.analytic_cholesky(V[0])
    np.testing.assert_allclose(f.L[0].cpu().numpy(), ref, atol=5e-4 * ref.max())


def test_potrf_info_reports_failure(ops):
    n = 256
    K, _, _ = _problem(2, n)
    K[1] -= 5.0 * np.eye(n, dtype=np.float32)          # indefinite
    f = ops.potrf(dev(K), torch.full((2,), SIG2, device="cuda"))
    info = f.info.cpu().numpy()
    assert info[0] == 0 and info[1] > 0
    Kn = K.copy()
    Kn[0, 130, 130] = np.nan
    f = ops.potrf(dev(Kn), torch.full((2,), SIG2, device="cuda"))
    assert int(f.info[0]) == 131


@pytest.mark.parametrize("B,n", [(2, 256), (3, 300), (2, 640)])
def test_trsv_and_cholesky_solve(ops, B, n):
    K, y, mean = _problem(B, n)
    r = (y - mean).astype(np.float32)
    f = ops.potrf(dev(K), torch.full((B,), SIG2, device="cuda"))
    z = ops.trsv(f, dev(r)).cpu().numpy()
    a = ops.cholesky_solve(f, dev(r)).cpu().numpy()
    for b in range(B):
        Ks = K[b].astype(np.float64) + SIG2 * np.eye(n)
        L64 = np.linalg.cholesky(Ks)
        z64 = np.linalg.solve(L64, r[b].astype(np.float64))
        a64 = np.linalg.solve(Ks, r[b].astype(np.float64))
        assert np.abs(z[b] - z64).max() <= 1e-4 * np.abs(z64).max()
        assert np.abs(a[b] - a64).max() <= 1e-4 * np.abs(a64).max()


@pytest.mark.parametrize("B,n", [(2, 256), (1, 300), (2, 512)])
def test_trtri(ops, B, n):
    K, _, _ = _problem(B, n)
    f = ops.potrf(dev(K), torch.full((B,), SIG2, device="cuda"))
    Y = ops.trtri(f).cpu().numpy()
    for b in range(B):
        L64 = np.linalg.cholesky(K[b].astype(np.float64) + SIG2 * np.eye(n))
        Y64 = np.linalg.inv(L64).T
        assert np.abs(Y[b] - Y64).max() <= 1e-4 * np.abs(Y64).max()


# ------------------------------------------------------------------ a5: the step
@pytest.mark.parametrize("B,n", [(1, 256), (4, 512), (2, 399), (64, 256), (2, 1024)])
@pytest.mark.parametrize("want_grad", [True, False])
def test_mll_step_vs_oracle(ops, B, n, want_grad):
    K, y, mean = _problem(B, n)
    r = (y - mean).astype(np.float32)
    o = vo.mll_and_grads(K, y.astype(np.float32), mean.astype(np.float32), 1e-5)
    out, alpha, info = ops.mll_step(dev(K), dev(r), torch.full((B,), SIG2, device="cuda"), want_grad=want_grad)
    out = out.cpu().numpy().astype(np.float64)
    assert int(info.abs().sum()) == 0
    np.testing.assert_allclose(out[:, 0], o["mll"], rtol=2e-5)
    np.testing.assert_allclose(out[:, 2], o["quad"], rtol=1e-4)
    np.testing.assert_allclose(out[:, 3], o["logdet"], rtol=2e-5, atol=1e-3)
    if want_grad:
        dsig = 0.5 * (o["aa"] - o["trinv"]) / n
        np.testing.assert_allclose(out[:, 4], o["trinv"], rtol=1e-4)
        np.testing.assert_allclose(out[:, 5], o["aa"], rtol=1e-4)
        np.testing.assert_allclose(out[:, 1], dsig, rtol=1e-3, atol=1e-6)
        a = alpha.cpu().numpy()
        assert np.abs(a - o["alpha"]).max() <= 1e-4 * np.abs(o["alpha"]).max()


def test_mll_step_full_size_properties(ops):
    """BASELINE metric size (N=4096), reduced batch: size-independent checks done on the device.
    L L^T v == (K + s2 I) v for random probes, Y^T (L^T... ) identities, and agreement of the
    forward-only and gradient paths."""
    B, n = 2, 4096
    x, F, vol = sde_batch(B, n)
    Kd = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
    s2 = torch.full((B,), SIG2, device="cuda")
    f = ops.potrf(Kd, s2)
    assert int(f.info.abs().sum()) == 0
    L = f.L.double()
    g = torch.Generator(device="cuda").manual_seed(0)
    v = torch.randn(B, n, 4, device="cuda", generator=g, dtype=torch.float64)
    lhs = L @ (L.transpose(-1, -2) @ v)
    rhs = Kd.double() @ v + SIG2 * v
    assert ((lhs - rhs).norm() / rhs.norm()).item() < 5e-6
    Y = ops.trtri(f).double()
    resid_inv = (Y.transpose(-1, -2) @ (L @ v) - v).norm() / v.norm()      # L^-1 L v = v
    assert resid_inv.item() < 1e-4
    y = torch.log(dev(F[:, 1:]))
    r = y - y.mean(-1, keepdim=True)
    o1, a1, i1 = ops.mll_step(Kd, r, s2, want_grad=True)
    o1 = o1.clone()
    o0, _, _ = ops.mll_step(Kd, r, s2, want_grad=False)
    assert torch.allclose(o1[:, 0], o0[:, 0], rtol=1e-5)
    assert torch.allclose(o1[:, 2], o0[:, 2], rtol=1e-4)
    # alpha = (K + s2 I)^-1 r  <=>  (K + s2 I) alpha = r
    back = Kd.double() @ a1.double().unsqueeze(-1) + SIG2 * a1.double().unsqueeze(-1)
    assert ((back.squeeze(-1) - r.double()).norm() / r.double().norm()).item() < 1e-3


@pytest.mark.parametrize("B,n", [(64, 2048), (64, 4096)])
def test_mll_step_baseline_batches_properties(ops, B, n):
    """BASELINE configs 3 (64 x 2048) and the metric's 64 x 4096 at FULL batch, through the 4-stream schedule the bench
    times: size-independent properties per series, checked on the device in fp64 --
    (K + s2 I) alpha = r;  quad = r'alpha;  the analytic log-det of this kernel's factor
    (K = C diag(d) C' => chol(K + 0 I)[i,i] = sqrt(d_i)) is covered at small N, here
    logdet(K + s2 I) must lie between N log s2 and N log(s2 + max K);  tr(K_s^-1) in (0, N/s2];
    every series of the batch is processed (no group of the 4-stream split dropped): outputs differ per series and
    a permutation of the batch permutes the outputs."""
    x, F, vol = sde_batch(B, n)
    Kd = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
    s2v = torch.linspace(0.3, 0.9, B, device="cuda")
    rows = [0, B // 2, B - 1]                                   # first / middle / last series: both stream groups
    s2v[rows] = float(vo.noise_from_raw(1e-5))                  # the noise the oracle derives from raw_noise = 1e-5
    y = torch.log(dev(F[:, 1:]))
    ymean = y.mean(-1, keepdim=True).expand_as(y)
    r = (y - ymean).float()
    o, a, info = ops.mll_step(Kd, r, s2v, want_grad=True)
    o, a = o.clone(), a.clone()
    assert int(info.abs().sum()) == 0 and bool(torch.isfinite(o).all())
    # ... and three rows of the FULL batch against the fp64 oracle (same tolerances as test_mll_step_vs_oracle)
    oo = vo.mll_and_grads(Kd[rows].cpu().numpy(), y[rows].cpu().numpy(), ymean[rows].cpu().numpy(), 1e-5)
    oh, ah = o[rows].cpu().numpy().astype(np.float64), a[rows].cpu().numpy()
    np.testing.assert_allclose(oh[:, 0], oo["mll"], rtol=2e-5)
    np.testing.assert_allclose(oh[:, 1], 0.5 * (oo["aa"] - oo["trinv"]) / n, rtol=1e-3)
    np.testing.assert_allclose(oh[:, 4], oo["trinv"], rtol=1e-4)
    assert np.abs(ah - oo["alpha"]).max() <= 1e-4 * np.abs(oo["alpha"]).max()
    rd, ad = r.double(), a.double()
    back = torch.empty_like(rd)
    for b0 in range(0, B, 8):                                   # fp64 K v in slices: 8 x N^2 doubles at a time
        sl = slice(b0, b0 + 8)
        back[sl] = (Kd[sl].double() @ ad[sl].unsqueeze(-1)).squeeze(-1) + s2v[sl].double().unsqueeze(-1) * ad[sl]
    rel = (back - rd).norm(dim=-1) / rd.norm(dim=-1)
    assert float(rel.max()) < 2e-3, float(rel.max())
    quad = (rd * ad).sum(-1)
    assert float(((o[:, 2].double() - quad).abs() / quad.abs()).max()) < 1e-4
    kmax = Kd.amax(dim=(-1, -2)).double()
    ld = o[:, 3].double()
    assert bool((ld > n * torch.log(s2v.double())).all()) and bool((ld < n * torch.log(s2v.double() + n * kmax)).all())
    tr = o[:, 4].double()
    assert bool((tr > 0).all()) and bool((tr <= n / s2v.double() * (1 + 1e-5)).all())
    assert torch.unique(o[:, 0]).numel() == B
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    o2, a2, _ = ops.mll_step(Kd[perm].contiguous(), r[perm].contiguous(), s2v[perm].contiguous(), want_grad=True)
    assert torch.allclose(o2[:, :6], o[perm][:, :6], rtol=1e-5, atol=1e-6)
    assert torch.allclose(a2, a[perm], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["n12d3", "n9d9", "n5d8", "n130d2"])
def test_last_dim_is_batch_golden_bit_exact(ops, golden, tag):
    """VolatilityKernel.forward(..., last_dim_is_batch=True) with and without diag (VolKernel.py:24-26,35-40) against the
    reference's own outputs (tests/golden/fill_ldb.npz): bit-exact."""
    from volt_amd.kernels import VolatilityKernel
    g = golden("fill_ldb")
    x, vol = dev(g[f"{tag}_x"]), dev(g[f"{tag}_vol"])
    kern = VolatilityKernel()
    assert np.array_equal(kern.forward(x, vol, last_dim_is_batch=True).cpu().numpy(), g[f"{tag}_K"])
    assert np.array_equal(kern.forward(x, vol, diag=True, last_dim_is_batch=True).cpu().numpy(), g[f"{tag}_diag"])


def test_ops_refuse_cpu_tensors(ops):
    from volt_amd._lib import VoltHipError
    with pytest.raises(VoltHipError):
        ops.fill(torch.zeros(4))
