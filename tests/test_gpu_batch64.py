"""The fp64 one-launch factorisation / gradient step (csrc/batch64_step.hip, round 5): D(i) tiles that solve against the
diagonal block above them 32 columns at a time while it is still being factored, chased sums, progress words.
reference call sites: psd_safe_cholesky at voltron/rollout_utils.py:35 on the noise-free train block; a double-precision
model's step (voltron/train_utils.py:243-254 in the caller's dtype).

All through the C ABI (ops.potrf -> volt_potrf_ws_f64, ops.mll_step -> volt_mll_step_f64):
  * against fp64 LAPACK on the device (torch.linalg.cholesky + triangular solves): factor, mll, d mll / d sigma2, alpha;
  * bitwise repeatable (the one launch has no atomics, unlike the K-sliced launches it replaces);
  * both hand-off protocols -- batches that are a multiple of 8 (one XCD per matrix, no fences) and those that are not;
  * 2 and 3 block columns (no look-ahead piece / one), ragged N, a matrix that is not positive definite in the batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from volt_amd import ops as o
    return o


def _problem(ops, B, n, s2v=0.05):
    from volt_amd.synthetic import sde_batch
    x, F, vol = sde_batch(min(B, 4), n)
    rep = B // min(B, 4) + 1
    vol = np.tile(vol, (rep, 1))[:B] * (1.0 + 0.01 * np.arange(B)[:, None])
    F = np.tile(F, (rep, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda().double())
    r = (y - y.mean(-1, keepdim=True)).contiguous()
    s2 = torch.full((B,), s2v, device="cuda", dtype=torch.float64)
    return K, r, s2


def _lapack(K, r, s2):
    B, n, _ = K.shape
    Lr = torch.linalg.cholesky(K + torch.diag_embed(s2[:, None].expand(B, n)))
    st = torch.linalg.solve_triangular                 # (torch.cholesky_solve on a batch: launch failure on this image)
    al = st(Lr.mT, st(Lr, r[..., None], upper=False), upper=True)[..., 0]
    Li = st(Lr, torch.eye(n, device=K.device, dtype=torch.float64).expand(B, n, n), upper=False)
    q = (r * al).sum(-1)
    ld = 2 * torch.log(torch.diagonal(Lr, dim1=-2, dim2=-1)).sum(-1)
    mll = -0.5 * (q + ld + n * np.log(2 * np.pi)) / n
    dm = 0.5 * ((al ** 2).sum(-1) - (Li ** 2).sum((-2, -1))) / n
    return Lr, mll, dm, al


CASES = [(1, 4096), (8, 1024), (3, 2048), (2, 1000), (16, 512), (5, 300), (4, 256), (8, 2500), (24, 700)]


@pytest.mark.parametrize("B,n", CASES)
def test_one_launch_fp64_matches_lapack_and_repeats(ops, B, n):
    from volt_amd import _lib
    L = _lib.lib()
    Np = ops.padded_n(n)
    assert L.volt_potrf_workspace_bytes_f64(B, Np) > 0, "the shape must run as one launch"
    K, r, s2 = _problem(ops, B, n)
    f = ops.potrf(K, s2)
    assert int(f.info.abs().sum()) == 0
    ws = ops.MllWorkspace(B, n, True, K.device, torch.float64)
    o, a, info = ops.mll_step(K, r, s2, ws)
    assert int(info.abs().sum()) == 0
    Lr, mll, dm, al = _lapack(K, r, s2)
    tol = 1e-10 if n <= 1024 else 1e-9                  # (the tolerances the fp64 path has held since round 2)
    assert float((f.L - Lr).abs().amax() / Lr.abs().amax()) < tol
    assert float(((o[:, 0] - mll).abs() / mll.abs()).max()) < tol
    assert float(((o[:, 1] - dm).abs() / dm.abs()).max()) < 100 * tol
    assert float((a - al).abs().amax() / al.abs().amax()) < 100 * tol
    A0, o0, a0 = f.A.clone(), o.clone(), a.clone()
    for _ in range(3):
        f2 = ops.potrf(K, s2)
        o2, a2, _i = ops.mll_step(K, r, s2, ws)
        assert torch.equal(torch.tril(f2.A), torch.tril(A0)) and torch.equal(o2, o0) and torch.equal(a2, a0)


def test_one_launch_fp64_reports_a_matrix_that_is_not_pd(ops):
    B, n = 4, 1024
    K, r, s2 = _problem(ops, B, n)
    K = K.clone()
    K[2, 700, 700] = -5.0                               # block column 5, pivot 60 of it
    f = ops.potrf(K, s2)
    info = f.info.cpu().tolist()
    assert info[0] == 0 and info[1] == 0 and info[3] == 0
    assert info[2] == 701                               # 1-based index of the first bad pivot (potrf's convention)
    Lr = torch.linalg.cholesky(K[[0, 1, 3]] + torch.diag_embed(s2[:3, None].expand(3, n)))
    assert float((f.L[[0, 1, 3]] - Lr).abs().amax() / Lr.abs().amax()) < 1e-10


@pytest.mark.parametrize("kind", ["negative", "zero", "nan"])
@pytest.mark.parametrize("at", [0, 31, 32, 63, 127, 128, 300])
def test_fp64_first_bad_pivot_is_lapacks_at_every_lane_of_the_pivot_wave(ops, kind, at):
    """Round 6: the pivot phase no longer tests d > 0 pivot by pivot -- lane j keeps 1/sqrt(d_j) and the first lane whose value is
    not > 0 names the pivot (csrc/tiles64.h pivot_phase64).  A pivot that is <= 0 or NaN must be reported at LAPACK's index
    (torch.linalg.cholesky_ex) wherever it sits in a 32-pivot phase, with later pivots' NaNs not taking its place; both the
    one launch and the launch-per-column path."""
    from volt_amd import _lib
    L = _lib.lib()
    B, n = 2, 384
    K, r, s2 = _problem(ops, B, n)
    K = K.clone()
    M = K[1] + torch.diag(s2[1].expand(n))
    if kind == "nan":
        M[at, at] = float("nan")
        want = at + 1
    elif kind == "zero":
        if at != 0: pytest.skip("an exactly zero pivot can only be placed at index 0 (elsewhere its sign is a matter of summation order)")
        M[0, 0] = 0.0
        want = 1
    else:
        Lr = torch.linalg.cholesky(M)
        M[at, at] -= 1.5 * Lr[at, at] ** 2                        # the Schur complement's pivot `at` becomes -0.5 d
        want = int(torch.linalg.cholesky_ex(M).info)
        assert want == at + 1
    K[1] = M - torch.diag(s2[1].expand(n))
    f = ops.potrf(K, s2)
    info = f.info.cpu().tolist()
    assert info[0] == 0 and info[1] == want, (info, want)
    A = torch.empty_like(f.A)
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), B, n, _lib.stream_ptr()), "prep")
    W, info2 = torch.empty_like(f.Winv), torch.empty_like(f.info)
    _lib.check(L.volt_potrf_ws_f64(A.data_ptr(), W.data_ptr(), info2.data_ptr(), B, n, None, 0, _lib.stream_ptr()), "potrf")
    assert info2.cpu().tolist() == [0, want]


def test_null_workspace_is_the_launch_per_column_path(ops):
    """ws == NULL is volt_potrf_f64: same factor to fp64 round-off (the K-sliced path sums with atomics)."""
    from volt_amd import _lib
    L = _lib.lib()
    B, n = 2, 1024
    K, r, s2 = _problem(ops, B, n)
    f = ops.potrf(K, s2)
    A = torch.empty_like(f.A)
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), B, n, _lib.stream_ptr()), "prep")
    W, info = torch.empty_like(f.Winv), torch.empty_like(f.info)
    _lib.check(L.volt_potrf_ws_f64(A.data_ptr(), W.data_ptr(), info.data_ptr(), B, n, None, 0, _lib.stream_ptr()), "potrf")
    assert int(info.abs().sum()) == 0
    assert float((torch.tril(A) - torch.tril(f.A)).abs().amax()) < 1e-12 * float(f.A.abs().amax())
    assert float((W - f.Winv).abs().amax()) < 1e-9 * float(f.Winv.abs().amax())


@pytest.mark.parametrize("B,n", [(2, 1000), (8, 1024), (1, 2048)])
def test_factor_straight_from_k_equals_the_prepared_copy(ops, B, n):
    """volt_potrf_k_f64 reads its tiles from K (sigma2 + jitter added on the way, identity in the padding); volt_prepare_f64 +
    volt_potrf_ws_f64 factor a prepared copy: the same arithmetic on the same numbers -- bitwise the same factor."""
    from volt_amd import _lib
    L = _lib.lib()
    K, r, s2 = _problem(ops, B, n)
    f = ops.potrf(K, s2, jitter=1e-7)                    # -> volt_potrf_k_f64
    A = torch.empty_like(f.A)
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 1e-7, A.data_ptr(), B, n, _lib.stream_ptr()), "prep")
    W, info = torch.empty_like(f.Winv), torch.empty_like(f.info)
    ops.potrf_f64_inplace(A, W, info)
    assert int(info.abs().sum()) == 0 and int(f.info.abs().sum()) == 0
    assert torch.equal(torch.tril(A), torch.tril(f.A)) and torch.equal(W, f.Winv)


@pytest.mark.parametrize("B,n", [(1, 640), (8, 1024), (3, 1500), (16, 512), (2, 4096)])
def test_trtri_f64_one_launch_vs_lapack_and_launch_per_row(B, n):
    """volt_trtri_ws_f64 runs the whole inverse as ONE launch (the TD / T pieces of the one-launch step with the
    factor complete: csrc/batch64_step.hip) -- against torch's fp64 triangular inverse (1e-11 of max |Y|), bitwise against
    volt_trtri_f64's launch-per-row kernels where the tiles' arithmetic is the same (it is: one gemm64 chain per tile, then the
    product with W_i), and bitwise repeatable.  B = 8 / 16 take the fence-free LOCAL hand-offs."""
    import torch
    from volt_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(n)
    M = torch.randn(B, n, n, device="cuda", dtype=torch.float64, generator=g)
    K = M @ M.mT / n + torch.eye(n, device="cuda", dtype=torch.float64)
    f = ops.potrf(K)
    assert int(f.info.abs().sum()) == 0
    Np = f.A.shape[1]
    nb = int(L.volt_trtri_workspace_bytes_f64(B, Np))
    assert nb > 0
    ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    wp = (ws.data_ptr() + 255) // 256 * 256
    Y = torch.full((B, Np, Np), float("nan"), device="cuda", dtype=torch.float64)
    _lib.check(L.volt_trtri_ws_f64(f.A.data_ptr(), f.Winv.data_ptr(), Y.data_ptr(), B, Np, wp, nb, _lib.stream_ptr()), "trtri_ws")
    Y1 = Y[:, :n, :n].triu().clone()
    ref = torch.linalg.inv(torch.linalg.cholesky(K.cpu())).mT.triu()          # L^-T
    assert float((Y1.cpu() - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
    Y2 = torch.empty_like(Y)
    _lib.check(L.volt_trtri_f64(f.A.data_ptr(), f.Winv.data_ptr(), Y2.data_ptr(), B, Np, _lib.stream_ptr()), "trtri")
    assert float((Y2[:, :n, :n].triu() - Y1).abs().max()) <= 1e-13 * float(ref.abs().max())
    for _ in range(3):
        Y.fill_(float("nan"))
        _lib.check(L.volt_trtri_ws_f64(f.A.data_ptr(), f.Winv.data_ptr(), Y.data_ptr(), B, Np, wp, nb, _lib.stream_ptr()), "trtri_ws")
        assert torch.equal(Y[:, :n, :n].triu(), Y1)
    assert torch.equal(ops.trtri(f), Y1)                                      # the Python op takes the one launch too
