"""The one-launch batched step (csrc/batch_step.hip, round 5): for batches of longer series the whole MLL + gradient step --
factorisation, triangular inverse, z = Y'r and alpha's partial sums -- is ONE launch whose workgroups run a host-ordered
piece list and hand their tiles on through progress words.  reference: one loop body per step, voltron/train_utils.py:243-254.

Checked here, all through the C ABI:
  * against the fp64 oracle (the tolerances of the launch-per-column path);
  * BITWISE against the launch-per-column path where that path sums the same way (an uninitialised workspace runs the
    table-free plain schedule for 8, 9 and >= 22 matrices): the tile bodies are shared, so every bit of mll, its gradient
    and alpha must agree;
  * bitwise repeatable over repeated steps on one workspace (a hand-off race shows up as run-to-run differences);
  * both hand-off protocols: batches that are a multiple of 8 (one XCD per matrix, no fences) and those that are not
    (agent-scope release / acquire, written-through tiles);
  * the factorisation alone (ops.potrf -> volt_potrf_k_f32) through the same launch."""
import numpy as np
import pytest
import torch

from test_gpu_contract import SIG2, _check_vs_oracle, _series_problem, dev, ops  # noqa: F401

pytestmark = pytest.mark.gpu


def _plain_reference(ops, K, r, s2, B, n):
    """The same entry point on a workspace NOT declared initialised: launch-per-column, table-free."""
    from volt_amd import _lib
    L = _lib.lib()
    raw = torch.zeros(L.volt_mll_workspace_bytes(B, n, 1) + 1024, dtype=torch.uint8, device="cuda")
    ptr = (raw.data_ptr() + 511) // 512 * 512 + 256
    out, alpha = torch.empty(B, 8, device="cuda"), torch.empty(B, n, device="cuda")
    info = torch.empty(B, dtype=torch.int32, device="cuda")
    _lib.check(L.volt_mll_step_f32(K.data_ptr(), n, n * n, r.data_ptr(), s2.data_ptr(), 0.0, out.data_ptr(), alpha.data_ptr(),
                                   info.data_ptr(), ptr, B, n, _lib.WANT_GRAD, _lib.stream_ptr()), "step")
    assert int(info.abs().sum()) == 0
    return out, alpha


# (B, n, bitwise?)  bitwise only where the uninitialised path is the plain launch-per-column schedule (no K-slices)
CASES = [(8, 2048, True), (8, 1500, True), (24, 1536, True), (32, 1024, True), (64, 2048, True), (8, 4096, True),
         (3, 3072, False), (12, 2100, False), (6, 2048, False), (2, 4000, False), (20, 1400, False)]


@pytest.mark.parametrize("B,n,bitwise", CASES)
def test_one_launch_step_matches_oracle_and_launch_per_column(ops, B, n, bitwise):
    from volt_amd import _lib
    L = _lib.lib()
    assert L.volt_batch_describe(B, ops.padded_n(n) // 128, 1, 0, None, 0) > 0
    x, vol, y, mean = _series_problem(B, n)
    K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
    r = dev(y - mean)
    s2 = torch.full((B,), SIG2, device="cuda")
    ws = ops.MllWorkspace(B, n, True, K.device)
    out1 = ops.mll_step(K, r, s2, ws)[0].clone()
    alpha1 = ws.alpha.clone()
    assert int(ws.info.abs().sum()) == 0
    rows = sorted({0, B // 2, B - 1})
    _check_vs_oracle(K[rows].cpu().numpy(), y, mean, 1e-5, out1.cpu().numpy(), alpha1.cpu().numpy(), rows)
    out2, alpha2 = _plain_reference(ops, K, r, s2, B, n)
    if bitwise:
        assert torch.equal(out1, out2) and torch.equal(alpha1, alpha2)
    else:
        assert torch.allclose(out1[:, :6], out2[:, :6], rtol=2e-5, atol=1e-6)
        assert (alpha1 - alpha2).abs().max() <= 1e-4 * alpha2.abs().max()
    for _ in range(10):                                              # the same bits every time
        o, a, info = ops.mll_step(K, r, s2, ws)
        assert int(info.abs().sum()) == 0 and torch.equal(o, out1) and torch.equal(a, alpha1)


def test_one_launch_step_is_what_the_shape_runs(ops):
    """The roofline hook reports ONE launch for the shapes the gate gives to the batched step (and the launch-per-column
    classes otherwise): what bench.py's roofline object is built from."""
    import ctypes
    from volt_amd import _lib
    L = _lib.lib()
    for B, n, one in ((8, 2048, True), (4, 1024, False)):
        x, vol, y, mean = _series_problem(B, n)
        K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
        r = dev(y - mean)
        s2 = torch.full((B,), SIG2, device="cuda")
        ws = ops.MllWorkspace(B, n, True, K.device)
        ref = ops.mll_step(K, r, s2, ws)[0].clone()
        ms_sum, ms_un, cnt = (ctypes.c_float * 2)(), (ctypes.c_float * 2)(), (ctypes.c_int * 2)()
        inf = torch.empty(B, dtype=torch.int32, device="cuda")
        _lib.check(L.volt_profile_step_f32(K.data_ptr(), n, n * n, r.data_ptr(), s2.data_ptr(), ws.out.data_ptr(),
                                           ws.alpha.data_ptr(), ws.ptr, inf.data_ptr(), B, n, 0, _lib.stream_ptr(), ms_sum, ms_un,
                                           cnt, None), "profile")
        assert (list(cnt) == [1, 0]) == one, (B, n, list(cnt))
        assert int(inf.abs().sum()) == 0 and ms_un[0] > 0
        if one:
            assert torch.equal(ws.out, ref)                          # the profiled launch IS the step's
        else:                                                        # (short series: the hook times the launch-per-column path)
            assert torch.allclose(ws.out[:, :6], ref[:, :6], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("B,n", [(8, 2048), (16, 4096), (5, 3000), (24, 1536)])
def test_factorisation_alone_in_one_launch(ops, B, n):
    """ops.potrf (volt_potrf_k_f32 with its workspace initialised) factors these shapes in one launch; the factor against
    fp64 LAPACK, and bitwise against the same entry point without a workspace where that is the plain schedule."""
    from volt_amd import _lib
    x, vol, y, mean = _series_problem(B, n)
    K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
    s2 = torch.full((B,), SIG2, device="cuda")
    f = ops.potrf(K, s2)
    assert int(f.info.abs().sum()) == 0
    Kd = K[:2].double() + SIG2 * torch.eye(n, device="cuda", dtype=torch.float64)
    Lref = torch.linalg.cholesky(Kd)
    assert (f.L[:2].double() - Lref).abs().max() <= 2e-5 * Lref.abs().max()
    if B in (8, 24):
        Np = ops.padded_n(n)
        A = torch.empty(B, Np, Np, device="cuda")
        Winv = torch.empty(B, Np // 128, 128, 128, device="cuda")
        info = torch.empty(B, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().volt_potrf_k_f32(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), Winv.data_ptr(),
                                               info.data_ptr(), B, n, None, 0, 0, _lib.stream_ptr()), "potrf")
        assert torch.equal(torch.tril(A[:, :n, :n]), f.L)
    f2 = ops.potrf(K, s2)
    assert torch.equal(f2.L, f.L)


def test_not_positive_definite_is_reported_per_matrix(ops):
    """info[b] = 1-based failing pivot for the matrix that is not PD, 0 for the others -- and the launch ends (every
    hand-off word is still published, the matrices are independent)."""
    B, n = 8, 2048
    x, vol, y, mean = _series_problem(B, n)
    K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True)).clone()
    K[3, 700, 700] = -5.0
    r = dev(y - mean)
    s2 = torch.full((B,), SIG2, device="cuda")
    ws = ops.MllWorkspace(B, n, True, K.device)
    ops.mll_step(K, r, s2, ws)
    info = ws.info.cpu().numpy()
    assert info[3] == 701 and (np.delete(info, 3) == 0).all()


@pytest.mark.parametrize("B,n", [(16, 4096), (3, 2560), (64, 2048), (8, 1500)])
def test_two_call_factorisation_equals_the_one_straight_from_k(ops, B, n):
    """volt_prepare_f32 + volt_potrf_ws_f32 (a caller that keeps its own prepared copy) runs the same one launch as
    volt_potrf_k_f32 -- its tiles read their input from A itself -- and gives the same factor bit for bit."""
    from volt_amd import _lib
    L = _lib.lib()
    x, vol, y, mean = _series_problem(B, n)
    K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
    s2 = torch.full((B,), SIG2, device="cuda")
    f = ops.potrf(K, s2)
    Np = ops.padded_n(n)
    wp, nbytes = ops._potrf_workspace(B, Np, K.device)
    assert wp is not None
    A, W, info = torch.empty_like(f.A), torch.empty_like(f.Winv), torch.empty_like(f.info)
    st = _lib.stream_ptr()
    _lib.check(L.volt_prepare_f32(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), B, n, st), "prepare")
    _lib.check(L.volt_potrf_ws_f32(A.data_ptr(), W.data_ptr(), info.data_ptr(), B, Np, wp, nbytes, _lib.WS_INITIALISED, st), "potrf")
    assert int(info.abs().sum()) == 0 and int(f.info.abs().sum()) == 0
    assert torch.equal(torch.tril(A), torch.tril(f.A)) and torch.equal(W, f.Winv)
