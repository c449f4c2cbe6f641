"""fp32 factorisation: straight from K (volt_potrf_k_f32, what ops.potrf calls) against prepare + volt_potrf_ws_f32 (a caller
that keeps its own prepared copy).  ms per matrix.  usage: python scripts/bench_potrf_paths.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


L = _lib.lib()
for B, n in ((64, 4096), (16, 4096), (64, 2048), (8, 1500)):
    x, F, vol = sde_batch(min(B, 8), n)
    vol = np.tile(vol, (B // min(B, 8) + 1, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    s2 = torch.full((B,), 0.05, device="cuda")
    f = ops.potrf(K, s2)
    Np = ops.padded_n(n)
    wp, nbytes = ops._potrf_workspace(B, Np, K.device)
    A = torch.empty_like(f.A)
    st = _lib.stream_ptr()

    def two_calls():
        _lib.check(L.volt_prepare_f32(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), B, n, st), "prep")
        _lib.check(L.volt_potrf_ws_f32(A.data_ptr(), f.Winv.data_ptr(), f.info.data_ptr(), B, Np, wp, nbytes, _lib.WS_INITIALISED if wp else 0, st), "potrf")
    t_prep = timeit(lambda: _lib.check(L.volt_prepare_f32(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), B, n, st), "prep"))
    t2 = timeit(two_calls)
    t1 = timeit(lambda: ops.potrf(K, s2))
    ok = torch.equal(torch.tril(A), torch.tril(ops.potrf(K, s2).A))
    print(f"{B:3d} x {n}: from K {t1:.3f} ms | prepare {t_prep:.3f} + potrf_ws {t2 - t_prep:.3f} ms | same factor {ok}", flush=True)
