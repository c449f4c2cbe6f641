"""fp64 factorisation, triangular inverse and MLL step timings (GPU box).  usage: bench_f64_step.py [tag]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

def timeit(fn, reps=int(os.environ.get('REPS', 3))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

tag = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = [(8, 4096), (1, 4096), (2, 4096), (32, 2048), (64, 399)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
rows = []
for B, n in shapes:
    x, F, vol = sde_batch(B, n)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
    s2 = torch.full((B,), 0.6932, device="cuda", dtype=torch.float64)
    Np = ops.padded_n(n)
    f = ops.potrf(K, s2)
    A, W, info = f.A, f.Winv, f.info
    from volt_amd import _lib
    L = _lib.lib()
    A0 = A.clone()
    def potrf_only():
        # refactor an already prepared copy (prepare is a memory pass, timed separately)
        A.copy_(Aprep)
        ops.potrf_f64_inplace(A, W, info)
    Aprep = torch.empty_like(A)
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, Aprep.data_ptr(), B, n, _lib.stream_ptr()), "prep")
    t_copy = timeit(lambda: A.copy_(Aprep))
    t_potrf = timeit(potrf_only) - t_copy
    Y = torch.empty_like(A)
    t_trtri_rows = timeit(lambda: _lib.check(L.volt_trtri_f64(A.data_ptr(), W.data_ptr(), Y.data_ptr(), B, Np, _lib.stream_ptr()), "trtri"))
    Yr = Y.triu().clone()
    tb = int(L.volt_trtri_workspace_bytes_f64(B, Np))
    tws = torch.empty(tb + 256, dtype=torch.uint8, device="cuda")
    twp = (tws.data_ptr() + 255) // 256 * 256 if tb else None
    t_trtri = timeit(lambda: _lib.check(L.volt_trtri_ws_f64(A.data_ptr(), W.data_ptr(), Y.data_ptr(), B, Np, twp, tb, _lib.stream_ptr()), "trtri"))
    trtri_dev = float((Y.triu() - Yr).abs().max() / Yr.abs().max())
    r = torch.randn(B, n, device="cuda", dtype=torch.float64)
    ws = ops.MllWorkspace(B, n, True, "cuda", torch.float64)
    t_step = timeit(lambda: ops.mll_step(K, r, s2, ws))
    ws0 = ops.MllWorkspace(B, n, False, "cuda", torch.float64)
    t_fwd = timeit(lambda: ops.mll_step(K, r, s2, ws0, want_grad=False))
    fl = B * Np ** 3 / 3
    rows.append({"tag": tag, "B": B, "N": n, "potrf_ms": round(t_potrf, 3), "potrf_TF": round(fl / t_potrf / 1e9, 2),
                 "trtri_ms": round(t_trtri, 3), "trtri_TF": round(fl / t_trtri / 1e9, 2), "trtri_launch_per_row_ms": round(t_trtri_rows, 3),
                 "trtri_one_launch": bool(tb), "trtri_rel_dev_from_launch_per_row": trtri_dev,
                 "mll_step_ms": round(t_step, 3), "mll_step_TF": round(2 * fl / t_step / 1e9, 2), "mll_fwd_ms": round(t_fwd, 3),
                 "info": int(info.abs().sum())})
    print(json.dumps(rows[-1]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/bench_f64_step_{tag}.json", "w"), indent=1)
