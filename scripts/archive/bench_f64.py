"""fp64 factorisation / one-launch TRSV timings on the GPU box."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

rows = []
for B, n in [(8, 4096), (1, 4096), (8, 2048), (64, 399)]:
    x, F, vol = sde_batch(B, n)
    V = ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True)
    K = ops.fill(V)
    ms = timeit(lambda: ops.potrf(K))
    f = ops.potrf(K)
    rhs = torch.randn(B, n, device="cuda", dtype=torch.float64)
    t_f = timeit(lambda: ops.trsv(f, rhs)); t_b = timeit(lambda: ops.trsv(f, rhs, transpose=True))
    K32 = ops.fill(V.float())
    s2 = torch.full((B,), 0.69, device="cuda")
    f32 = ops.potrf(K32, s2)
    r32 = rhs.float()
    t_f32 = timeit(lambda: ops.trsv(f32, r32)); t_b32 = timeit(lambda: ops.trsv(f32, r32, transpose=True))
    Np = ops.padded_n(n)
    rows.append({"B": B, "N": n, "potrf_f64_ms": round(ms, 3), "potrf_f64_TF": round(B * Np ** 3 / 3 / ms / 1e9, 2),
                 "trsv_f64_fwd_ms": round(t_f, 3), "trsv_f64_bwd_ms": round(t_b, 3),
                 "trsv_f32_fwd_ms": round(t_f32, 3), "trsv_f32_bwd_ms": round(t_b32, 3),
                 "trsv_f32_fwd_GBps": round(B * Np * Np / 2 * 4 / t_f32 / 1e6, 1)})
    print(json.dumps(rows[-1]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/bench_f64.json", "w"), indent=1)
