"""ms per gradient step (volt_mll_step_f32, K resident) for short series -- the one-launch step (N <= 1024, few series)
against the launch-per-column path (VOLT_SMALL_NMAX=0 in the environment switches the former off; shapes the library
gates out of the one-launch path run the launch-per-column path in both).  Usage: bench_small_step.py [tag]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

tag = sys.argv[1] if len(sys.argv) > 1 else ("per-column" if os.environ.get("VOLT_SMALL_NMAX") == "0" else "one-launch")
shapes = [(1, 100), (1, 256), (1, 399), (1, 512), (1, 640), (1, 1023), (2, 399), (4, 399), (8, 399), (8, 640), (8, 1023),
          (16, 399), (16, 640), (16, 1023), (24, 399), (24, 640), (32, 399), (32, 640), (48, 399), (64, 100), (64, 256),
          (64, 399), (96, 256), (128, 256), (128, 399), (512, 399)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
print(f"# {tag}: ms per step (median of 5 x 200 steps)")
for B, n in shapes:
    x, F, vol = sde_batch(B, n, 7)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol, device="cuda"), torch.tensor(x, device="cuda"), square=True))
    r = torch.tensor(np.log(F[:, 1:]), device="cuda", dtype=torch.float32)
    r = r - r.mean(-1, keepdim=True)
    s2 = torch.full((B,), 1e-3, device="cuda")
    ws = ops.MllWorkspace(B, n, True, K.device)
    for _ in range(20):
        ops.mll_step(K, r, s2, ws)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.mll_step(K, r, s2, ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 200)
    assert int(ws.info.abs().sum()) == 0
    print(f"B={B:4d} N={n:5d}  {sorted(ts)[2]:.4f} ms   (min {min(ts):.4f})", flush=True)
