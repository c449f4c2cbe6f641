"""Does replaying the MLL+grad step from a hipGraph shorten the launch boundaries?  ms/step eager vs replayed, small batches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for B in (1, 2, 4, 8, 16):
    x, F, vol = sde_batch(B, n)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    r = torch.tensor(np.log(F[:, 1:]) - np.log(F[:, 1:]).mean(-1, keepdims=True)).float().cuda()
    s2 = torch.full((B,), 0.6933, device="cuda")
    for _ in range(3): out = ops.mll_step(K, r, s2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): out = ops.mll_step(K, r, s2)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): out = ops.mll_step(K, r, s2)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.mll_step(K, r, s2)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
    print(f"B={B:2d} N={n}: eager {te*1e3:.3f} ms/step, replayed from a graph {tg*1e3:.3f} ms/step")
