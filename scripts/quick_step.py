"""ms per MLL+grad step (volt_mll_step_f32, K resident) for a list of B x N shapes -- the quick A/B harness of round 4.
    python scripts/quick_step.py [BxN ...]        default: 64x4096 8x4096 16x4096 64x2048 1x4096 64x399
Timing: torch events around `reps` back-to-back calls on the current stream, median of 5 rounds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

shapes = [a for a in sys.argv[1:] if "x" in a] or ["64x4096", "8x4096", "16x4096", "64x2048", "1x4096", "64x399"]
raw = float(os.environ.get("RAW", "1e-5"))
for sh in shapes:
    B, n = map(int, sh.split("x"))
    x, F, vol = sde_batch(min(B, 4), n)
    vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 4) + 1, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda())
    r = (y - y.mean(-1, keepdim=True)).contiguous()
    s2 = torch.full((B,), float(np.log1p(np.exp(raw)) + 1e-4), device="cuda")
    ws = ops.MllWorkspace(B, n, True, K.device)
    for _ in range(3): ops.mll_step(K, r, s2, ws)
    torch.cuda.synchronize()
    flops = B * 2.0 * (ops.padded_n(n)) ** 3 / 3
    reps = max(3, int(0.2 / max(1e-4, flops / 100e12)))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.mll_step(K, r, s2, ws)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ms = float(np.median(ts))
    assert os.environ.get("NOCHECK") or int(ws.info.abs().sum()) == 0
    print(f"{sh:>9s}: {ms:8.4f} ms/step  {B * 2.0 * n ** 3 / 3 / ms / 1e9:6.1f} TF/s  (min {min(ts):.4f} max {max(ts):.4f}, {reps} reps)", flush=True)
    del K, ws
