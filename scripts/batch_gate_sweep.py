"""ms per MLL+grad step (and per batched factorisation) over a grid of (B, N): run once with VOLT_TUNE=1 VOLT_BATCH=0 (the
launch-per-column schedules) and once with VOLT_BATCH=3 (the one-launch batched step everywhere), then
    python scripts/batch_gate_sweep.py --join a.json b.json
prints the ratio table the (B, N) gate of volt_internal_batch_applies is read from."""
import json, os, sys
if "--join" in sys.argv:
    a, b = (json.load(open(f)) for f in sys.argv[sys.argv.index("--join") + 1:][:2])
    for what in ("step", "potrf"):
        print(f"# {what}: ms launch-per-column / ms one-launch (> 1: the one launch wins)")
        Bs = sorted({int(k.split("x")[0]) for k in a[what]}); Ns = sorted({int(k.split("x")[1]) for k in a[what]})
        print("   B\\N " + " ".join(f"{n:>7d}" for n in Ns))
        for B in Bs:
            print(f"{B:>6d} " + " ".join((f"{a[what][f'{B}x{n}'] / b[what][f'{B}x{n}']:7.3f}" if f"{B}x{n}" in a[what] and f"{B}x{n}" in b[what] else "      -") for n in Ns))
        print("   ms (one launch):")
        for B in Bs:
            print(f"{B:>6d} " + " ".join((f"{b[what][f'{B}x{n}']:7.3f}" if f"{B}x{n}" in b[what] else "      -") for n in Ns))
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch

Bs = [int(v) for v in os.environ.get("BS", "2,3,4,6,8,10,12,16,20,24,32,40,48,64,96").split(",")]
Ns = [int(v) for v in os.environ.get("NS", "1024,1536,2048,3072,4096").split(",")]
out = {"step": {}, "potrf": {}}


def timeit(fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    reps = max(3, int(0.05 / max(1e-4, flops / 100e12)))
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


for n in Ns:
    x, F, vol = sde_batch(4, n)
    for B in Bs:
        if B * n * n * 4 * 3.2 > 200e9:
            continue
        v = np.tile(vol, (B // 4 + 1, 1))[:B]; f = np.tile(F, (B // 4 + 1, 1))[:B]
        K = ops.fill(ops.cumtrapz(torch.tensor(v, dtype=torch.float32).cuda(), torch.tensor(x).cuda(), square=True))
        y = torch.log(torch.tensor(f[:, 1:]).cuda()); r = (y - y.mean(-1, keepdim=True)).float().contiguous()
        s2 = torch.full((B,), 0.05, device="cuda")
        ws = ops.MllWorkspace(B, n, True, K.device)
        Np = ops.padded_n(n)
        out["step"][f"{B}x{n}"] = timeit(lambda: ops.mll_step(K, r, s2, ws), B * 2.0 * Np ** 3 / 3)
        assert int(ws.info.abs().sum()) == 0
        del ws
        out["potrf"][f"{B}x{n}"] = timeit(lambda: ops.potrf(K, s2), B * 1.0 * Np ** 3 / 3)
        print(f"{B}x{n}: step {out['step'][f'{B}x{n}']:.4f} potrf {out['potrf'][f'{B}x{n}']:.4f}", flush=True)
        del K
json.dump(out, open(sys.argv[1], "w"))
