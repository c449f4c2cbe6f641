"""The one-launch batched step (csrc/batch_step.hip) against the plain launch-per-column path: BITWISE on out, alpha, info and
on per-tile checksums of the factor and the inverse (the tile bodies are shared, so every bit must agree), plus ms/step.
    reference process:  VOLT_TUNE=1 VOLT_BATCH=0 VOLT_SCHED=0 VOLT_SPLITK_TARGET=1 VOLT_LONG=0 VOLT_SMALL_NMAX=0 \\
                        python scripts/batch_check.py --dump /tmp/ref.pt 8x4096 ...
    candidate process:  VOLT_TUNE=1 VOLT_BATCH=2 python scripts/batch_check.py --cmp /tmp/ref.pt 8x4096 ...
(one process per schedule: the knobs are read once).  --notime skips the timing; --reps-check N repeats the step N times and
checks every repeat bitwise against the first (hand-off races show up as run-to-run differences)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch

args = sys.argv[1:]
shapes = [a for a in args if "x" in a and a[0].isdigit()] or ["2x384", "3x1000", "8x1024", "16x2048", "8x4096", "64x2048", "64x4096"]
notime = "--notime" in args
dump = args[args.index("--dump") + 1] if "--dump" in args else None
cmp_ = args[args.index("--cmp") + 1] if "--cmp" in args else None
nrep = int(args[args.index("--reps-check") + 1]) if "--reps-check" in args else 0
ref = torch.load(cmp_) if cmp_ else {}
full = [None]
out = {}


def tile_sums(M, Np):
    B = M.shape[0]
    v = M.contiguous().view(torch.int32).view(B, Np // 128, 128, Np // 128, 128).to(torch.int64)
    return v.sum(dim=(2, 4))


def snapshot(ws, B, n, Np):
    nA = B * Np * Np
    base = (ws.buf.data_ptr() + 255) // 256 * 256 - ws.buf.data_ptr()
    flat = ws.buf[base:].view(torch.float32)
    al = lambda f: (f * 4 + 255) // 256 * 256 // 4
    offY = al(nA) + al(B * (Np // 128) * 128 * 128) + 4 * al(B * Np)
    A = torch.tril(flat[:nA].view(B, Np, Np))
    Y = torch.triu(flat[offY: offY + nA].view(B, Np, Np))
    if "--full" in args and nA * 4 <= (1 << 29):
        full[0] = (A.cpu().clone(), Y.cpu().clone())
    return tile_sums(A, Np).cpu(), tile_sums(Y, Np).cpu()


for sh in shapes:
    B, n = map(int, sh.split("x"))
    x, F, vol = sde_batch(min(B, 4), n)
    vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 4) + 1, 1))[:B]
    vol = vol * (1.0 + 0.01 * np.arange(B)[:, None])                    # the matrices differ
    K = ops.fill(ops.cumtrapz(torch.tensor(vol, dtype=torch.float32).cuda(), torch.tensor(x).cuda(), square=True))
    y = torch.log(torch.tensor(F[:, 1:]).cuda())
    r = (y - y.mean(-1, keepdim=True)).float().contiguous()
    s2 = torch.full((B,), 0.05, device="cuda")
    Np = ops.padded_n(n)
    ws = ops.MllWorkspace(B, n, True, K.device)
    o, a, info = ops.mll_step(K, r, s2, ws)
    torch.cuda.synchronize()
    tA, tY = snapshot(ws, B, n, Np)
    cur = dict(out=o.cpu().clone(), alpha=a.cpu().clone(), info=info.cpu().clone(), tA=tA, tY=tY)
    if full[0] is not None:
        cur["A"], cur["Y"] = full[0]
        full[0] = None
    out[sh] = cur
    msg = f"{sh:>9s}: info {int(info.abs().sum())}"
    if sh in ref:
        rf = ref[sh]
        eq = {k: bool(torch.equal(rf[k], cur[k])) for k in ("out", "alpha", "info", "tA", "tY")}
        msg += "  bitwise " + ("OK  " if all(eq.values()) else "FAIL " + str(eq))
        if not eq["tA"]: msg += f" L tiles {(rf['tA'] != tA).nonzero()[:5].tolist()}"
        if not eq["tY"]: msg += f" Y tiles {(rf['tY'] != tY).nonzero()[:5].tolist()}"
        for nm, tk in (("A", "tA"), ("Y", "tY")):
            if nm in rf and nm in cur and not eq[tk]:
                d = (rf[nm] != cur[nm])
                bad = (rf[tk] != cur[tk]).nonzero()
                # the differing tile that nothing wrong feeds: smallest (row + col) sum is a decent proxy
                b0, r0, c0 = min(bad.tolist(), key=lambda t: (t[1] + t[2], t[0]))
                sub = d[b0, r0 * 128:(r0 + 1) * 128, c0 * 128:(c0 + 1) * 128]
                rows = sub.any(dim=1).nonzero().flatten().tolist(); cols = sub.any(dim=0).nonzero().flatten().tolist()
                msg += f"\n      {nm} tile {(b0, r0, c0)}: {int(sub.sum())} elements differ; rows {rows[:40]}{'...' if len(rows) > 40 else ''} cols {cols[:40]}{'...' if len(cols) > 40 else ''}"
                rr, cc = rows[0], cols[0]
                msg += f" e.g. [{rr},{cc}] ref {rf[nm][b0, r0 * 128 + rr, c0 * 128 + cc].item():.6g} got {cur[nm][b0, r0 * 128 + rr, c0 * 128 + cc].item():.6g}"
        if not eq["out"]: msg += f" out ref {rf['out'][0, :6].tolist()} got {cur['out'][0, :6].tolist()}"
    if nrep:
        bad = 0
        for _ in range(nrep):
            o2, a2, i2 = ops.mll_step(K, r, s2, ws)
            if not (torch.equal(o2.cpu(), cur["out"]) and torch.equal(a2.cpu(), cur["alpha"]) and int(i2.abs().sum()) == 0): bad += 1
        msg += f"  repeats {nrep} bad {bad}"
    if not notime:
        for _ in range(3): ops.mll_step(K, r, s2, ws)
        torch.cuda.synchronize()
        flops = B * 2.0 * Np ** 3 / 3
        reps = max(3, int(0.2 / max(1e-4, flops / 100e12)))
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ops.mll_step(K, r, s2, ws)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        ms = float(np.median(ts))
        msg += f"   {ms:8.4f} ms/step {B * 2.0 * n ** 3 / 3 / ms / 1e9:6.1f} TF/s (min {min(ts):.4f})"
    print(msg, flush=True)
    del K, ws
if dump:
    torch.save(out, dump)
