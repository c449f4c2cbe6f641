"""Timeline of the one-launch batched step from its per-workgroup stamps (volt_tune_batch_stamps): per piece kind the
count, summed and mean duration; the launch's span; how busy the workgroup slots were; the tail.
    VOLT_TUNE=1 VOLT_BATCH=2 python scripts/batch_stamps.py 64x4096 [out.npz]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch

sh = sys.argv[1] if len(sys.argv) > 1 else "64x4096"
B, n = map(int, sh.split("x"))
L = _lib.lib()
x, F, vol = sde_batch(min(B, 4), n)
vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 4) + 1, 1))[:B]
K = ops.fill(ops.cumtrapz(torch.tensor(vol, dtype=torch.float32).cuda(), torch.tensor(x).cuda(), square=True))
y = torch.log(torch.tensor(F[:, 1:]).cuda()); r = (y - y.mean(-1, keepdim=True)).float().contiguous()
s2 = torch.full((B,), 0.05, device="cuda")
ws = ops.MllWorkspace(B, n, True, K.device)
nb = ops.padded_n(n) // 128
lad = int(os.environ.get("VOLT_BATCH_LAD", "0"))
while lad > 0 and 2 * lad * B > 128: lad -= 1              # batch_step.hip, batch_lad()
order = int(os.environ.get("VOLT_BATCH_ORDER", "-1"))    # -1: the order the step picks for this shape (round 6: windowed for large batches)
order = -1 if order < 0 else (1000 + order if order >= 2 else order | lad << 1)
cnt = L.volt_batch_describe(B, nb, 1, order, None, 0)
buf = (C.c_int * (4 * cnt))()
assert L.volt_batch_describe(B, nb, 1, order, buf, cnt) == cnt
items = np.array(buf).reshape(cnt, 4)
for _ in range(3): ops.mll_step(K, r, s2, ws)
st = torch.zeros(cnt, 8, dtype=torch.int64, device="cuda")
L.volt_tune_batch_stamps(C.c_void_p(st.data_ptr()))
ops.mll_step(K, r, s2, ws)
torch.cuda.synchronize()
L.volt_tune_batch_stamps(None)
s = st.cpu().numpy()
t0 = s[:, 0].min()
beg, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0; kb, ke = (s[:, 3] - t0) / 100.0, (s[:, 4] - t0) / 100.0          # us
dur = end - beg
kind, b = items[:, 0] & 7, items[:, 0] >> 3
xcc = (s[:, 2] >> 32) & 15
hw = s[:, 2] & 0xffffffff
cu = (hw >> 8) & 15; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 7
span = end.max()
print(f"{sh}: {cnt} pieces, launch span {span:.1f} us, summed workgroup time {dur.sum() / 1e3:.2f} ms = {dur.sum() / span:.1f} slots busy on average")
names = ["diag", "lookahead", "panel", "trtri", "trtri_diag", "alpha"]
for k in range(6):
    m = kind == k
    if m.any(): print(f"  {names[k]:11s} n {m.sum():6d}  sum {dur[m].sum() / 1e3:9.2f} ms  mean {dur[m].mean():8.2f} us  max {dur[m].max():8.2f}")
# K blocks per piece -> microseconds per block for the two-phase tiles
blocks = np.where(kind == 2, items[:, 2], np.where(kind == 3, items[:, 1] - items[:, 2], 0)).astype(float)
for k in (2, 3):
    m = (kind == k) & (blocks >= 8)
    if m.any():
        A_ = np.vstack([blocks[m], np.ones(m.sum())]).T
        sl, ic = np.linalg.lstsq(A_, dur[m], rcond=None)[0]
        print(f"  {names[k]}: duration ~ {sl:.2f} us per K block + {ic:.1f} us (pieces of >= 8 blocks)")
# where a two-phase tile's time goes: entry -> pipeline (table, polls issued, input tile loaded), pipeline, store + publish
for k in (2, 3):
    m = kind == k
    if m.any():
        print(f"  {names[k]}: entry->pipeline {np.mean(kb[m] - beg[m]):6.2f} us, pipeline {np.mean(ke[m] - kb[m]):7.2f} us, store+publish {np.mean(end[m] - ke[m]):6.2f} us (means)")
if s.shape[1] >= 8 and (s[:, 5] > 0).any():
    t5, t6, t7 = (s[:, 5] - t0) / 100.0, (s[:, 6] - t0) / 100.0, (s[:, 7] - t0) / 100.0
    for k in (2, 3):
        m = kind == k
        print(f"  {names[k]}: entry->item loaded {np.mean(t5[m] - beg[m]):5.2f}, ->pipeline {np.mean(kb[m] - t5[m]):5.2f}; pipeline end->reduced {np.mean(t6[m] - ke[m]):5.2f}, ->tile stored + drained (thread 0) {np.mean(t7[m] - t6[m]):5.2f}, ->exit {np.mean(end[m] - t7[m]):5.2f} us")
md = (kind == 0) & (items[:, 1] >= 2)
if md.any() and (s[md, 3] > 0).all():
    print(f"  diag (k >= 2): entry->inputs seen {np.mean(kb[md] - beg[md]):7.2f} us, last K block into the image {np.mean(ke[md] - kb[md]):6.2f}, pivots + inverse + publish {np.mean(end[md] - ke[md]):6.2f} (means; alone: ~7 and ~25)")
# per CU: how much of the launch had 0 / 1 / 2 workgroups INSIDE a pipeline (two-phase tiles: stamps 3..4; look-ahead: whole piece)
key = ((xcc * 8 + se) * 2 + sh_) * 16 + cu
pb = np.where((kind == 2) | (kind == 3), kb, beg); pe = np.where((kind == 2) | (kind == 3), ke, end)
mf = (kind == 1) | (kind == 2) | (kind == 3)
occ = np.zeros(3)
for kk in np.unique(key):
    m = (key == kk) & mf
    e2 = np.concatenate([np.stack([pb[m], np.ones(m.sum())], 1), np.stack([pe[m], -np.ones(m.sum())], 1)])
    e2 = e2[np.argsort(e2[:, 0], kind="stable")]
    c2 = np.cumsum(e2[:, 1]); t2 = e2[:, 0]; w2 = np.diff(t2, append=span)
    for q in range(3): occ[q] += w2[np.minimum(c2, 2) == q].sum()
    occ[0] += t2[0]
occ /= len(np.unique(key)) * span
print(f"  per CU, share of the launch with 0 / 1 / 2 workgroups inside a tile pipeline: {occ[0]:.3f} / {occ[1]:.3f} / {occ[2]:.3f}")
# concurrency over time
ev = np.concatenate([np.stack([beg, np.ones_like(beg)], 1), np.stack([end, -np.ones_like(end)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1]); tt = ev[:, 0]
w = np.diff(tt, append=tt[-1])
print(f"  resident workgroups: time-weighted mean {np.sum(conc * w) / span:.1f}, time with < 256 resident {np.sum(w[conc < 256]):.1f} us, < 448: {np.sum(w[conc < 448]):.1f} us")
print(f"  first piece of the last block column starts at {beg[(kind == 0) & (items[:, 1] == nb - 1)].min():.1f} us; last trtri row starts {beg[(kind == 4) & (items[:, 1] == nb - 1)].min():.1f}")
print("  xcc of workgroup w (first 16):", xcc[:16].tolist(), " distinct (se,sh,cu):", len(set(zip(xcc.tolist(), se.tolist(), sh_.tolist(), cu.tolist()))))
mism = (xcc != (np.arange(cnt) % 8)).sum()
print(f"  pieces run on an XCD other than (piece % 8) -- queues adopted by another XCD, or the one queue of a batch that is not a multiple of 8: {mism}")
# dispatch order: is entry time monotone in w per XCD?
for xq in range(1):
    m = xcc == xq
    inv = (np.diff(beg[m]) < -5.0).sum()
    print(f"  XCD {xq}: {m.sum()} pieces, out-of-order starts (> 5 us early) {inv}")
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], items=items, stamps=s)
