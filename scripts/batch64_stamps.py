"""Timeline of the fp64 one-launch step from its per-workgroup stamps (volt_tune_batch64_stamps): the latency chain of matrix
0 column by column (D(k): entry, chased product done, exit; US(k+1,k): sum parked, W seen, product done, exit), per kind the
mean duration, microseconds per K block of the chased products, the launch's span and slot occupancy.
    python scripts/batch64_stamps.py 1x4096 [potrf|step]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch

sh = sys.argv[1] if len(sys.argv) > 1 else "1x4096"
what = sys.argv[2] if len(sys.argv) > 2 else "potrf"
B, n = map(int, sh.split("x"))
L = _lib.lib()
x, F, vol = sde_batch(min(B, 4), n)
vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]; F = np.tile(F, (B // min(B, 4) + 1, 1))[:B]
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
y = torch.log(torch.tensor(F[:, 1:]).cuda().double()); r = (y - y.mean(-1, keepdim=True)).contiguous()
s2 = torch.full((B,), 0.05, device="cuda", dtype=torch.float64)
nb = ops.padded_n(n) // 128
has_y = what == "step"
cnt = L.volt_batch64_describe(B, nb, int(has_y), None, 0)
buf = (C.c_int * (4 * cnt))()
assert L.volt_batch64_describe(B, nb, int(has_y), buf, cnt) == cnt
items = np.array(buf).reshape(cnt, 4)
ws = ops.MllWorkspace(B, n, True, K.device, torch.float64)
run = (lambda: ops.mll_step(K, r, s2, ws)) if has_y else (lambda: ops.potrf(K, s2))
for _ in range(3): run()
DS = "VOLT_B64_DIAG_STAMPS" in os.environ.get("VOLT_EXTRA_FLAGS", "")
st = torch.zeros(cnt * 8 + (32 * nb * B if DS else 0), dtype=torch.int64, device="cuda")
L.volt_tune_batch64_stamps(C.c_void_p(st.data_ptr()))
run()
torch.cuda.synchronize()
L.volt_tune_batch64_stamps(None)
sall = st.cpu().numpy()
s = sall[:cnt * 8].reshape(cnt, 8)
assert (s[:, 0] > 0).all(), "the shape did not run as one launch"
t0 = s[:, 0].min()
T = lambda col: (s[:, col] - t0) / 100.0
beg, end, t3, t4, t5, t6, t7 = T(0), T(1), T(3), T(4), T(5), T(6), T(7)
kind, row, col, mat = items.T
dur = end - beg
span = end.max()
print(f"{sh} {what}: {cnt} pieces, launch span {span:.1f} us, {dur.sum() / span:.1f} slots busy on average")
names = ["D", "US", "TD", "T", "LA"]
for k in range(5):
    m = kind == k
    if m.any(): print(f"  {names[k]:3s} n {m.sum():6d}  mean {dur[m].mean():8.2f} us  max {dur[m].max():8.2f}")
blocks = np.where(kind == 1, col, np.where(kind == 3, row - col, 0)).astype(float)
for k in (1, 3):
    m = (kind == k) & (blocks >= 8)
    if m.sum() > 4:
        A_ = np.vstack([blocks[m], np.ones(m.sum())]).T
        sl, ic = np.linalg.lstsq(A_, dur[m], rcond=None)[0]
        print(f"  {names[k]}: duration ~ {sl:.2f} us per K block + {ic:.1f} us")
m = kind == 1
if m.any(): print(f"  US: sum parked -> last sub-block seen {np.mean(t4[m] - t3[m]):7.2f} us, -> tile complete {np.mean(t5[m] - t4[m]):6.2f} us (means)")
m = kind == 3
if m.any(): print(f"  T: sum parked -> W seen {np.mean(t4[m] - t3[m]):7.2f} us, second product {np.mean(t5[m] - t4[m]):6.2f} us, store + publish {np.mean(end[m] - t5[m]):6.2f} us (means)")
print("chain of matrix 0 (us): D(i) entry | sum parked | last sub-block of D(i-1) seen | tile (i,i-1) + rank-32 done | exit  (column = step between 'seen' times)")
prev = None
for i in range(nb):
    d = np.where((kind == 0) & (row == i) & (mat == 0))[0][0]
    if i == 0:
        print(f"  i  0  in {beg[d]:8.1f}                                        out {end[d]:8.1f}")
        continue
    xs = np.where((kind == 1) & (row == i) & (col == i - 2) & (mat == 0))[0]
    xt = f" | X tile ({i},{i - 2}): in {beg[xs[0]]:7.1f} sum {t3[xs[0]]:7.1f} last sub seen {t4[xs[0]]:7.1f} out {end[xs[0]]:7.1f}" if len(xs) else ""
    line = f"  i {i:2d}  in {beg[d]:8.1f} sum {t3[d]:8.1f} sub-blocks 0, 1 seen {t6[d]:8.1f} {t7[d]:8.1f} seen {t4[d]:8.1f} done {t5[d]:8.1f} (+{t5[d] - t4[d]:4.1f}) out {end[d]:8.1f}"
    if prev is not None: line += f" | column {t4[d] - prev:6.1f}"
    line += xt
    prev = t4[d]
    print(line)
if DS:
    ds = sall[cnt * 8:].reshape(nb, B, 32)[:, 0, :]        # matrix 0
    lab = {0: "entry", 1: "image", 2: "piv0", 3: "trail0", 5: "inv0+pub1", 6: "piv1", 7: "trail1", 9: "inv1+pub2", 10: "piv2", 11: "trail2",
           13: "inv2+pub3", 14: "piv3", 15: "-", 17: "inv3+pub4", 18: "(Lout)", 19: "Wcomp", 20: "Wout"}
    for i in (1, nb // 2, nb - 1):
        v = ds[i]
        idx = [j for j in sorted(lab) if v[j] > 0]
        print(f"diagonal block {i} (us since its entry): " + "  ".join(f"{lab[j]} {(v[j] - v[0]) / 100.0:.1f}" for j in idx)
              + (f"  [phase 1, us after pub1: waves 0, 1 (pivots) {(v[21] - v[5]) / 100.0:.1f} {(v[22] - v[5]) / 100.0:.1f}, waves 2, 3 (idle) {(v[23] - v[5]) / 100.0:.1f} {(v[24] - v[5]) / 100.0:.1f}]" if v[21] > 0 else "")
              + (f"  [publication of sub-block 0 by wave 3: enters at {(v[25] - v[0]) / 100.0:.1f}, word stored and acknowledged at {(v[26] - v[0]) / 100.0:.1f}]" if v[26] > 0 else "")
              + (f"  [s_memtime ran at {(v[31] - v[30]) / ((v[20] - v[0]) / 100.0):.0f} MHz]" if v[31] > v[30] > 0 else ""))
