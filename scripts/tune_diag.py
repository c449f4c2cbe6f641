"""Where the diagonal-block kernel's time goes: s_memtime stamps of its phases (thread 0), lone workgroups."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import _lib, ops
from volt_amd.synthetic import sde_batch
B, n = 8, 512
x, F, vol = sde_batch(B, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True)) + 0.69 * torch.eye(n, device="cuda")
L = _lib.lib()
A = K.clone(); W = torch.zeros(B, n // 128, 128, 128, device="cuda"); info = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.zeros(B, 32, dtype=torch.int64, device="cuda")
for rep in range(3):
    A.copy_(K)
    _lib.check(L.volt_tune_diag_f32(A.data_ptr(), W.data_ptr(), info.data_ptr(), B, n, 0, st.data_ptr(), _lib.stream_ptr()), "tune_diag")
    torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
names = ["load image", "A0 pivot 0 (+3 panel blocks)", "B0 trailing x3", "A1 pivot 1 | X0 | 3 updates", "B1 trailing x2",
         "A2 pivot 2 | X1 | 1 update", "B2 trailing x1", "A3 pivot 3 | X2 | W10", "-", "A4 X3 | W20 W21 ...", "T  W30 W31 W32",
         "drain + barrier", "release + flag", "L out (behind the publish)"]
d = (s[:, 1:15] - s[:, 0:14]) / 2400.0         # s_memtime counts shader clocks (~2.4 GHz) -> us
for i, nm in enumerate(names):
    if nm != "-" and i != 9:
        print(f"{nm:30s} {np.median(d[:, i]):7.2f} us")
    elif i == 9:
        print(f"{nm:30s} {np.median(s[:, 10] - s[:, 8]) / 2400.0:7.2f} us")
f = s[:, [3, 16, 17, 18, 19, 20, 4]]
for nm, v in zip(["  A1 wave 0: row loads", "  pivots 0..15", "  rows out + Schur (MFMA)", "  right half in", "  pivots 16..31", "  panel out + barrier"],
                 np.median(np.diff(f, axis=1), axis=0) / 2400.0):
    print(f"{nm:30s} {v:7.2f} us")
for ph, base, ref in (("A1", 21, 3), ("A3", 25, 7)):
    print(f"  {ph}: waves reach the barrier after", " ".join(f"{np.median(s[:, base + w] - s[:, ref]) / 2400.0:5.2f}" for w in range(4)), "us")
print(f"{'total to the publish':30s} {np.median(s[:, 13] - s[:, 0]) / 2400.0:7.2f} us   info {info.tolist()}")
