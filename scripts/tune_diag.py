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
st = torch.zeros(B, 16, dtype=torch.int64, device="cuda")
for rep in range(3):
    A.copy_(K)
    _lib.check(L.volt_tune_diag_f32(A.data_ptr(), W.data_ptr(), info.data_ptr(), B, n, 0, st.data_ptr(), _lib.stream_ptr()), "tune_diag")
    torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
d = (s[:, 1:11] - s[:, 0:10]) / 2400.0         # s_memtime counts shader clocks (2.4 GHz) -> us
names = ["load image", "factor32(0)", "panel+trail+factor32(1)", "panel+trail+factor32(2)", "panel+trail+factor32(3)",
         "-", "inverse (MFMA)", "W out", "drain+barrier", "release+flag"]
for i, nm in enumerate(names):
    print(f"{nm:28s} {np.median(d[:, i]):7.2f} us")
f = s[:, [1, 11, 12, 13, 14]]
for nm, v in zip(["  factor32(0): load+pivots", "  L_kk out", "  X = L^-1", "  X to image"], np.median(np.diff(f, axis=1), axis=0) / 2400.0):
    print(f"{nm:28s} {v:7.2f} us")
print(f"{'L out (behind the publish)':28s} {np.median(s[:, 15] - s[:, 10]) / 2400.0:7.2f} us")
print(f"{'total to the publish':28s} {np.median(s[:, 10] - s[:, 0]) / 2400.0:7.2f} us   info {info.tolist()}")
