"""Phase stamps of the fp64 diagonal-block kernel (volt_tune_diag_f64).  GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import _lib, ops
from volt_amd.synthetic import sde_batch
B, n = 8, 512
x, F, vol = sde_batch(B, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
s2 = torch.full((B,), 0.6932, device="cuda", dtype=torch.float64)
Np = ops.padded_n(n)
L = _lib.lib()
A = torch.empty(B, Np, Np, device="cuda", dtype=torch.float64)
W = torch.empty(B, Np // 128, 128, 128, device="cuda", dtype=torch.float64)
info = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.zeros(B, 32, dtype=torch.int64, device="cuda")
for rep in range(3):
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), B, n, _lib.stream_ptr()), "prep")
    _lib.check(L.volt_tune_diag_f64(A.data_ptr(), W.data_ptr(), info.data_ptr(), B, Np, 0, st.data_ptr(), _lib.stream_ptr()), "tune")
    torch.cuda.synchronize()
s = st[0].cpu().numpy().astype(np.float64)
names = {1: "load", 18: "L out", 19: "W compute", 20: "W out"}
for kb in range(4):
    names.update({2 + 4 * kb: f"chol32[{kb}]", 3 + 4 * kb: f"wave 0: out, trailing[{kb}]", 5 + 4 * kb: f"barrier (inv32[{kb}] on wave 3)"})
prev = s[0]
print("s_memrealtime (100 MHz): microseconds")
for i in sorted(names):
    if s[i] == 0: continue
    print(f"{names[i]:26s} {(s[i] - prev) * 0.01:8.2f} us   (t = {(s[i] - s[0]) * 0.01:7.2f})")
    prev = s[i]
if s[31] > s[30] > 0:
    print(f"s_memtime ran at {(s[31] - s[30]) / ((s[20] - s[0]) * 0.01):.0f} MHz over the block")
if s[21] > 0:
    print("phase 1, us after the barrier behind trailing[0]: waves 0, 1 (pivots) %.2f %.2f, waves 2, 3 (idle) %.2f %.2f" % tuple((s[21 + w] - s[5]) * 0.01 for w in range(4)))
