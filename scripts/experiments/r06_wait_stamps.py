import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch
B, n = 1, 4096
L = _lib.lib()
x, F, vol = sde_batch(1, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
s2 = torch.full((B,), 0.05, device="cuda", dtype=torch.float64)
nb = 32
cnt = L.volt_batch64_describe(B, nb, 0, None, 0)
buf = (C.c_int * (4 * cnt))(); L.volt_batch64_describe(B, nb, 0, buf, cnt)
items = np.array(buf).reshape(cnt, 4)
for _ in range(3): ops.potrf(K, s2)
st = torch.zeros(cnt * 8, dtype=torch.int64, device="cuda")
L.volt_tune_batch64_stamps(C.c_void_p(st.data_ptr())); ops.potrf(K, s2); torch.cuda.synchronize(); L.volt_tune_batch64_stamps(None)
s = st.cpu().numpy().reshape(cnt, 8).astype(float) / 100.0
kind, row, col, mat = items.T
print("D(i): done(i-1) -> polls succeed | -> acquire done | -> barrier passed (stamp 6) | step 0 (to stamp 7)")
for i in range(18, 26):
    d = np.where((kind == 0) & (row == i))[0][0]; dp = np.where((kind == 0) & (row == i - 1))[0][0]
    print(f"  i {i}: {s[d,2]-s[dp,5]:6.1f} {s[d,3]-s[d,2]:6.2f} {s[d,6]-s[d,3]:6.2f} {s[d,7]-s[d,6]:6.2f}")
