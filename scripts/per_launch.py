"""Per-launch durations of one factor+inverse in lockstep (one stream, whole batch per launch), against the
MFMA-peak time of each launch's algorithmic flops.  GPU box."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import _lib, ops
from volt_amd.synthetic import sde_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
x, F, vol = sde_batch(B, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
Np = ops.padded_n(n); nb = Np // 128
ws = ops.MllWorkspace(B, n, True, "cuda"); resid = torch.randn(B, n, device="cuda")
inf = torch.empty(B, dtype=torch.int32, device="cuda"); s2 = torch.full((B,), 0.6933, device="cuda")
GROUPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L = _lib.lib()
ms_s, ms_u, cnt = (ctypes.c_float * 2)(), (ctypes.c_float * 2)(), (ctypes.c_int * 2)()
per = (ctypes.c_float * (4 * nb + 16))()
tot = np.zeros(nb + 1)
reps = 3
for r in range(reps + 1):
    _lib.check(L.volt_profile_step_f32(K.data_ptr(), n, n * n, resid.data_ptr(), s2.data_ptr(), ws.out.data_ptr(),
                                       ws.alpha.data_ptr(), ws.ptr, inf.data_ptr(), B, n, GROUPS, _lib.stream_ptr(), ms_s, ms_u,
                                       cnt, per), "profile")
    if r: tot += np.array(list(per))[:nb + 1]
tot /= reps
c = 128.0 ** 3
rows = []
for k in range(nb):
    fl = (nb - k - 1) * (2 * k + 1) * c + k * c + 2 * c / 3 + (sum(2 * c * (k - 1 - j) for j in range(k - 1)) if k > 0 else 0)
    tiles = 1 + (1 if 1 <= k < nb - 1 else 0) + (nb - k - 1) + k
    rows.append((k, tiles * B, tot[k] * 1e3, B * fl / 157.3e12 * 1e6))
    print(f"k={k:2d} tiles={tiles*B:5d}  {tot[k]*1e3:8.1f} us   peak-time {B*fl/157.3e12*1e6:8.1f} us   eff {B*fl/157.3e12*1e3/tot[k]:.3f}")
print(f"last trtri row: {tot[nb]*1e3:.1f} us; total {tot.sum():.3f} ms")
