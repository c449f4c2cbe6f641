"""Where one short-series step spends its time: per-piece timestamps of small_step_kernel (volt_tune_small_stamps).
Usage: small_stamps.py [B] [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 399
x, F, vol = sde_batch(B, n, 7)
K = ops.fill(ops.cumtrapz(torch.tensor(vol, device="cuda"), torch.tensor(x, device="cuda"), square=True))
r = torch.tensor(np.log(F[:, 1:]), device="cuda", dtype=torch.float32)
r = r - r.mean(-1, keepdim=True)
s2 = torch.full((B,), 1e-3, device="cuda")
ws = ops.MllWorkspace(B, n, True, K.device)
for _ in range(10):
    ops.mll_step(K, r, s2, ws)
nb = ops.padded_n(n) // 128
L = _lib.lib()
# ONE series of more than VOLT_LONG_NMIN (2) block columns runs long_step_kernel: the piece list comes from the library (csrc/long_sched.h)
LONG = B == 1 and nb > int(os.environ.get("VOLT_LONG_NMIN", 2)) and nb <= 32 and os.environ.get("VOLT_LONG", "1") != "0"
if LONG:
    first, emin = int(os.environ.get("VOLT_LONG_FIRST", 0)), int(os.environ.get("VOLT_LONG_EMIN", -1))
    npieces = L.volt_long_describe(nb, first, emin, None, 0, None, None)
    items = np.zeros((npieces, 4), dtype=np.int32)
    L.volt_long_describe(nb, first, emin, items.ctypes.data, npieces, None, None)
    G = npieces
else:
    G = B * (nb * (nb + 1) // 2 + (nb - 1) * (nb - 2) // 2 + nb + max(nb - 2, 0))
st = torch.zeros(G, 16, dtype=torch.int64, device="cuda")
L.volt_tune_small_stamps(st.data_ptr())
ops.mll_step(K, r, s2, ws)
torch.cuda.synchronize()
L.volt_tune_small_stamps(0)
s = st.cpu().numpy().astype(np.float64)
t0 = s[:, 0].min()
us = (s - t0) / 100.0
us[s == 0] = np.nan
KINDS = ["D", "S", "P", "U", "T", "T", "eP", "eT", "eU", "R"]


def name_long(w):
    x, y = int(items[w, 0]), int(items[w, 1])
    kind, a, bb = x & 255, (x >> 8) & 255, (x >> 16) & 255
    if kind == 0:
        return 0, "D(0)"
    if kind in (1, 3, 9):
        return 0, f"{KINDS[kind]}({a})" + (f"/{y}" if y else "")
    if 6 <= kind <= 8:
        return 0, f"{KINDS[kind]}({a},{bb})[{y & 255}:{(y >> 8) & 255}]"
    return 0, f"{KINDS[kind]}({a},{bb})" + (f"/{y}" if y else "")


def name(w):
    if LONG:
        return name_long(w)
    for k in range(nb + 1):
        npan = max(nb - k - 2, 0) if k < nb else 0
        nu = 1 if (k >= 1 and k + 1 <= nb - 1) else 0
        nh = 1 + npan + nu if k < nb else 0
        if w < nh * B:
            p, b = divmod(w, B)
            if p == 0:
                return b, f"S({k})" if k else "D(0)"
            if p <= npan:
                return b, f"P({k + 1 + p},{k})"
            return b, f"U({k + 1})"
        if w < (nh + k) * B:
            w -= nh * B
            if B * nb <= 224:
                return w % B, f"T({k - 1},{w // B})"
            return w // k, f"T({k - 1},{w % k})"
        w -= (nh + k) * B
    raise ValueError


print(f"B={B} N={n}: step spans {np.nanmax(us):.1f} us;  columns: enter, wait1, wait2 (spine: ahead part done), work (spine: image ready), published, exit (us from the first entry)")
for w in range(G):
    b, nm = name(w)
    if b in (0, B - 1) and (not LONG or nm[0] in os.environ.get("SHOW", "DS")):
        print(f"  series {b:3d} {nm:8s} " + " ".join("    -  " if np.isnan(v) else f"{v:7.1f}" for v in us[w, :6]))
        if nm.startswith("R("):
            print("             R: slabs seen " + " ".join(f"{v:6.1f}" for v in us[w, 6:10]))
        if nm.startswith("S("):
            print("             spine: flags seen " + " ".join(f"{v:6.1f}" for v in us[w, 6:10]) + "  solved %.1f  all there %.1f  rank32 %.1f  image %.1f  out %.1f" % tuple(us[w, 10:15]))
