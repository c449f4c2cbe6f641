# step 1 of the D piece taken apart (thread 0 = wave 0): wait over | products done | barrier passed | rank-32 update done
# (needs scripts/experiments/r06_b64_instep_stamps.diff applied: a tuning-only macro, kept out of the sources so that their hash stays the evidence's)
# (the diff also holds -DVOLT_B64_PREFETCH: the next step's operands requested before the rank-32 update)
# measured at 1 x 4096, us: wait over -> products done | barrier | -> rank-32 update done (wave 0)
#   as shipped   step 0: 7.4 | 0.04 | 4.1     step 1: 5.6 | 0.04 | 3.6         (MFMA alone: 3.4 / 2.6 and 2.6)
#   prefetch     step 0: 4.9 | 0.04 | 7.5     step 1: 3.2 | 0.2 - 1.2 | 5.5     step 2: 2.2 | 0.1 - 0.6 | 3.9 - 5.4
#   -> operands in registers take 2.4 us off the products; the acquire + the 48 scattered loads per wave issued in front of the
#      update put 2 back, and sub-block 0 is seen 5 us later: a wash (1.775 vs 1.75 ms).  8 x 2048 (XCD-local hand-offs): the same steps.
cd $GRAFT_REPO_ROOT
for FL in "-DVOLT_B64_INSTEP_KB=0" "-DVOLT_B64_INSTEP_KB=1" "-DVOLT_B64_INSTEP_KB=0 -DVOLT_B64_PREFETCH" "-DVOLT_B64_INSTEP_KB=1 -DVOLT_B64_PREFETCH" "-DVOLT_B64_INSTEP_KB=2 -DVOLT_B64_PREFETCH"; do echo "== $FL"; VOLT_EXTRA_FLAGS="-DVOLT_B64_INSTEP_STAMPS -DVOLT_B64_SLICE_STAMPS $FL" python - <<'PY'
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch
for B, n in ((1, 4096),):
    L = _lib.lib()
    x, F, vol = sde_batch(min(B, 4), n)
    vol = np.tile(vol, (B // min(B, 4) + 1, 1))[:B]
    K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
    s2 = torch.full((B,), 0.05, device="cuda", dtype=torch.float64)
    nb = ops.padded_n(n) // 128
    cnt = L.volt_batch64_describe(B, nb, 0, None, 0)
    buf = (C.c_int * (4 * cnt))(); L.volt_batch64_describe(B, nb, 0, buf, cnt)
    items = np.array(buf).reshape(cnt, 4)
    for _ in range(3): ops.potrf(K, s2)
    st = torch.zeros(cnt * 8, dtype=torch.int64, device="cuda")
    L.volt_tune_batch64_stamps(C.c_void_p(st.data_ptr())); ops.potrf(K, s2); torch.cuda.synchronize(); L.volt_tune_batch64_stamps(None)
    s = st.cpu().numpy().reshape(cnt, 8)
    kind, row, col, mat = items.T
    d = [np.where((kind == 0) & (row == i) & (mat == 0))[0][0] for i in range(nb // 2, nb // 2 + 6)]
    v = s[d][:, [3, 6, 7, 2]].astype(float) / 100.0
    print(f"{B}x{n}: step 1 of D(i), us: wait over -> products done (loads + MFMA + LDS) | -> barrier passed | -> rank-32 update done (wave 0)")
    for r in v: print("   ", " ".join(f"{x:6.2f}" for x in np.diff(r)))
PY
done
