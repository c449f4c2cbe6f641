# step 1 of the D piece taken apart (thread 0 = wave 0): wait over | products done | barrier passed | rank-32 update done
# (needs scripts/experiments/r06_b64_instep_stamps.diff applied: a tuning-only macro, kept out of the sources so that their hash stays the evidence's)
# measured: 1 x 4096 / 8 x 2048, us: wait over -> products done 5.4 - 6.0 (2.6 of MFMA) | barrier 0.1 | rank-32 update 3.5 (2.6 of MFMA) -- the same with XCD-local and with agent-scope hand-offs
cd $GRAFT_REPO_ROOT
VOLT_EXTRA_FLAGS="-DVOLT_B64_INSTEP_STAMPS -DVOLT_B64_SLICE_STAMPS" python - <<'PY'
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch
for B, n in ((1, 4096), (8, 2048)):
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
