import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import rollout_engine as re_, ops
from volt_amd.synthetic import sde_series, rollout_inputs
from oracle import volt_oracle as vo
n, S, H, k = 96, 4, 10, 7
F, vol = sde_series(n, 11)
pv, z = rollout_inputs(vol[-1], S, H, seed=5)
tx = torch.arange(n, device="cuda") / 252.
test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
logy = torch.tensor(np.log(F[1:]), device="cuda")
samples, info = re_.rollout_series(tx, logy[None], torch.tensor(np.log(vol), device="cuda")[None], test_x,
                                   torch.tensor(pv, device="cuda")[None], torch.tensor(z, device="cuda")[None], 0, k)
print("info", info.cpu().numpy())
out = samples[0].cpu().numpy()
x = (np.arange(n) / 252.).astype(np.float32)
ly = np.log(F[1:]).astype(np.float32)
ref = np.zeros((S, H))
for s in range(S):
    ys = ly.copy()
    for i in range(H):
        full = vo.ewma(ys, k)
        ref[s, i] = (ys[-1] - full[-2]) + full[-1] + np.sqrt(0.5 / 252. * float(pv[s, i]) ** 2) * z[s, i]
        ys = np.append(ys, np.float32(ref[s, i]))
np.set_printoptions(precision=5, linewidth=200)
print(out); print(ref); print(out - ref)
