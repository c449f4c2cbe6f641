import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from volt_amd import rollout_engine as re_
from volt_amd.synthetic import rollout_inputs, sde_series
from oracle import volt_oracle as vo
d = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")
for H, mode, theta in [(1, 0, None), (2, 1, None), (50, 2, None), (256, 0, 0.3), (257, 0, None), (300, 1, None), (512, 3, None), (700, 0, None), (1024, 0, 0.1), (1024, 1, None), (1024, 2, None)]:
    n, S, k = 200, 9, 25
    F, vol = sde_series(n, 4)
    pv, z = rollout_inputs(vol[-1], S, H, seed=H)
    tx = torch.arange(n, device="cuda") / 252.
    test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
    logy = torch.log(d(F)[1:])[None]
    kw = dict(latent_mean=logy.mean(), theta=theta) if theta is not None else {}
    args = (tx, logy, torch.log(d(vol))[None], test_x, d(pv)[None], d(z)[None], mode, k)
    a, ia = re_.rollout_series(*args, **kw)
    b, ib = re_.rollout_series(*args, resubstitute=True, **kw)
    print(H, mode, theta, "dev", float((a - b).abs().max()), "range", float(a.min()), float(a.max()), "info eq", bool(torch.equal(ia, ib)), flush=True)
