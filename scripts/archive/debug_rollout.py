"""Append-only vs full re-substitution rollouts: first step at which the two differ, and by how much.  GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import rollout_engine as re_
from volt_amd.synthetic import rollout_inputs, sde_series
n, S, k = 200, 9, 25
F, vol = sde_series(n, 4)
for H in (3, 4, 5, 8, 50, 300):
    for mode in (0, 2):
        pv, z = rollout_inputs(vol[-1], S, H, seed=H)
        tx = torch.arange(n, device="cuda") / 252.
        test_x = torch.arange(H, device="cuda") / 252. + tx[-1] + tx[1]
        logy = torch.log(torch.tensor(F).cuda()[1:])[None]
        args = (tx, logy, torch.log(torch.tensor(vol).cuda())[None], test_x, torch.tensor(pv).cuda()[None], torch.tensor(z).cuda()[None], mode, k)
        a, ia = re_.rollout_series(*args)
        b, ib = re_.rollout_series(*args, resubstitute=True)
        d = (a - b).abs()[0]
        steps = (d.max(0).values > 0).nonzero().flatten().tolist()
        print(f"H={H} mode={mode}: max diff {float(d.max()):.3e}, first differing step {steps[:1]}, #steps differing {len(steps)}")
