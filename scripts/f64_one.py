import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from volt_amd import ops, _lib
from volt_amd.synthetic import sde_batch
B, n = int(sys.argv[1]), int(sys.argv[2])
x, F, vol = sde_batch(B, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda().double(), torch.tensor(x).cuda().double(), square=True))
s2 = torch.full((B,), 0.6932, device="cuda", dtype=torch.float64)
for _ in range(2):
    f = ops.potrf(K, s2)
torch.cuda.synchronize()
print(int(f.info.abs().sum()))
