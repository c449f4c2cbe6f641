"""One fp32 factorisation alone (ops.potrf = volt_potrf_k_f32), for kernel traces.  usage: B N"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd import ops
from volt_amd.synthetic import sde_batch
B, n = int(sys.argv[1]), int(sys.argv[2])
x, F, vol = sde_batch(B, n)
K = ops.fill(ops.cumtrapz(torch.tensor(vol).cuda(), torch.tensor(x).cuda(), square=True))
s2 = torch.full((B,), 0.6932, device="cuda")
for _ in range(3):
    f = ops.potrf(K, s2)
torch.cuda.synchronize()
print(int(f.info.abs().sum()))
