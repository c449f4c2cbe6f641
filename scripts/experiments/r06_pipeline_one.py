"""One ticker, one window of the reference's pipeline at its default sizes, the three fits only (for a kernel trace)."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volt_amd.synthetic import sde_batch
from volt_amd.train_utils import LearnGPCV, TrainVolModel, TrainVoltMagpieModel
warnings.simplefilter("ignore")
ntrain, iters, k = 400, 400, 300
_, F, _ = sde_batch(2, ntrain - 1, seed=7)
closes = torch.tensor(F).cuda()
for rep in range(2):
    train_y = closes[rep]
    train_x = (torch.arange(train_y.shape[0] - 1) / 252.).cuda()
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vol = LearnGPCV(train_x, train_y, train_iters=iters, printing=False)
    torch.cuda.synchronize(); t["LearnGPCV"] = time.perf_counter() - t0; t0 = time.perf_counter()
    vmod, vlh = TrainVolModel(train_x, vol, train_iters=iters, printing=False)
    torch.cuda.synchronize(); t["TrainVolModel"] = time.perf_counter() - t0; t0 = time.perf_counter()
    voltron, lh = TrainVoltMagpieModel(train_x, train_y[1:], vmod, vlh, vol, printing=False, train_iters=iters, k=k, mean_func="ewma")
    torch.cuda.synchronize(); t["TrainVoltMagpieModel"] = time.perf_counter() - t0
    print(t, flush=True)
