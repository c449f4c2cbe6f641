"""Wall-clock of the reference's per-window pipeline (experiments/stocks/GenerateMultiMeanPreds.py:85-128) at its default
sizes -- ntrain=400, train_iters=400, nsample=1000, forecast_horizon=20, k=300 -- on the HIP path: one ticker the
reference's way (statement for statement), and 64 tickers through the batched driver."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd.synthetic import sde_batch
from volt_amd.train_utils import LearnGPCV, TrainVolModel, TrainVoltMagpieModel
from volt_amd.rollout_utils import Rollouts
from volt_amd.forecast import GenerateStockPredictionsBatch

warnings.simplefilter("ignore")
ntrain, iters, S, H, k = 400, 400, 1000, 20, 300
_, F, _ = sde_batch(64, ntrain - 1, seed=7)
closes = torch.tensor(F).cuda()                                  # [64, 400] prices


def one_ticker(train_y, graph=None):
    dt = 1. / 252
    train_x = (torch.arange(train_y.shape[0] - 1) * dt).cuda()
    test_x = (torch.arange(H) * dt).cuda() + train_x[-1] + train_x[1]
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vol = LearnGPCV(train_x, train_y, train_iters=iters, printing=False, graph=graph)
    torch.cuda.synchronize(); t["LearnGPCV"] = time.perf_counter() - t0; t0 = time.perf_counter()
    vmod, vlh = TrainVolModel(train_x, vol, train_iters=iters, printing=False, graph=graph)
    torch.cuda.synchronize(); t["TrainVolModel"] = time.perf_counter() - t0; t0 = time.perf_counter()
    voltron, lh = TrainVoltMagpieModel(train_x, train_y[1:], vmod, vlh, vol, printing=False, train_iters=iters, k=k,
                                       mean_func="ewma", graph=graph)
    torch.cuda.synchronize(); t["TrainVoltMagpieModel"] = time.perf_counter() - t0; t0 = time.perf_counter()
    vmod.eval()
    samples = Rollouts(train_x, train_y, test_x, voltron, nsample=S)
    torch.cuda.synchronize(); t["Rollouts"] = time.perf_counter() - t0
    assert tuple(samples.shape) == (S, H) and bool(torch.isfinite(samples).all())
    return t


one_ticker(closes[0])                                            # warm-up (library load, allocator)
t = one_ticker(closes[1])
print("one ticker, one window, the reference's arguments (no `graph`: captured where the step is launch-bound):",
      {k_: round(v, 3) for k_, v in t.items()}, "total %.2f s" % sum(t.values()))
one_ticker(closes[2], graph=False)
t = one_ticker(closes[1], graph=False)
print("  graph=False (every iteration eager):", {k_: round(v, 3) for k_, v in t.items()}, "total %.2f s" % sum(t.values()))
one_ticker(closes[2], graph=True)
t = one_ticker(closes[1], graph=True)
print("  graph=True:", {k_: round(v, 3) for k_, v in t.items()}, "total %.2f s" % sum(t.values()))
torch.cuda.synchronize(); t0 = time.perf_counter()
out = GenerateStockPredictionsBatch([f"T{i}" for i in range(64)], closes, forecast_horizon=H, train_iters=iters,
                                    nsample=S, ntrain=ntrain - 1, mean="ewma", k=k, ntimes=1, vol_iters=iters)
torch.cuda.synchronize(); tb = time.perf_counter() - t0
print("64 tickers, one window, batched driver: %.2f s  (%.3f s per ticker)" % (tb, tb / 64), tuple(out.shape))
torch.cuda.synchronize(); t0 = time.perf_counter()
out = GenerateStockPredictionsBatch([f"T{i}" for i in range(64)], closes, forecast_horizon=H, train_iters=iters,
                                    nsample=S, ntrain=ntrain - 1, mean="ewma", k=k, ntimes=1, vol_iters=iters, graph=True)
torch.cuda.synchronize(); tb = time.perf_counter() - t0
print("  with hipGraph-captured GPCV / vol-model loops: %.2f s  (%.3f s per ticker)" % (tb, tb / 64))
