#!/bin/bash
# The kernel sequence of ONE hipGraph-replayed iteration of TrainVoltMagpieModel (one ticker, n = 399)
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/kt_iter
cat > /tmp/one_stage.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
from volt_amd import train_utils as tu
from volt_amd.synthetic import sde_series
n = 399
F, vol = sde_series(n, 7)
tx = torch.arange(n, device="cuda") / 252.
prices = torch.tensor(F, device="cuda"); v = torch.tensor(vol, device="cuda")
stage = os.environ.get("STAGE", "magpie")
if stage == "magpie":
    tu.TrainVoltMagpieModel(tx, prices[1:], None, None, v, train_iters=12, k=300, graph=True)
elif stage == "vol":
    tu.TrainVolModel(tx, v, train_iters=12, graph=True)
else:
    tu.LearnGPCV(tx, prices, train_iters=12, graph=True)
torch.cuda.synchronize()
PY
(cd /tmp && R=$R rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_iter -o kt -- python /tmp/one_stage.py > $R/gpurun_out/kt_iter.log 2>&1)
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/kt_iter/*kernel_trace.csv") + glob.glob("gpurun_out/kt_iter/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# last iteration: from the last prepare_kernel on ... back to the one before
idx = [i for i, r in enumerate(rows) if "fill_kernel" in r[2] or "prepare_kernel" in r[2]]
marks = [i for i, r in enumerate(rows) if "mll_scalars" in r[2] or "small_step_kernel" in r[2]]
a, b = marks[-2], marks[-1]
t0 = rows[a + 1][0]
for s, e, nm in rows[a + 1:b + 1]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  {nm[:110]}")
print("iteration span", (rows[b][1] - rows[a][1]) / 1e3, "us,", b - a, "kernels")
PY
