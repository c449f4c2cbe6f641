#!/bin/bash
# Kernel trace of one MLL+grad step at a small batch (split-K schedule): per-launch duration and gap, all kernels of the step
B=${1:-1}
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_small -o kt -- python $R/bench.py --batch $B --n 4096 --steps 3 --warmup 1 --no-rollouts --no-cpu-baseline --no-aux-legs > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/kt_small/*kernel_trace.csv") + glob.glob("gpurun_out/kt_small/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("volt::", ""), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
idx = [i for i, r in enumerate(rows) if "factor_step_s" in r[2]]
# the last timed step: from the kernel after the 34th-from-last split launch's predecessor up to the next non-step kernel
last = idx[-33]
lo = last
while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 200000 and "factor_step" not in rows[lo - 1][2]: lo -= 1
hi = idx[-1]
while hi + 1 < len(rows) and rows[hi + 1][0] - rows[hi][1] < 200000 and "factor_step" not in rows[hi + 1][2]: hi += 1
prev = None
for s, e, nm, g in rows[lo:hi + 1]:
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{nm[:44]:44s} grid {g//256:5d} WGs  dur {(e-s)/1e3:7.1f} us  gap {gap:6.1f} us")
    prev = e
print("span", (rows[hi][1] - rows[lo][0]) / 1e3, "us")
PY
