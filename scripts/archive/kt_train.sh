#!/bin/bash
# Kernel trace of the one-ticker training stages with hipGraph-captured iterations: which kernels one iteration is made of
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/kt_train
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_train -o kt -- python $R/scripts/bench_graph.py > $R/gpurun_out/kt_train.log 2>&1)
tail -3 $R/gpurun_out/kt_train.log
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kt_train/*kernel_stats.csv") + glob.glob("gpurun_out/kt_train/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot/1e6:.1f} ms")
    for r in rows[:28]:
        print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):7d}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  avg {float(r['AverageNs'])/1e3:7.2f} us")
PY
