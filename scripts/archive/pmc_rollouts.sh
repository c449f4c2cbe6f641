#!/bin/bash
# Read-request size mix of rollout_bordered_kernel at config 5's per-GPU share (settles FETCH_SIZE 528 GB vs the
# 895 GB the algorithm streams: FETCH_SIZE tallies every request at 64 B).  One TCC pass, hard timeout.
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_rollreq
cd /tmp
timeout -k 5 ${PMC_TIMEOUT:-200} rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rollreq/rdreq -o rdreq -- python $R/scripts/bench_rollouts.py > $R/gpurun_out/pmc_rollreq/rdreq.log 2>&1
echo "rc=$?"
cd $R
python - <<'PY'
import csv, glob, json
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob("gpurun_out/pmc_rollreq/rdreq/*counter_collection.csv") + glob.glob("gpurun_out/pmc_rollreq/rdreq/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "rollout" in k or "y_times" in k or "fill" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
out = {}
for k, a in agg.items():
    disp = max(n[(k, c)] for c in a)
    r = {c: v / disp for c, v in a.items()}
    tot, r32, r64, r128 = (r.get("TCC_EA0_RDREQ_sum", 0), r.get("TCC_EA0_RDREQ_32B_sum", 0), r.get("TCC_EA0_RDREQ_64B_sum", 0),
                           r.get("TCC_EA0_RDREQ_128B_sum", 0))
    r["bytes_if_sized"] = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(0.0, tot - r32 - r64 - r128)
    r["bytes_at_64_each"] = 64 * tot
    out[k] = r
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_rollreq/summary.json", "w"), indent=1)
PY
