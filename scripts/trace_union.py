"""From a rocprofv3 --kernel-trace CSV: per kernel name, launches, summed duration, and the length of the UNION of
the launch intervals (what bench.py's roofline leg measures live with HIP events when launches of several groups
overlap).  The bench command profiled runs 1 warm-up + 3 timed steps + 6 profiled factorisations = 10 passes."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
iv = defaultdict(list)
for f in glob.glob(os.path.join(root, "*kernel_trace.csv")) + glob.glob(os.path.join(root, "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        iv[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print(f"{'kernel':52s} {'launches':>8s} {'sum_ms':>10s} {'union_ms':>10s} {'avg_us':>9s}")
for k, v in sorted(iv.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    v.sort()
    tot = sum(e - s for s, e in v)
    un, lo, hi = 0, None, None
    for s, e in v:
        if hi is None or s > hi:
            if hi is not None:
                un += hi - lo
            lo, hi = s, e
        else:
            hi = max(hi, e)
    un += hi - lo
    if tot > 1e5:
        print(f"{k[:52]:52s} {len(v):8d} {tot / 1e6:10.3f} {un / 1e6:10.3f} {tot / len(v) / 1e3:9.1f}")
