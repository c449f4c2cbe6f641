"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("volt::", "").replace("void ", "")
            c = row["Counter_Name"]
            agg[k][c] += float(row["Counter_Value"])
            calls[k][c] += 1
names = sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", agg[k].get("GRBM_GUI_ACTIVE", 0)))
for k in names:
    if not k.startswith(("potrf", "trtri", "factor_", "batch_step", "alpha_sum", "diag_trtri", "fill", "prepare", "reduce", "y_times", "trsv", "rollout", "sum_zpart", "mll_", "tune_", "update64", "trsm64", "diag64")):
        continue
    a = agg[k]
    n = max(calls[k].values())
    print(f"== {k}  dispatches={n}")
    for c in sorted(a):
        print(f"   {c:32s} {a[c]:.6g}   per-dispatch {a[c]/max(1,calls[k][c]):.6g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        # MFMA busy cycles are summed over SIMDs? report ratios that are unit-free
        if a.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
            print(f"   MFMA pipe utilisation = mfma_busy / (gui_active/8 * 1024 SIMDs) = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
        if a.get("SQ_WAVE_CYCLES"):
            w = a["SQ_WAVE_CYCLES"]
            print(f"   wait_any/wave {a.get('SQ_WAIT_ANY',0)/w:.3f}  wait_inst_any/wave {a.get('SQ_WAIT_INST_ANY',0)/w:.3f}  active_inst_any/wave {a.get('SQ_ACTIVE_INST_ANY',0)/w:.3f}")
    if "TCC_HIT_sum" in a:
        h, m = a["TCC_HIT_sum"], a.get("TCC_MISS_sum", 0)
        print(f"   L2 hit rate {h/(h+m+1e-9):.3f}")
