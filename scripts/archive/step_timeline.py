"""From a rocprofv3 --kernel-trace CSV of bench.py: the launches of ONE timed step (the last complete one before the
roofline leg), start/end relative to the step's first launch, per queue -- what the end of a step looks like (last
block columns, last trtri row, the O(N^2) tail).  Usage: step_timeline.py <dir> [step_index_from_end]"""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "*kernel_trace.csv")) + glob.glob(os.path.join(root, "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
                     r["Kernel_Name"].split("(")[0].replace("void ", "").replace("volt::", ""), r.get("Grid_Size_X", "")))
rows.sort()
# a step starts with pad_resid_kernel
starts = [i for i, r in enumerate(rows) if r[3].startswith("pad_resid")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = starts[which]
b = starts[which + 1] if which + 1 < len(starts) else len(rows)
t0 = rows[a][0]
print(f"# step {which}: {b - a} launches, {(max(r[1] for r in rows[a:b]) - t0) / 1e3:.1f} us")
for s, e, q, name, g in rows[a:b]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{q:>3s}  {name[:40]:40s} {g}")
