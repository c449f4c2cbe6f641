"""All launches of the LAST call in a rocprofv3 --kernel-trace CSV (a call starts with a prepare kernel): start, end, duration,
queue, kernel, workgroups.  usage: launch_list.py <dir> [marker-substring]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "*kernel_trace.csv"))[0]
mark = sys.argv[2] if len(sys.argv) > 2 else "prepare"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
rows = rows[st[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r['Queue_Id']:>2s} {r['Kernel_Name'].split('(')[0].replace('void ', '').replace('volt::', '')[:36]:36s} {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))}")
