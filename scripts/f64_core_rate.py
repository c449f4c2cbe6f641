"""Per-launch rate of the fp64 128x128 core from a rocprofv3 --kernel-trace of `VOLT_F64_LOOKAHEAD=0 scripts/f64_one.py 64 4096`
(one stream: update64 launch k has (32 - k) * 64 tiles of k K-blocks and the chip to itself).  usage: f64_core_rate.py <dir>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
upd = [r for r in rows if "update64" in r["Kernel_Name"]]
upd = upd[len(upd) // 2:]
out = []
for k, r in enumerate(upd, 1):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tiles = (32 - k) * 64
    if k in (4, 8, 12, 16, 20, 24, 28):
        out.append(f"k={k}: {d:.0f} us {tiles * k * 4.194304 / d:.1f} TF/s")
print("  ".join(out))
