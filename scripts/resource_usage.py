"""Per-kernel register / LDS / spill table from hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed).
    python scripts/resource_usage.py > profiles/r02/resource_usage.txt"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volt_amd.build import CSRC, FLAGS, SOURCES, _hipcc

rows = []
for src in SOURCES:
    cmd = [_hipcc(), *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"file": src, "kernel": re.sub(r"\(.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
cols = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize", "Occupancy", "LDS Size"]
print(f"{'file':12s} {'kernel':48s} " + " ".join(f"{c:>12s}" for c in cols))
for r in rows:
    print(f"{r['file']:12s} {r['kernel'][:48]:48s} " + " ".join(f"{r.get(c, 0):12d}" for c in cols))
