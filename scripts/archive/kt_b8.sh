export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for tgt in 256; do
VOLT_SPLITK_TARGET=$tgt rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_b8 -o kt -- python $R/bench.py --batch 8 --n 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-aux-legs > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/kt_b8/*kernel_trace.csv") + glob.glob("gpurun_out/kt_b8/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
# last step: take the last 34 split launches
sp = [r for r in rows if "factor_step_split" in r[2]]
last = sp[-33:]
t0 = last[0][0]
for i, (s, e, nm, g) in enumerate(last):
    gap = (s - last[i-1][1]) / 1e3 if i else 0
    print(f"launch {i:2d} grid {g//256:5d} WGs  dur {(e-s)/1e3:7.1f} us  gap {gap:6.1f} us")
print("span", (last[-1][1] - t0) / 1e3, "us;  sum dur", sum(e - s for s, e, _, _ in last) / 1e3)
PY
