#!/bin/bash
# Per-launch durations (us) of the factorisation inside one MLL+grad step, for several schedules side by side:
#   scripts/kt_sched_table.sh B "ENV1=.. ENV2=.." "ENV.." ...      (each argument after B: the environment of one variant)
B=$1; shift
export TMPDIR=/tmp
R=$PWD
i=0
for v in "$@"; do
  rm -rf $R/gpurun_out/kts_$i
  (cd /tmp && env $v rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kts_$i -o kt -- python $R/bench.py --batch $B --n 4096 --steps 3 --warmup 1 --no-rollouts --no-cpu-baseline --no-aux-legs > /dev/null 2>&1)
  i=$((i+1))
done
python - "$@" <<'PY'
import csv, glob, sys
cols = []
for i, v in enumerate(sys.argv[1:]):
    rows = []
    for f in glob.glob(f"gpurun_out/kts_{i}/*kernel_trace.csv") + glob.glob(f"gpurun_out/kts_{i}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last TIMED step ends with its mll_scalars_kernel (bench.py's roofline leg, which runs behind it, has none):
    # the 33 factorisation launches in front of that one (32 columns + the trailing trtri row), whatever kernel each is
    end = max(i for i, r in enumerate(rows) if "mll_scalars" in r[2])
    last = [r for r in rows[:end] if "factor_step" in r[2]][-33:]
    cols.append([(e - s) / 1e3 for s, e, _ in last])
print("k    " + "  ".join(f"{v[:22]:>22s}" for v in sys.argv[1:]))
for k in range(33):
    print(f"{k:2d}   " + "  ".join(f"{c[k]:22.1f}" for c in cols))
print("sum  " + "  ".join(f"{sum(c):22.1f}" for c in cols))
PY
