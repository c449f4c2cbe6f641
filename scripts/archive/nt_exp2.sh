export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
export TMPDIR=/tmp
R=$PWD
for nt in 0 1; do
  sed -i "s/^#define VOLT_OUT_NT .*/#define VOLT_OUT_NT $nt/" volt_amd/csrc/chol.hip
  python -m volt_amd.build > /dev/null 2>&1
  python bench.py --steps 40 --no-cpu-baseline --no-rollouts --no-aux-legs | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('out_nt $nt', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['lockstep']['frac'])"
  cd /tmp
  VOLT_GROUPS=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ont_$nt -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-aux-legs --no-rollouts > /dev/null 2>&1
  cd $R
  python - <<PY
import csv, glob
tot = n = 0
for f in glob.glob("gpurun_out/ont_$nt/*counter_collection.csv") + glob.glob("gpurun_out/ont_$nt/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "factor_step_kernel<true>" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("out_nt $nt: factor_step<true> read GB per launch (x2):", 2 * tot * 1024 / max(n, 1) / 1e9, "launches", n)
PY
done
