#!/bin/bash
# Round-5 evidence (GPU box, through gpurun).  usage: scripts/collect_r05.sh [part ...]
#   bench    the full bench line                                         -> profiles/r05/bench.json
#   trace    rocprofv3 --kernel-trace --stats of `bench.py --profile-only 10` (the one-launch step: the kernel's average
#            duration over the same ten launches reproduces roofline.class_ms) and of the launch-per-column schedule
#   pmc      SQ / FETCH / WRITE passes of `bench.py --profile-only 3`     -> profiles/pmc_traffic.json (+ source hash)
#   tables   stamps of the one-launch step, schedule choice, soak, pipeline, fp64 table
#   fp64     the fp64 one-launch step: gate sweeps against chol64.hip's schedules, errors vs LAPACK + repeatability, stamps,
#            diagonal-block phases, kernel trace of ten factorisations of 1 x 4096 and 8 x 4096
R=$PWD
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-bench trace pmc tables fp64}
for P in $PARTS; do case $P in
bench)
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json;;
trace)
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/po -o po -- python $R/bench.py --profile-only 10 > $OUT/profile_only.log 2>&1
  VOLT_TUNE=1 VOLT_BATCH=0 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pc -o pc -- python $R/bench.py --profile-only 10 > $OUT/profile_only_per_column.log 2>&1
  cd $R
  grep "^{" $OUT/profile_only.log > $OUT/profile_only.json; grep "^{" $OUT/profile_only_per_column.log > $OUT/profile_only_per_column.json
  cp $(find $OUT/po -name "*kernel_stats.csv" | head -1) $OUT/profile_only_kernel_stats.csv
  cp $(find $OUT/pc -name "*kernel_stats.csv" | head -1) $OUT/profile_only_per_column_kernel_stats.csv
  python scripts/trace_union.py $OUT/pc > $OUT/profile_only_per_column_union.txt
  cat $OUT/profile_only.json; head -3 $OUT/profile_only_kernel_stats.csv | cut -c1-60,180-330; cat $OUT/profile_only_per_column.json; head -4 $OUT/profile_only_per_column_union.txt;;
pmc)
  PMC_PASSES="sq1 fetch write" scripts/pmc.sh r05 --profile-only 3
  python scripts/pmc_traffic.py gpurun_out/pmc_r05 4096 64 3 > gpurun_out/pmc_r05/traffic.log 2>&1
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp gpurun_out/pmc_r05/summary.txt $OUT/pmc_summary.txt
  python - <<'PY'
import json; d=json.load(open("profiles/pmc_traffic.json")); print({k:v for k,v in d.items() if k!="kernels"}); print({k:v for k,v in d["kernels"].items() if k.startswith("batch_step")})
PY
  ;;
tables)
  export VOLT_TUNE=1
  (VOLT_BATCH=2 python scripts/batch_stamps.py 64x4096; VOLT_BATCH=2 python scripts/batch_stamps.py 8x4096; VOLT_BATCH=2 python scripts/batch_stamps.py 64x2048) 2>&1 | grep -v amdgpu.ids > $OUT/batch_stamps.txt
  unset VOLT_TUNE
  python scripts/sched_choice_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/sched_choice.txt
  python scripts/soak.py 6 2>&1 | grep -v amdgpu.ids | tee $OUT/soak.txt
  python scripts/bench_pipeline.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pipeline.txt
  python scripts/resource_usage.py > $OUT/resource_usage.txt 2>/dev/null;;
fp64)
  python scripts/batch64_gate_sweep.py $OUT/batch64_gate_sweep.txt 2>&1 | grep -v amdgpu.ids | tail -3
  SHAPES=32x4096,48x4096,64x4096,96x4096,32x3072,48x3072,96x3072,48x2048,64x2048,96x2048,192x2048,64x1024,128x1024,256x1024,256x512,512x512,64x399 python scripts/batch64_gate_sweep.py $OUT/batch64_gate_sweep_large.txt 2>&1 | grep -v amdgpu.ids | tail -3
  python scripts/batch64_check.py 1x512 2x1000 5x300 1x4096 3x2048 8x1024 8x4096 24x700 --reps-check 5 2>&1 | grep -v amdgpu.ids > $OUT/batch64_check.txt; tail -3 $OUT/batch64_check.txt | cut -c1-200
  (python scripts/batch64_stamps.py 1x4096 potrf; python scripts/batch64_stamps.py 8x4096 step | head -16) 2>&1 | grep -v amdgpu.ids > $OUT/batch64_stamps.txt
  python scripts/tune_diag64.py 2>&1 | grep -v amdgpu.ids > $OUT/diag64_phases.txt
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f64 -o f64 -- python $R/scripts/f64_trace.py 2>&1 | grep "fp64 potrf" > $OUT/f64_trace.txt
  cd $R
  cp $(find $OUT/f64 -name "*kernel_stats.csv" | head -1) $OUT/f64_kernel_stats.csv
  python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/f64/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "batch64_step_kernel" in r["Kernel_Name"]:
        d[int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(out + "/f64_trace.txt", "a") as o:
    for g, v in sorted(d.items()):
        v = v[-10:]
        o.write(f"rocprofv3 kernel trace: batch64_step_kernel grid {g} threads: {len(v)} launches, average {sum(v) / len(v):.3f} ms, min {min(v):.3f}\n")
print(open(out + "/f64_trace.txt").read())
PY
  rm -rf $OUT/f64;;
esac; done
rm -rf $OUT/po $OUT/pc
ls -la $OUT
