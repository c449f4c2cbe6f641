#!/bin/bash
# Round-5 evidence (GPU box, through gpurun).  usage: scripts/collect_r05.sh [part ...]
#   bench    the full bench line                                         -> profiles/r05/bench.json
#   trace    rocprofv3 --kernel-trace --stats of `bench.py --profile-only 10` (the one-launch step: the kernel's average
#            duration over the same ten launches reproduces roofline.class_ms) and of the launch-per-column schedule
#   pmc      SQ / FETCH / WRITE passes of `bench.py --profile-only 3`     -> profiles/pmc_traffic.json (+ source hash)
#   tables   stamps of the one-launch step, schedule choice, soak, pipeline, fp64 table
R=$PWD
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-bench trace pmc tables}
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
esac; done
rm -rf $OUT/po $OUT/pc
ls -la $OUT
