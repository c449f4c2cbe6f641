#!/bin/bash
# Round-6 evidence (GPU box, through gpurun).  usage: scripts/collect_r06.sh [part ...]
#   bench    the full bench line                                         -> profiles/r06/bench.json
#   trace    rocprofv3 --kernel-trace --stats of `bench.py --profile-only 10` (the one-launch step: the kernel's average
#            duration over the same ten launches reproduces roofline.class_ms)
#   pmc      SQ / FETCH / WRITE passes of `bench.py --profile-only 3`     -> profiles/pmc_traffic.json (+ source hash)
#   rollout  kernel trace + issue counters of the rollout kernel at C5's share -> profiles/rollout_pmc.json (+ source hash)
#   tables   stamps of the one-launch step, schedule choice, soak, pipeline (+ its kernel trace), fp64 table, resource usage
R=$PWD
OUT=$R/gpurun_out/r06c
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-bench trace pmc rollout tables}
for P in $PARTS; do case $P in
bench)
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json;;
trace)
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/po -o po -- python $R/bench.py --profile-only 10 > $OUT/profile_only.log 2>&1
  cd $R
  grep "^{" $OUT/profile_only.log > $OUT/profile_only.json
  cp $(find $OUT/po -name "*kernel_stats.csv" | head -1) $OUT/profile_only_kernel_stats.csv
  cat $OUT/profile_only.json; head -3 $OUT/profile_only_kernel_stats.csv | cut -c1-60,180-330;;
pmc)
  PMC_PASSES="sq1 fetch write" scripts/pmc.sh r06 --profile-only 3
  python scripts/pmc_traffic.py gpurun_out/pmc_r06 4096 64 3 > gpurun_out/pmc_r06/traffic.log 2>&1
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp gpurun_out/pmc_r06/summary.txt $OUT/pmc_summary.txt
  python - <<'PY'
import json; d=json.load(open("profiles/pmc_traffic.json")); print({k:v for k,v in d.items() if k!="kernels"}); print({k:v for k,v in d["kernels"].items() if k.startswith("batch_step")})
PY
  ;;
rollout)
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ro -o ro -- python $R/scripts/bench_rollouts.py > $OUT/rollout_trace.log 2>&1
  cd $R
  cp $(find $OUT/ro -name "*kernel_stats.csv" | head -1) $OUT/rollout_kernel_stats.csv
  grep "^{" $OUT/rollout_trace.log > $OUT/rollout_bench.json
  PMC_TIMEOUT=400 bash scripts/pmc_rollout_issue.sh > $OUT/rollout_pmc.log 2>&1
  cp gpurun_out/pmc_rollissue/rollout_pmc.json $OUT/rollout_pmc.json
  head -4 $OUT/rollout_kernel_stats.csv | cut -c1-80,150-300; python -c "import json;d=json.load(open('$OUT/rollout_pmc.json'));print({k:v for k,v in d.items() if k!='per_dispatch'})";;
tables)
  export VOLT_TUNE=1
  (VOLT_BATCH=2 python scripts/batch_stamps.py 64x4096; VOLT_BATCH=2 python scripts/batch_stamps.py 8x4096; VOLT_BATCH=2 python scripts/batch_stamps.py 64x2048) 2>&1 | grep -v amdgpu.ids > $OUT/batch_stamps.txt
  unset VOLT_TUNE
  python scripts/sched_choice_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/sched_choice.txt
  python scripts/soak.py 6 2>&1 | grep -v amdgpu.ids | tee $OUT/soak.txt
  python scripts/bench_pipeline.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pipeline.txt
  python scripts/bench_f64_step.py 2>&1 | grep "^{" > $OUT/f64_table.txt
  python scripts/batch64_check.py 1x512 2x1000 5x300 1x4096 3x2048 8x1024 8x4096 24x700 --reps-check 5 2>&1 | grep -v amdgpu.ids > $OUT/batch64_check.txt; tail -3 $OUT/batch64_check.txt | cut -c1-200
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe -o pipe -- python $R/scripts/experiments/r06_pipeline_one.py > $OUT/pipeline_trace.log 2>&1
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o c2 -- python $R/scripts/quick_step.py 1x4096 1x399 64x399 > $OUT/long_step_trace.log 2>&1
  cd $R
  cp $(find $OUT/pipe -name "*kernel_stats.csv" | head -1) $OUT/pipeline_kernel_stats.csv
  cp $(find $OUT/c2 -name "*kernel_stats.csv" | head -1) $OUT/long_step_kernel_stats.csv
  python scripts/resource_usage.py > $OUT/resource_usage.txt 2>/dev/null;;
esac; done
rm -rf $OUT/po $OUT/ro $OUT/pipe $OUT/c2
ls -la $OUT
