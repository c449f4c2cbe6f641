#!/bin/bash
export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
# Round-3 evidence in one go (GPU box, through gpurun): bench line, rocprofv3 kernel-trace summaries of the same command
# in both schedules, the rollout and fp64 kernels' stats, PMC traffic passes, and the tables DESIGN.md quotes.
# usage: scripts/collect_r03.sh [part ...]   parts: bench trace roll f64 pmc tables small   (default: all)
R=$PWD
OUT=$R/gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-bench trace roll f64 pmc tables small}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-aux-legs --no-rollouts"
for P in $PARTS; do case $P in
bench)
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json;;
trace)
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2 -o t2 -- python $R/bench.py $ARGS > $OUT/t2.log 2>&1
  VOLT_GROUPS=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t1 -o t1 -- python $R/bench.py $ARGS > $OUT/t1.log 2>&1
  cd $R
  python scripts/trace_union.py $OUT/t2 > $OUT/timed_schedule_union.txt
  python scripts/trace_union.py $OUT/t1 > $OUT/lockstep_union.txt
  cp $(find $OUT/t2 -name "*kernel_stats.csv" | head -1) $OUT/timed_schedule_kernel_stats.csv
  cp $(find $OUT/t1 -name "*kernel_stats.csv" | head -1) $OUT/lockstep_kernel_stats.csv
  cat $OUT/timed_schedule_union.txt $OUT/lockstep_union.txt;;
roll)
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roll -o roll -- python $R/scripts/bench_rollouts.py > $OUT/rollouts_bench.txt 2>&1
  cd $R
  cp $(find $OUT/roll -name "*kernel_stats.csv" | head -1) $OUT/rollouts_kernel_stats.csv
  tail -5 $OUT/rollouts_bench.txt; head -5 $OUT/rollouts_kernel_stats.csv;;
f64)
  cd /tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f64 -o f64 -- python $R/scripts/bench_f64_step.py r03 > $OUT/f64_table.txt 2>&1
  cd $R
  cp $(find $OUT/f64 -name "*kernel_stats.csv" | head -1) $OUT/f64_kernel_stats.csv
  grep "^{" $OUT/f64_table.txt; python scripts/tune_diag64.py > $OUT/diag64_phases.txt 2>&1; tail -18 $OUT/diag64_phases.txt;;
pmc)
  VOLT_GROUPS=1 PMC_PASSES="sq1 fetch write" scripts/pmc.sh r03 --no-aux-legs --no-rollouts
  python scripts/pmc_traffic.py gpurun_out/pmc_r03 4096 64 7 > gpurun_out/pmc_r03/traffic.log 2>&1
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp gpurun_out/pmc_r03/summary.txt $OUT/pmc_lockstep_summary.txt;;
tables)
  (echo "# scripts/small_batch.sh: ms/step of the MLL+grad step, N=4096"; bash scripts/small_batch.sh 1 2 3 4 6 8 12 16 24 32) | tee $OUT/small_batch_table.txt
  (echo "# scripts/configs_table.sh: the MLL+grad step at BASELINE's other configurations"; bash scripts/configs_table.sh) | tee $OUT/configs_table.txt;;
small)
  # the one-launch step for short series against the launch-per-column path (VOLT_SMALL_NMAX=0), the per-piece stamps of
  # one step at the reference's own size, and rocprofv3's view of the launch
  (python scripts/bench_small_step.py; VOLT_SMALL_NMAX=0 VOLT_LONG=0 python scripts/bench_small_step.py per-column) 2>&1 | grep -v amdgpu.ids | tee $OUT/small_step_table.txt
  (SHOW=DSRPUT python scripts/small_stamps.py 1 399; python scripts/small_stamps.py 8 399) 2>&1 | grep -v amdgpu.ids > $OUT/small_step_stamps.txt
  cd /tmp
  SHAPES=1x399,8x399,32x399 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small -o small -- python $R/scripts/bench_small_step.py > $OUT/small_trace.log 2>&1
  cd $R
  cp $(find $OUT/small -name "*kernel_stats.csv" | head -1) $OUT/small_step_kernel_stats.csv
  # one long series: the one-launch step with sliced early parts against the launch-per-column path
  (SHAPES=1x1100,1x1500,1x2048,1x3000,1x4096 python scripts/bench_small_step.py; SHAPES=1x1100,1x1500,1x2048,1x3000,1x4096 VOLT_LONG=0 python scripts/bench_small_step.py per-column) 2>&1 | grep -v amdgpu.ids | tee $OUT/long_step_table.txt
  SHOW=DSR python scripts/small_stamps.py 1 4096 2>&1 | grep -v amdgpu.ids > $OUT/long_step_stamps.txt
  cd /tmp
  SHAPES=1x4096 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/long -o long -- python $R/scripts/bench_small_step.py > $OUT/long_trace.log 2>&1
  cd $R
  cp $(find $OUT/long -name "*kernel_stats.csv" | head -1) $OUT/long_step_kernel_stats.csv
  head -4 $OUT/small_step_kernel_stats.csv;;
esac; done
rm -rf $OUT/t1 $OUT/t2 $OUT/roll $OUT/f64 $OUT/small $OUT/long
ls -la $OUT
