#!/bin/bash
export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
# ms/step of the MLL+grad step over a grid of shapes, default schedules against the alternatives (which knob would have
# been better where): scripts/shape_sweep.sh > gpurun_out/shape_sweep.txt
run() { env $3 python bench.py --n $1 --batch $2 --steps 40 --no-rollouts --no-cpu-baseline --no-aux-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])"; }
printf "%-6s %-4s %10s %10s %10s %10s\n" N B default groups=1 groups=2 sched=0
for n in 399 1000 2048 3000 4096; do
  for B in 1 2 4 8 16 32 64 128; do
    if [ $n -ge 3000 ] && [ $B -ge 128 ]; then continue; fi
    a=$(run $n $B "X=1"); b=$(run $n $B "VOLT_GROUPS=1"); c=$(run $n $B "VOLT_GROUPS=2"); d=$(run $n $B "VOLT_SCHED=0")
    printf "%-6s %-4s %10s %10s %10s %10s\n" $n $B $a $b $c $d
  done
done
