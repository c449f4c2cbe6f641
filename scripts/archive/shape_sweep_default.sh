#!/bin/bash
# ms/step of the MLL+grad step over the shape grid of scripts/shape_sweep.sh, default schedules only (regression check
# against profiles/r02/e_shape_sweep.txt's first column)
run() { python bench.py --n $1 --batch $2 --steps 40 --no-rollouts --no-cpu-baseline --no-aux-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])"; }
printf "%-6s %-4s %10s\n" N B default
for n in 399 1000 2048 3000 4096; do
  for B in 1 2 4 8 16 32 64 128; do
    if [ $n -ge 3000 ] && [ $B -ge 128 ]; then continue; fi
    printf "%-6s %-4s %10s\n" $n $B $(run $n $B)
  done
done
