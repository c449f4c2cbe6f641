#!/bin/bash
# ms/step of the MLL+grad step at small batch sizes: bench.py's own timed loop, nothing else
for B in ${@:-1 2 4 8 12 16}; do
  python bench.py --batch $B --steps 100 --no-rollouts --no-cpu-baseline --no-aux-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('B=%d  %.3f ms/step' % ($B, d['ms_per_step']))"
done
