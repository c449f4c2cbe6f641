#!/bin/bash
# ms/step of the MLL+grad step at BASELINE's other configurations ("Other BASELINE configurations")
run() { python bench.py --batch $1 --n $2 --steps ${3:-100} --no-rollouts --no-cpu-baseline --no-aux-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('B=%-4d N=%-5d %8.3f ms/step  %6.1f TF/s' % ($1, $2, d['ms_per_step'], d['step_tflops']))"; }
run 1 4096; run 64 2048; run 32 4096; run 16 4096; run 8 4096; run 512 399 200; run 1 399 300; run 1 256 300
