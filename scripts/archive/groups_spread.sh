export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
run() { python bench.py --batch $1 --n $2 --steps ${3:-100} --no-rollouts --no-cpu-baseline --no-aux-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('B=%-4d N=%-5d %8.3f ms/step  %6.1f TF/s' % ($1, $2, d['ms_per_step'], d['step_tflops']))"; }
for G in 2 4 8; do for SP in 0 320 600; do echo "GROUPS=$G SPREAD=$SP"; export VOLT_GROUPS=$G VOLT_PLAIN_SPREAD=$SP; run 64 2048; run 64 1400; run 64 3000 50; done; done
