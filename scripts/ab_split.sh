# sweep of the one-launch long step's tunables with the split spine (run on the GPU box)
export TMPDIR=/tmp
python -m volt_amd.build > /dev/null 2>&1
for f in 3 4; do for e in 0 1 2; do echo "== FIRST=$f EMIN=$e"; VOLT_LONG_FIRST=$f VOLT_LONG_EMIN=$e SHAPES=1x1500,1x2048,1x3000,1x4096 python scripts/bench_small_step.py 2>&1 | grep "B="; done; done
for m in 2 3 4 7; do echo "== LONG_NMIN=$m"; VOLT_LONG_NMIN=$m SHAPES=1x256,1x399,1x512,1x640,1x768,1x1023 python scripts/bench_small_step.py 2>&1 | grep "B="; done
