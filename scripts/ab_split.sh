export TMPDIR=/tmp
S1=1x399,1x640,1x1023,8x399,8x1023,16x640,64x399,1x1100,1x2048,1x4096
echo "== default (deferred publish, split=1)"; SHAPES=$S1 python scripts/bench_small_step.py 2>&1 | grep -v amdgpu.ids
for f in 2 3 6; do echo "== LONG_FIRST=$f"; VOLT_LONG_FIRST=$f SHAPES=1x2048,1x4096 python scripts/bench_small_step.py 2>&1 | grep "B="; done
for e in 1 3 4; do echo "== LONG_EMIN=$e"; VOLT_LONG_EMIN=$e SHAPES=1x2048,1x4096 python scripts/bench_small_step.py 2>&1 | grep "B="; done
echo "== LONG_XCD=1"; VOLT_LONG_XCD=1 SHAPES=1x2048,1x4096 python scripts/bench_small_step.py 2>&1 | grep "B="
echo "== LONG_NMIN=4 (5..8 block columns through the long kernel)"; VOLT_LONG_NMIN=4 SHAPES=1x640,1x768,1x1023 python scripts/bench_small_step.py 2>&1 | grep "B="
echo "== publish now"; export VOLT_EXTRA_FLAGS=-DVOLT_PUBLISH_NOW; python -m volt_amd.build > /dev/null 2>&1; SHAPES=$S1 python scripts/bench_small_step.py 2>&1 | grep -v amdgpu.ids
VOLT_LONG_SPLIT=0 SHAPES=1x2048,1x4096 python scripts/bench_small_step.py 2>&1 | grep "B="
