export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
# sweep of the one-launch long step's tunables (run on the GPU box);  with arguments: same-box A/B of a -D switch, e.g.
# scripts/ab_split.sh -DVOLT_SOMETHING
export TMPDIR=/tmp
python -m volt_amd.build > /dev/null 2>&1
S=${SHAPES:-1x2048,1x3000,1x4096}
if [ $# -gt 0 ]; then
  for r in 1 2; do echo "== default build"; SHAPES=$S python scripts/bench_small_step.py 2>&1 | grep "B="; done
  export VOLT_EXTRA_FLAGS="$@"
  python -m volt_amd.build > /dev/null 2>&1
  for r in 1 2; do echo "== $VOLT_EXTRA_FLAGS"; SHAPES=$S python scripts/bench_small_step.py 2>&1 | grep "B="; done
  python -m pytest tests/test_gpu_contract.py -m gpu -x -q -k "short_series_run_as_one_launch" 2>&1 | tail -2
else
  echo "== defaults"; SHAPES=$S python scripts/bench_small_step.py 2>&1 | grep "B="
  for f in 3 4 5; do for e in 1 2; do echo "== FIRST=$f EMIN=$e"; VOLT_LONG_FIRST=$f VOLT_LONG_EMIN=$e SHAPES=$S python scripts/bench_small_step.py 2>&1 | grep "B="; done; done
  echo "== XCD=1"; VOLT_LONG_XCD=1 SHAPES=$S python scripts/bench_small_step.py 2>&1 | grep "B="
fi
