# sweep of the one-launch long step's tunables with the split spine (run on the GPU box)
export TMPDIR=/tmp
python -m volt_amd.build > /dev/null 2>&1
VOLT_LONG_OWN=1 python -m pytest tests/test_gpu_contract.py -m gpu -x -q -k "short_series_run_as_one_launch" 2>&1 | tail -2
for own in 0 1 2; do for sp in 1 2; do echo "== VOLT_LONG_OWN=$own VOLT_LONG_SPLIT=$sp"; VOLT_LONG_OWN=$own VOLT_LONG_SPLIT=$sp SHAPES=1x1023,1x2048,1x4096 python scripts/bench_small_step.py 2>&1 | grep "B="; done; done
VOLT_LONG_OWN=1 SHOW=DSR python scripts/small_stamps.py 1 4096 2>&1 | grep -A3 "S(2[0-2])"
