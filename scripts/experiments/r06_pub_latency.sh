# (needs scripts/experiments/r06_pub_latency_stamps.diff: two stamps around the publication of sub-block 0 in the tuning build)
# measured: wave 3 enters the publication 9.6 us after the block's entry, the word is stored and acknowledged at 10.7 (agent scope, 1 x 4096)
# and 9.8 -> 10.8 with XCD-local hand-offs (8 x 2048): the release itself is 1.1 us, not the 7 us between the barrier and the consumer
cd $GRAFT_REPO_ROOT
VOLT_EXTRA_FLAGS="-DVOLT_B64_DIAG_STAMPS" python scripts/batch64_stamps.py 1x4096 potrf 2>&1 | grep "^  i 2[0-3]\|diagonal block 16" | cut -c1-150,380-700
VOLT_EXTRA_FLAGS="-DVOLT_B64_DIAG_STAMPS" python scripts/batch64_stamps.py 8x2048 potrf 2>&1 | grep "^  i 1[0-2]\|diagonal block 8" | cut -c1-150,380-700
