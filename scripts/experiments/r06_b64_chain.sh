# fp64 1 x 4096 potrf: the latency chain of the one launch, with the diagonal block's own stamps (and the clock they ran at)
cd $GRAFT_REPO_ROOT
python scripts/tune_diag64.py 2>&1 | grep -v amdgpu | tail -3
VOLT_EXTRA_FLAGS="-DVOLT_B64_DIAG_STAMPS" python scripts/batch64_stamps.py 1x4096 potrf 2>&1 | grep -v amdgpu | grep "diagonal block\|launch span"
VOLT_EXTRA_FLAGS="-DVOLT_B64_DIAG_STAMPS" python scripts/batch64_stamps.py 8x4096 potrf 2>&1 | grep -v amdgpu | grep "diagonal block\|launch span"
