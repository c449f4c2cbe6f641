cd $GRAFT_REPO_ROOT
VOLT_EXTRA_FLAGS="-DVOLT_D64_TWICE" python scripts/tune_diag64.py 2>&1 | grep "chol32"
