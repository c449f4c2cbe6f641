cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_contract.py -x -q -m gpu -k "f64 or fp64 or batch64 or rollout" 2>&1 | tail -3
python scripts/tune_diag64.py 2>&1 | grep -v amdgpu | tail -12
python scripts/bench_f64_step.py 2>&1 | grep "^{" | cut -c1-330
