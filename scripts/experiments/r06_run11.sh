cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_contract.py -x -q -m gpu -k "one_launch or long_series or short_series or 4096" 2>&1 | tail -3
python scripts/quick_step.py 1x4096 1x2048 1x1500 1x399 1x3000 2>&1 | grep -v amdgpu
VOLT_TUNE=1 VOLT_LONG_PULLERS=0 python scripts/quick_step.py 1x4096 1x2048 1x399 2>&1 | grep -v amdgpu
