cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_contract.py tests/test_gpu_edge.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
python scripts/quick_step.py 64x399 40x399 8x399 16x512 130x300 1x399 2>&1 | grep -v amdgpu
