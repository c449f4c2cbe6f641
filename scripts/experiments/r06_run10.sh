cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_contract.py tests/test_gpu_baselines.py tests/test_gpu_gpcv.py -x -q -m gpu 2>&1 | tail -3
python scripts/bench_pipeline.py 2>&1 | grep -v amdgpu.ids
