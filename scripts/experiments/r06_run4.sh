set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/test_gpu_contract.py tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_baselines.py tests/test_gpu_edge.py -x -q -m gpu -k "rollout or Rollout or nonvol or prediction or Prediction or forecast" 2>&1 | tail -5
PMC_TIMEOUT=400 bash scripts/pmc_rollout_issue.sh 2>&1 | tail -30
