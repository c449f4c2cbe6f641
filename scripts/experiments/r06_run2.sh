set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_topology.py tests/test_gpu_batch64.py tests/test_gpu_batch_step.py -x -q -m gpu 2>&1 | tail -15
