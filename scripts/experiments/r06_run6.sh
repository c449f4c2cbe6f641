set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_batch_step.py tests/test_gpu_topology.py -x -q -m gpu 2>&1 | tail -4
python scripts/quick_step.py 64x4096 8x4096 16x4096 64x2048 32x4096 4x2048 24x2048 2>&1 | tee gpurun_out/r06/quick_ahead.txt
