set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_batch_step.py tests/test_gpu_topology.py -x -q -m gpu 2>&1 | tail -5
python scripts/quick_step.py 64x4096 8x4096 16x4096 64x2048 32x4096 2>&1 | tee gpurun_out/r06/quick_persist.txt
VOLT_TUNE=1 VOLT_BATCH_PULLERS=0 python scripts/quick_step.py 64x4096 8x4096 64x2048 2>&1 | tee gpurun_out/r06/quick_allgrid.txt
VOLT_TUNE=1 VOLT_BATCH_XSKEW=3 python scripts/quick_step.py 64x4096 8x4096 2>&1 | tee gpurun_out/r06/quick_skew.txt
VOLT_TUNE=1 VOLT_BATCH_XDROP=0x0b python scripts/quick_step.py 64x4096 8x4096 2>&1 | tee gpurun_out/r06/quick_drop.txt
VOLT_TUNE=1 VOLT_BATCH_LOCAL=0 python scripts/quick_step.py 64x4096 8x4096 2>&1 | tee gpurun_out/r06/quick_nolocal.txt
