cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
SH="8x4096 12x4096 16x4096 24x4096 16x2048 24x2048 32x2048 64x2048 32x1536 64x1024 24x3072"
for sp in -1 0 100000; do
  echo "== VOLT_BATCH_SPREAD=$sp (-1: default rule)"
  if [ $sp -lt 0 ]; then python scripts/quick_step.py $SH 2>&1 | grep -v amdgpu; else VOLT_TUNE=1 VOLT_BATCH_SPREAD=$sp python scripts/quick_step.py $SH 2>&1 | grep -v amdgpu; fi
done | tee gpurun_out/r06/spread_sweep.txt
