set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for o in 0 54 22 118 50 86 26; do
  echo "== order $o"
  VOLT_TUNE=1 VOLT_BATCH_ORDER=$o python scripts/quick_step.py 64x4096 64x2048 32x4096 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06/order_sweep.txt
