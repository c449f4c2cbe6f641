cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for o in 58 30; do
  echo "== order $o"
  VOLT_TUNE=1 VOLT_BATCH_ORDER=$o python scripts/quick_step.py 64x4096 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06/order_sweep2.txt
ORDERS="26 58 30" bash scripts/experiments/r06_order_traffic.sh 2>&1 | grep "^order" | tee gpurun_out/r06/order_traffic2.txt
