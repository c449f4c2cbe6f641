cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_batch64.py -x -q -m gpu 2>&1 | tail -5
python scripts/bench_f64_step.py 2>&1 | grep "^{" | tee gpurun_out/r06/f64_3.txt
