set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/test_gpu_contract.py tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_batch64.py -x -q -m gpu 2>&1 | tail -15
python scripts/bench_rollouts.py 2>&1 | tail -3 | tee gpurun_out/r06/rollouts_1.txt
python scripts/bench_f64_step.py 2>&1 | tail -6 | tee gpurun_out/r06/f64_2.txt
python bench.py --steps 10 --no-aux-legs --with-rollouts --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get(\"rollouts\"),indent=1)[:3000])"
