# the fp64 diagonal block's pivot phase with lane-kept reciprocal pivots and DPP row_newbcast updates: tests, stamps, fp64 tables
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_contract.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -3
python scripts/tune_diag64.py 2>&1 | grep -v amdgpu | tail -14
python scripts/bench_f64_step.py 2>&1 | grep "^{" | cut -c1-400
