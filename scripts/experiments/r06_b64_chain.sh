# fp64 1 x 4096 potrf: the latency chain of the one launch, with the diagonal block's own stamps (and the clock they ran at)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/tune_diag64.py 2>&1 | grep -v amdgpu > gpurun_out/tune_diag64.txt
VOLT_EXTRA_FLAGS="-DVOLT_B64_DIAG_STAMPS" python scripts/batch64_stamps.py 1x4096 potrf 2>&1 | grep -v amdgpu > gpurun_out/b64_chain_1x4096.txt
VOLT_EXTRA_FLAGS="-DVOLT_B64_DIAG_STAMPS" python scripts/batch64_stamps.py 1x4096 step 2>&1 | grep -v amdgpu > gpurun_out/b64_chain_1x4096_step.txt
tail -4 gpurun_out/b64_chain_1x4096.txt
