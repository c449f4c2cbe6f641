export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
set -x
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_BUBBLE[A-Z_0-9]*" | sort -u | head -40 > gpurun_out/tcc_counters.txt
VOLT_GROUPS=1 PMC_PASSES="sq1 fetch write" scripts/pmc.sh r02 --no-aux-legs --no-rollouts
python scripts/pmc_traffic.py gpurun_out/pmc_r02 4096 64 7 > gpurun_out/pmc_r02/traffic.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_r02.json
PMC_PASSES="fetch write" scripts/pmc.sh r02roll --steps 1
