#!/bin/bash
# PMC passes for the bench (run on the GPU box through gpurun).  usage: scripts/pmc.sh <tag> [bench args]
# Each pass is its own rocprofv3 run (counters + --kernel-trace only) under a hard timeout: a TCC pass
# aborted rocprofv3 and hung for 25 minutes in round 1, so nothing here may run unbounded.
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
BARGS=("$@")
pass() { # name counters...
  NAME=$1; shift
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$NAME -o $NAME -- \
      python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "${BARGS[@]}" > $R/gpurun_out/pmc_$TAG/$NAME.log 2>&1
  echo "pass $NAME rc=$?"
}
for P in ${PMC_PASSES:-sq1 sq2 fetch write hit}; do
  case $P in
    sq1) pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE;;
    sq2) pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE;;
    fetch) pass fetch FETCH_SIZE;;
    write) pass write WRITE_SIZE;;
    hit) pass hit TCC_HIT_sum TCC_MISS_sum;;
  esac
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt 2>&1
cat gpurun_out/pmc_$TAG/summary.txt
