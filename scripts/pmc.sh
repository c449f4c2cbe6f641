#!/bin/bash
# PMC passes for the bench (run on the GPU box through gpurun). usage: scripts/pmc.sh <tag> [bench args]
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
pass() { # name counters...
  NAME=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$NAME -o $NAME -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "${BARGS[@]}" > $R/gpurun_out/pmc_$TAG/$NAME.log 2>&1
}
BARGS=("$@")
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
pass tcc1 TCC_HIT_sum TCC_MISS_sum FETCH_SIZE
pass tcc2 WRITE_SIZE TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt 2>&1
cat gpurun_out/pmc_$TAG/summary.txt
