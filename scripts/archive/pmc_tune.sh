#!/bin/bash
# SQ counters of the P1 tuning kernels (scripts/tune_gemm.py).  usage: scripts/pmc_tune.sh <tag> <variants>
TAG=$1; VARS=$2
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/sq1 -o sq1 -- python $R/scripts/tune_gemm.py $VARS > $R/gpurun_out/pmc_$TAG/sq1.log 2>&1
echo "rc=$?"
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_$TAG/summary.txt
