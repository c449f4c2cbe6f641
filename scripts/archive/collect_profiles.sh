#!/bin/bash
export VOLT_TUNE=1   # the VOLT_* schedule knobs are read only then (include/volt_hip_tune.h)
# End-of-round evidence: bench line + rocprofv3 kernel-trace summaries of the same command in both schedules.
# usage (GPU box): scripts/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
TAG=${1:-x}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-aux-legs --no-rollouts"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t4 -o t4 -- python $R/bench.py $ARGS > $OUT/t4.log 2>&1
VOLT_GROUPS=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t1 -o t1 -- python $R/bench.py $ARGS > $OUT/t1.log 2>&1
cd $R
python scripts/trace_union.py $OUT/t4 > $OUT/t4_union.txt
python scripts/trace_union.py $OUT/t1 > $OUT/t1_union.txt
cat $OUT/t4_union.txt $OUT/t1_union.txt
tail -c 600 $OUT/bench.json
