# kernel trace of the reference's per-window pipeline at its default sizes (scripts/bench_pipeline.py): which kernels an
# iteration of LearnGPCV / TrainVolModel / TrainVoltMagpieModel is made of
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/r06/pipe
export TMPDIR=/tmp
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/pipe -o pipe -- python $R/scripts/experiments/r06_pipeline_one.py > $R/gpurun_out/r06/pipe/log.txt 2>&1
cd $R
tail -5 gpurun_out/r06/pipe/log.txt
ls gpurun_out/r06/pipe
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r06/pipe/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:45]: print(r["Name"][:90].ljust(90), r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
