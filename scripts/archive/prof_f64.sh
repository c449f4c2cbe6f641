export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p64 -o p64 -- python $R/scripts/bench_f64.py > /dev/null 2>&1
cd $R
python scripts/trace_union.py gpurun_out/p64
