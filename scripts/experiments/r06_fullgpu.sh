cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06/gputest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
