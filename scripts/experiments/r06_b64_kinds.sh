# per-kind piece durations of the fp64 one launch at 8 x 4096 and 32 x 2048 (potrf), and the bench rows beside them
cd $GRAFT_REPO_ROOT
for sh in 8x4096 32x2048; do python scripts/batch64_stamps.py $sh potrf 2>&1 | grep -v amdgpu | sed -n 1,7p; done
python scripts/bench_f64_step.py 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['B'], d['N'], 'potrf', d['potrf_ms'], 'trtri', d['trtri_ms'], 'step', d['mll_step_ms'], 'fwd', d['mll_fwd_ms'])"
