# the latency chain of matrix 0 where every hand-off is XCD-local (8 x N: one XCD per matrix, no fences), beside 1 x N (agent scope)
cd $GRAFT_REPO_ROOT
for sh in 8x4096 8x2048 1x2048; do python scripts/batch64_stamps.py $sh potrf 2>&1 | grep -v "amdgpu\|warning\|^ *[0-9]* |\|^ *|\|^In file\|generated" | grep "launch span\|^  i  *\(9\|1[0-4]\) " ; done
