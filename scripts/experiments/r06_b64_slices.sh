# when the D piece's slice-by-slice product sees its inputs: columns "sub-blocks 0, 1 seen" become: X tile seen | slice 3 of D(i-1)'s tile seen
cd $GRAFT_REPO_ROOT
VOLT_EXTRA_FLAGS="-DVOLT_B64_SLICE_STAMPS" python scripts/batch64_stamps.py 1x4096 potrf 2>&1 | grep "^  i 2[0-4]"
