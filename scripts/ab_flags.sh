#!/bin/bash
# usage: ab.sh "<flags variant 1>" "<flags variant 2>" ... ; runs quick_step for each build
for v in "$@"; do
  echo "=== VOLT_EXTRA_FLAGS='$v'"
  VOLT_EXTRA_FLAGS="$v" python -m volt_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  VOLT_EXTRA_FLAGS="$v" python scripts/quick_step.py $SHAPES 2>&1 | grep -v amdgpu.ids
done
