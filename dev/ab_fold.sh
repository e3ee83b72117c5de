#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  python dev/fit_ab1.py 2>&1 | grep -v amdgpu
  PETAL_MEANS_FOLD_ROWS=50000 python dev/fit_ab1.py 2>&1 | grep -v amdgpu | sed 's/^/fold: /'
done
