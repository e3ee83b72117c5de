#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for lib in "$@"; do
  PETAL_HIP_LIBRARY=$PWD/$lib python dev/fit_ab1.py 2>&1 | grep -v amdgpu
done; done
