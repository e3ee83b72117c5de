#!/bin/bash
# usage: dev/ab_libs.sh REPS lib1.so lib2.so ... -- alternating whole-fit (dev/fit_ab.py) and fused-pass (dev/pow3_bench.py) timings on one box
cd "$GRAFT_REPO_ROOT"
reps=$1; shift
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    echo "== $lib"
    PETAL_HIP_LIBRARY=$PWD/$lib python dev/fit_ab.py 2>&1 | grep -v amdgpu.ids | sed 's/knobs=.*//'
    PETAL_HIP_LIBRARY=$PWD/$lib python dev/pow3_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
