#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu -x -k "rpca or fused or two_plane or means_folded or determinism or power_pass or low_iter or sharded" 2>&1 | tail -6
for knob in "" "PETAL_NO_POW3_FAST=1"; do
  echo "== $knob"
  env $knob python dev/fit_ab.py 2>&1 | grep -v amdgpu | sed 's/knobs=.*//'
  env $knob python dev/pow3_bench.py 2>&1 | grep -v amdgpu
done
python dev/low_iter_probe.py 2>&1 | grep -v amdgpu | grep "n_iter=3\|n_iter=5" | head -12
