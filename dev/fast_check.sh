#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu -x -k "rpca or fused or two_plane or means_folded or determinism or power_pass or low_iter or steering" 2>&1 | tail -4
python dev/fit_ab.py 2>&1 | grep -v amdgpu | sed 's/knobs=.*//'
PETAL_NO_POW3_FAST=1 python dev/fit_ab.py 2>&1 | grep -v amdgpu | sed 's/knobs=.*//'
